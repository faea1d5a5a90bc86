"""kernel micro-benchmarks through the library's launch log: python scratch/kbench.py [what ...]"""
import sys, torch
sys.path.insert(0, '.')
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg, upfirdn2d, bias_act, modulate
dev = torch.device('cuda:0')
what = sys.argv[1:] or ['conv', 'wgrad', 'fir', 'dot']

def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    recs = _lib.prof_fetch()
    out = {}
    for r in recs:
        k = (r['kind'], r['dims'])
        o = out.setdefault(k, [0, 0.0, 0.0, 0.0]); o[0] += 1; o[1] += r['ms']; o[2] += r['flops']; o[3] += r['bytes']
    return out

def show(tag, out):
    for (kind, dims), (cnt, ms, fl, by) in out.items():
        print(f"{tag:34s} {kind:13s} {str(dims):48s} avg {ms/cnt*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF  {by/ms/1e6:8.1f} GB/s", flush=True)

torch.manual_seed(0)
if 'conv1' in what:
    x = torch.randn(32, 128, 256, 256, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, device=dev) / 30).to(torch.bfloat16)
    for halo in [True, False]:
        cg.use_halo_kernel = halo
        show(f'conv3x3 256 halo={halo}', timed(lambda: cg._conv_forward(x, w, (1, 1), (1, 1)), reps=3, warm=1))
    cg.use_halo_kernel = True
if 'conv2' in what:
    for (n, cin, cout, r) in [(32, 128, 128, 256), (32, 512, 512, 64)]:
        x = torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev) / 30).to(torch.bfloat16)
        show(f"conv3x3 {n}x{cin}->{cout}@{r}", timed(lambda: cg._conv_forward(x, w, (1, 1), (1, 1))))
if 'conv' in what:
    for (n, cin, cout, r) in [(32, 128, 128, 256), (32, 256, 256, 128), (32, 512, 512, 64), (32, 512, 512, 32), (32, 512, 512, 16), (32, 512, 512, 8)]:
        x = torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev) / 30).to(torch.bfloat16)
        for halo in [True, False]:
            cg.use_halo_kernel = halo
            show(f"conv3x3 {n}x{cin}->{cout}@{r} halo={halo}", timed(lambda: cg._conv_forward(x, w, (1, 1), (1, 1))))
        cg.use_halo_kernel = True
    # strided / transposed
    x = torch.randn(32, 128, 257, 257, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 128, 3, 3, device=dev) / 30).to(torch.bfloat16)
    show("conv3x3 s2 128->256 @257", timed(lambda: cg._conv_forward(x, w, (2, 2), (0, 0))))
    x = torch.randn(32, 256, 128, 128, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 128, 3, 3, device=dev) / 30).to(torch.bfloat16)
    show("convT s2 256->128 @128", timed(lambda: cg._conv_transpose_forward(x, w, (2, 2), (0, 0), (0, 0))))
if 'wgrad' in what:
    for (n, ca, cb, r) in [(32, 128, 128, 256), (32, 256, 256, 128), (32, 512, 512, 64), (32, 512, 512, 16)]:
        a = torch.randn(n, ca, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(n, cb, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        taps = [(i - 1, j - 1) for i in range(3) for j in range(3)]
        show(f"wgrad3x3 {ca}x{cb}@{r}", timed(lambda: cg._wgrad(a, b, 1, taps), reps=5))
if 'wgrad2' in what:
    for (n, ca, cb, r) in [(32, 256, 128, 128), (32, 512, 256, 64), (32, 512, 512, 32)]:     # stride 2: a on the coarse grid r x r, b on (2r+1)^2
        a = torch.randn(n, ca, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(n, cb, 2 * r + 1, 2 * r + 1, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        taps = [(i, j) for i in range(3) for j in range(3)]
        show(f"wgrad3x3 s2 {ca}x{cb}@{r}", timed(lambda: cg._wgrad(a, b, 2, taps), reps=5))
if 'fir' in what:
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(dev)
    x = torch.randn(32, 128, 257, 257, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    show("fir 4x4 pad1 gain4 @257 (post convT)", timed(lambda: upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4)))
    x = torch.randn(32, 128, 256, 256, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    show("fir 4x4 pad2 @256 (pre s2 conv)", timed(lambda: upfirdn2d.upfirdn2d(x, f, padding=[2, 2, 2, 2])))
    show("fir up2 (grad of down path)", timed(lambda: upfirdn2d.upfirdn2d(x[:, :, :128, :128].contiguous(memory_format=torch.channels_last), f, up=2, padding=[2, 1, 2, 1], gain=4)))
if 'dot' in what:
    x = torch.randn(32, 128, 256, 256, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.randn_like(x)
    show("dot_hw(u,v) 128@256", timed(lambda: modulate._dot_hw_launch(x, y)))
    show("dot_hw(u) 128@256", timed(lambda: modulate._dot_hw_launch(x, None)))
    a = torch.randn(32, 128, device=dev)
    show("scale_nc 128@256", timed(lambda: modulate._scale_nc_launch(x, a, None)))
    b = torch.randn(128, device=dev, dtype=torch.bfloat16)
    show("bias_act lrelu 128@256", timed(lambda: bias_act.bias_act(x, b, act='lrelu')))
