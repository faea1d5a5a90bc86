"""A/B of kernel variants selected by the library's experiment word, interleaved rounds in ONE process on ONE device:
    python scratch/kbench_ab.py halo:0,2,34 up2:0,8 wgrad:0,16 gather:0,4"""
import sys, torch
sys.path.insert(0, '.')
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda:0')
lib = _lib.load()
jobs = dict(a.split(':') for a in sys.argv[1:] if ':' in a) or {'halo': '0,2'}
def setexp(v): lib.sbg_experiment_set(int(v))
def cl(t): return t.contiguous(memory_format=torch.channels_last)
torch.manual_seed(0)

def ab(tag, fn, variants, kind, rounds=5, reps=4):
    res = {v: [] for v in variants}; ref = None
    for rnd in range(rounds + 1):
        for v in variants:
            setexp(v)
            y = fn()
            if rnd == 0:
                y = y.float() if torch.is_tensor(y) else y
                if ref is None: ref = y
                elif torch.is_tensor(y) and not (v & 512):      # (a variant may sum the taps in another order: last-bit differences of the 16-bit result)
                    err = float((ref - y).abs().max()) / float(ref.abs().max())
                    assert err < 2.0 ** -6, f'{tag}: variant {v} changes the result ({err:.3e} of the largest element)'
                    if err: print(f'{tag}: variant {v} differs from the first by {err:.2e} of the largest element', flush=True)
                continue
            torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); _lib.prof_enable(False)
            recs = [r for r in _lib.prof_fetch() if r['kind'] == kind]
            ms = sum(r['ms'] for r in recs) / reps; fl = sum(r['flops'] for r in recs) / reps
            res[v].append((fl / ms / 1e9, ms * 1e3))
    setexp(0)
    for v in variants:
        t = sorted(x[0] for x in res[v]); u = sorted(x[1] for x in res[v])
        print(f'{tag:44s} variant {v:4d}: median {t[len(t)//2]:7.1f} TF ({u[len(u)//2]:8.1f} us)  min {t[0]:7.1f}  max {t[-1]:7.1f}', flush=True)

for job, vs in jobs.items():
    variants = [int(v) for v in vs.split(',')]
    if job == 'halo':
        for (n, cin, cout, r) in [(64, 128, 128, 256), (64, 256, 256, 128), (64, 512, 512, 64), (64, 512, 512, 32)]:
            x = cl(torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16)); w = (torch.randn(cout, cin, 3, 3, device=dev) / 30).to(torch.bfloat16)
            ab(f'halo conv3x3 {n}x{cin}->{cout}@{r}', lambda: cg._conv_forward(x, w, (1, 1), (1, 1)), variants, 'conv_igemm')
    if job == 'haloepi':       # the halo kernel with the synthesis layer's fused tail (demodulation, noise, bias, lrelu, gain, clamp)
        for (n, cin, cout, r) in [(64, 128, 128, 256), (64, 256, 256, 128), (64, 512, 512, 64)]:
            x = cl(torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16)); w = (torch.randn(cout, cin, 3, 3, device=dev) / 30).to(torch.bfloat16)
            epi = cg.Epilogue(oscale=torch.rand(n, cout, device=dev) + 0.5, noise=torch.randn(n, 1, r, r, device=dev), bias=torch.randn(cout, device=dev),
                              act='lrelu', alpha=0.2, gain=1.4, clamp=256.0)
            ab(f'halo conv3x3 + tail {n}x{cin}->{cout}@{r}', lambda: cg._conv_forward(x, w, (1, 1), (1, 1), epi=epi), variants, 'conv_igemm')
    if job == 'k64':           # stride-2 forward convolutions (the discriminator's conv1 layers / the data gradient of G's up-convolutions): 8-wave gather kernel
        for (n, cin, cout, r) in [(64, 128, 256, 257), (64, 256, 512, 129), (64, 512, 512, 65)]:
            x = cl(torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16)); w = (torch.randn(cout, cin, 3, 3, device=dev) / 30).to(torch.bfloat16)
            ab(f'conv3x3 s2 {n}x{cin}->{cout}@{r}', lambda: cg._conv_forward(x, w, (2, 2), (0, 0)), variants, 'conv_igemm')
    if job == 'up2':
        for (n, cin, cout, r) in [(64, 256, 128, 128), (64, 512, 256, 64), (64, 512, 512, 32)]:
            x = cl(torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16)); w = (torch.randn(cin, cout, 3, 3, device=dev) / 30).to(torch.bfloat16)
            ab(f'convT s2 {n}x{cin}->{cout}@{r}', lambda: cg._conv_transpose_forward(x, w, (2, 2), (0, 0), (0, 0)), variants, 'conv_igemm')
    if job == 'gather':
        for (n, cin, cout, r) in [(64, 512, 512, 8), (64, 512, 512, 16)]:
            x = cl(torch.randn(n, cin, r, r, device=dev, dtype=torch.bfloat16)); w = (torch.randn(cin, cout, 3, 3, device=dev) / 30).to(torch.bfloat16)
            ab(f'convT s2 (gather) {n}x{cin}->{cout}@{r}', lambda: cg._conv_transpose_forward(x, w, (2, 2), (0, 0), (0, 0)), variants, 'conv_igemm')
    if job == 'wgrad':
        taps = [(i - 1, j - 1) for i in range(3) for j in range(3)]
        for (n, ca, cb, r) in [(64, 128, 128, 256), (64, 256, 256, 128), (64, 512, 512, 64)]:
            a = cl(torch.randn(n, ca, r, r, device=dev, dtype=torch.bfloat16)); b = cl(torch.randn(n, cb, r, r, device=dev, dtype=torch.bfloat16))
            ab(f'wgrad3x3 {ca}x{cb}@{r}', lambda: cg._wgrad(a, b, 1, taps), variants, 'conv_wgrad')
        taps2 = [(i, j) for i in range(3) for j in range(3)]
        for (n, ca, cb, r) in [(64, 256, 128, 128), (64, 512, 256, 64)]:
            a = cl(torch.randn(n, ca, r, r, device=dev, dtype=torch.bfloat16)); b = cl(torch.randn(n, cb, 2 * r + 1, 2 * r + 1, device=dev, dtype=torch.bfloat16))
            ab(f'wgrad3x3 s2 {ca}x{cb}@{r}', lambda: cg._wgrad(a, b, 2, taps2), variants, 'conv_wgrad')
