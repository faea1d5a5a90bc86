"""thin-channel convolution kernel: correctness against torch on the CPU-computed reference (via F.conv2d on the device in fp32) + timing"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
def check(n, cin, cout, h, w, k, stride, transpose, dtype=torch.bfloat16):
    x = torch.randn(n, cin, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    if transpose:
        wt = (torch.randn(cin, cout, k, k, device=dev) / (cin * k * k) ** 0.5).to(dtype)
        y = cg._conv_transpose_forward(x, wt, (stride, stride), (0, 0), (0, 0))
        ref = F.conv_transpose2d(x.float(), wt.float(), stride=stride)
    else:
        wt = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).to(dtype)
        y = cg._conv_forward(x, wt, (stride, stride), (k // 2, k // 2))
        ref = F.conv2d(x.float(), wt.float(), stride=stride, padding=k // 2)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print(f'n{n} {cin}->{cout} {h}x{w} k{k} s{stride} T{int(transpose)} {tuple(y.shape)} rel err {err:.2e}', flush=True)
    assert y.shape == ref.shape and err < 2e-2, err
for args in [(2, 16, 16, 33, 37, 3, 1, False), (2, 32, 32, 20, 20, 3, 1, False), (2, 16, 32, 33, 33, 3, 2, False), (2, 32, 16, 17, 19, 1, 1, False),
             (3, 16, 8, 16, 16, 3, 1, False), (2, 32, 16, 16, 16, 3, 2, True), (2, 16, 16, 9, 11, 3, 2, True), (1, 64, 24, 12, 12, 3, 1, False),
             (2, 24, 64, 12, 12, 3, 1, False), (2, 8, 8, 40, 40, 3, 1, False)]:
    check(*args)
# fp32 output path / data gradient through autograd on a thin layer
x = torch.randn(2, 16, 24, 24, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
wt = (torch.randn(16, 16, 3, 3, device=dev) / 12).to(torch.bfloat16).requires_grad_(True)
y = cg.conv2d(x, wt, padding=1)
gx, gw = torch.autograd.grad(y.float().square().sum(), [x, wt])
xr, wr = x.detach().float().requires_grad_(True), wt.detach().float().requires_grad_(True)
yr = F.conv2d(xr, wr, padding=1)
gxr, gwr = torch.autograd.grad(yr.square().sum(), [xr, wr])
print('dx err', float((gx.float() - gxr).abs().max() / gxr.abs().max()), 'dw err', float((gw.float() - gwr).abs().max() / gwr.abs().max()))
# timing at ffhq_sg2's shapes
from style_big_gan_amd import _lib
for (n, cin, cout, r, k, s, tr) in [(32, 16, 16, 1024, 3, 1, False), (32, 32, 32, 512, 3, 1, False), (32, 32, 16, 512, 1, 1, False), (32, 16, 32, 1025, 3, 2, False), (32, 32, 16, 512, 3, 2, True), (32, 64, 32, 256, 3, 1, False), (32, 64, 64, 256, 3, 1, False)]:
    x = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(*( (cin, cout) if tr else (cout, cin)), k, k, device=dev) / 10).to(torch.bfloat16)
    fn = (lambda: cg._conv_transpose_forward(x, wt, (s, s), (0, 0), (0, 0))) if tr else (lambda: cg._conv_forward(x, wt, (s, s), (k // 2, k // 2)))
    for env in ('', '1'):
        pass
    for _ in range(2): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): y = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    by = (x.numel() + y.numel()) * 2
    print(f'{(n, cin, cout, r, k, s, tr)}: {dt * 1e6:9.1f} us  {by / dt / 1e9:7.1f} GB/s', flush=True)
