"""thin weight-gradient kernel: dw against torch autograd (fp32 on the device), then timing at ffhq_sg2's shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
def check(n, cin, cout, h, w, k, stride, transpose=False):
    x = torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    shape = (cin, cout, k, k) if transpose else (cout, cin, k, k)
    wt = (torch.randn(*shape, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16).requires_grad_(True)
    if transpose:
        y = cg.conv_transpose2d(x, wt, stride=stride)
        yr = F.conv_transpose2d(x.detach().float(), wt.detach().float().requires_grad_(True), stride=stride)
    else:
        y = cg.conv2d(x, wt, stride=stride, padding=k // 2)
    dy = torch.randn_like(y)
    (dw,) = torch.autograd.grad(y, wt, dy)
    xr, wr = x.detach().float(), wt.detach().float().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, stride=stride) if transpose else F.conv2d(xr, wr, stride=stride, padding=k // 2)
    (dwr,) = torch.autograd.grad(yr, wr, dy.float())
    err = float((dw.float() - dwr).abs().max() / dwr.abs().max())
    print(f'n{n} {cin}->{cout} {h}x{w} k{k} s{stride} T{int(transpose)} dw {tuple(dw.shape)} rel err {err:.2e}', flush=True)
    assert err < 2e-2, err
for args in [(2, 16, 16, 33, 37, 3, 1), (2, 32, 32, 20, 64, 3, 1), (2, 16, 32, 33, 33, 3, 2), (3, 32, 16, 17, 19, 1, 1), (2, 8, 24, 16, 40, 3, 1),
             (2, 32, 16, 16, 16, 3, 2, True), (2, 16, 16, 9, 11, 3, 2, True), (1, 24, 8, 12, 100, 3, 1), (2, 16, 16, 65, 65, 3, 2)]:
    check(*args)
from style_big_gan_amd import _lib
for (n, ca, cb, r, s) in [(32, 16, 16, 1024, 1), (32, 32, 32, 512, 1), (32, 32, 16, 512, 2)]:
    a = torch.randn(n, ca, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bsz = r if s == 1 else 2 * r + 1
    b = torch.randn(n, cb, bsz, bsz, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    taps = [(i - 1, j - 1) for i in range(3) for j in range(3)] if s == 1 else [(i, j) for i in range(3) for j in range(3)]
    for _ in range(2): out = cg._wgrad(a, b, s, taps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = cg._wgrad(a, b, s, taps)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'wgrad {(n, ca, cb, r, s)}: {dt * 1e6:9.1f} us  {(a.numel() + b.numel()) * 2 / dt / 1e9:7.1f} GB/s', flush=True)
