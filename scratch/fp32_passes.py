"""accuracy of the fp32 convolution paths (6 vs 3 bf16 products) against float64 on the CPU"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
print('passes', cg.fp32_mfma_passes)
for (n, cin, cout, r) in [(4, 512, 512, 4), (8, 64, 64, 64), (4, 256, 256, 16), (2, 1024, 1024, 8)]:
    x = torch.randn(n, cin, r, r); w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = cg.conv2d(xd, wd, padding=1)
    dy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, [xd, wd], dy)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], dy.cpu().double())
    e = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / b.abs().max())
    print((n, cin, cout, r), 'y', f'{e(y, yr):.2e}', 'dx', f'{e(gx, gxr):.2e}', 'dw', f'{e(gw, gwr):.2e}', flush=True)
