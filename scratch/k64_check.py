"""conv_k64 kernels vs an fp32 torch reference on the same inputs"""
import os, sys, torch
sys.path.insert(0, '.')
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda:0')
torch.manual_seed(0)
import torch.nn.functional as F
bad = 0
def run(tag, fn, ref):
    global bad
    y = fn().float()
    err = (y - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    ok = err < 2e-2
    bad += (not ok)
    print(f"{tag:60s} rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
cases = [(2, 128, 128, 64, 3), (3, 64, 192, 32, 3), (2, 72, 136, 16, 3), (2, 256, 256, 8, 3), (1, 8, 128, 64, 1), (2, 128, 3, 64, 1), (2, 40, 72, 24, 3), (4, 512, 512, 4, 3), (2,128,128,48,3)]
for (n, cin, cout, r, k) in cases:
    x = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16)
    ref = F.conv2d(x.float(), w.float(), padding=k // 2)
    run(f"conv{k}x{k} s1 n{n} {cin}->{cout} @{r}", lambda: cg._conv_forward(x, w, (1, 1), (k // 2, k // 2)), ref)
    if k == 3 and r >= 8:
        ref2 = F.conv2d(x.float(), w.float(), stride=2, padding=0)
        run(f"conv3x3 s2 n{n} {cin}->{cout} @{r}", lambda: cg._conv_forward(x, w, (2, 2), (0, 0)), ref2)
        wt = (torch.randn(cin, cout, 3, 3, device=dev) / (cin * 9) ** 0.5).to(torch.bfloat16)
        ref3 = F.conv_transpose2d(x.float(), wt.float(), stride=2)
        run(f"convT3x3 s2 n{n} {cin}->{cout} @{r}", lambda: cg._conv_transpose_forward(x, wt, (2, 2), (0, 0), (0, 0)), ref3)
# fused epilogue
n, cin, cout, r = 2, 128, 128, 32
x = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5).to(torch.bfloat16)
osc = torch.rand(n, cout, device=dev) + 0.5; noise = torch.randn(n, 1, r, r, device=dev); bias = torch.randn(cout, device=dev)
ref = F.conv2d(x.float(), w.float(), padding=1) * osc[:, :, None, None] + noise + bias[None, :, None, None]
ref = (F.leaky_relu(ref, 0.2) * 2 ** 0.5).clamp(-1.5, 1.5)
epi = cg.Epilogue(oscale=osc, noise=noise, bias=bias, act='lrelu', alpha=0.2, gain=2 ** 0.5, clamp=1.5)
run("fused epilogue 128->128 @32", lambda: cg._conv_forward(x, w, (1, 1), (1, 1), epi=epi), ref)
# fp32 path (split passes, accumulate)
x32 = torch.randn(2, 64, 16, 16, device=dev); w32 = torch.randn(96, 64, 3, 3, device=dev) / 24
ref = F.conv2d(x32.double(), w32.double(), padding=1).float()
y = cg._conv_forward(x32, w32, (1, 1), (1, 1))
err = (y - ref).abs().max().item() / ref.abs().max().item(); print(f"fp32 split conv rel err {err:.2e}", 'ok' if err < 2e-4 else 'FAIL'); bad += err >= 2e-4
x32 = torch.randn(8, 64, 64, 64, device=dev); ref = F.conv2d(x32.double(), w32.double(), padding=1).float()
cg.CONCAT_NUMEL = 0
y = cg._conv_forward(x32, w32, (1, 1), (1, 1))
err = (y - ref).abs().max().item() / ref.abs().max().item(); print(f"fp32 6-pass accumulate conv rel err {err:.2e}", 'ok' if err < 2e-4 else 'FAIL'); bad += err >= 2e-4
print('FAILURES', bad)
sys.exit(1 if bad else 0)
