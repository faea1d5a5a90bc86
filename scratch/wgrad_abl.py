"""ablation timing of the rows weight-gradient kernel (needs a -DSBG_K64_DEBUG build): SBG_WGRAD_ABL bits 1 = no MFMA, 2 = no DMA inside the loop, 4 = no fragment reads"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
taps = [(i - 1, j - 1) for i in range(3) for j in range(3)]
for (n, c, r, s) in [(32, 128, 256, 1), (32, 512, 64, 1), (32, 256, 128, 2)]:
    ro = r // s
    a = torch.randn(n, c, ro, ro, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(n, c, r + (1 if s == 2 else 0), r + (1 if s == 2 else 0), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    tp = taps if s == 1 else [(i, j) for i in range(3) for j in range(3)]
    for _ in range(3): y = cg._wgrad(a, b, s, tp)
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch(); t0 = time.perf_counter()
    for _ in range(10): y = cg._wgrad(a, b, s, tp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    _lib.prof_enable(False); rec = _lib.prof_fetch()
    kern = {}
    for q in rec: kern[q['kind']] = kern.get(q['kind'], 0.0) + q['ms'] / 10
    print(os.environ.get('SBG_WGRAD_ABL', '0'), (n, c, r, s), f'{dt * 1e6:8.1f} us  {2 * n * ro * ro * c * c * 9 / dt / 1e12:7.1f} TF', {k: round(v * 1e3, 1) for k, v in kern.items()}, flush=True)
