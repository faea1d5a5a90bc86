"""ablation timing of the halo conv kernel on the target shape (needs a -DSBG_K64_DEBUG build): SBG_K64_ABL bits 1 = no MFMA, 2 = no loads, 4 = no fragment reads, 8 = no epilogue"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
for (n, c, r) in [(32, 128, 256), (32, 512, 64)]:
    x = torch.randn(n, c, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) / 30).to(torch.bfloat16)
    for _ in range(3): y = cg._conv_forward(x, w, (1, 1), (1, 1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = cg._conv_forward(x, w, (1, 1), (1, 1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(os.environ.get('SBG_K64_ABL', '0'), (n, c, r), f'{dt * 1e6:8.1f} us  {2 * n * r * r * c * c * 9 / dt / 1e12:7.1f} TF', flush=True)
