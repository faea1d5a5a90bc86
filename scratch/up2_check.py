"""conv_up2 kernel: correctness against torch.conv_transpose2d (fp32 on the bf16-rounded operands) and timing at the benchmark's shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
torch.manual_seed(0)
def run(x, w):
    return cg._conv_transpose_forward(x, w, (2, 2), (0, 0), (0, 0))
for (n, ci, co, h, w_) in [(2, 64, 64, 8, 32), (2, 128, 192, 16, 64), (1, 96, 72, 24, 32), (3, 64, 136, 32, 32), (2, 256, 64, 16, 96), (2, 64, 64, 16, 16)]:
    x = torch.randn(n, ci, h, w_, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(ci, co, 3, 3, device=dev) / (ci ** 0.5)).to(torch.bfloat16)
    _lib.prof_enable(True); _lib.prof_fetch()
    y = run(x, w)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    codes = [r['dims'][6] for r in _lib.prof_fetch() if r['kind'] == 'conv_igemm']
    ref = torch.nn.functional.conv_transpose2d(x.float(), w.float(), stride=2)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print((n, ci, co, h, w_), tuple(y.shape), 'rel err', f'{err:.2e}', 'launch codes', codes, flush=True)
    assert y.shape == ref.shape and err < 1e-2, err
for (n, ci, co, h) in [(32, 256, 128, 128), (32, 512, 256, 64), (32, 512, 512, 32), (32, 512, 512, 16)]:
    x = torch.randn(n, ci, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(ci, co, 3, 3, device=dev) / 50).to(torch.bfloat16)
    for _ in range(3): y = run(x, w)
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch(); t0 = time.perf_counter()
    for _ in range(10): y = run(x, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    _lib.prof_enable(False); kern = {}
    for q in _lib.prof_fetch(): kern[(q['kind'], q['dims'][6])] = kern.get((q['kind'], q['dims'][6]), 0.0) + q['ms'] * 100
    print(os.environ.get('SBG_CONV_NO_UP2', '-'), (n, ci, co, h), f'{dt * 1e6:8.1f} us  {2 * n * h * h * ci * co * 9 / dt / 1e12:7.1f} TF', {k: round(v, 1) for k, v in kern.items()}, flush=True)
