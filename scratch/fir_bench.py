"""FIR kernel check + timing at the benchmark's shapes: sliding-window vs tile kernel (SBG_FIR_RPW=2), vs a depthwise-conv statement"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import upfirdn2d
dev = torch.device('cuda', 0)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
def run(x, pad, gain):
    return upfirdn2d.upfirdn2d(x, f, padding=pad, gain=gain)
torch.manual_seed(0)
for (n, c, h, w, pad) in [(2, 64, 17, 17, 1), (2, 128, 33, 37, 1), (1, 64, 16, 16, 2), (3, 192, 65, 65, 1), (2, 64, 40, 257, 2), (1, 64, 257, 40, 1), (2, 64, 100, 31, 2)]:
    x = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = run(x, pad, 4.0)
    xr = torch.nn.functional.pad(x.float(), [pad] * 4)
    ref = torch.nn.functional.conv2d(xr, (f * 4.0).flip([0, 1])[None, None].repeat(c, 1, 1, 1), groups=c)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    print((n, c, h, w, pad), tuple(y.shape), 'rel err', f'{err:.2e}', flush=True)
    assert err < 1e-2
for (n, c, h, pad) in [(32, 128, 257, 1), (32, 128, 256, 2), (32, 256, 129, 1), (32, 256, 128, 2), (32, 512, 65, 1), (32, 512, 64, 2), (32, 512, 33, 1), (32, 512, 32, 2)]:
    x = torch.randn(n, c, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3): y = run(x, pad, 4.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = run(x, pad, 4.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    by = (x.numel() + y.numel()) * 2
    print(f'{(n, c, h, pad)}: {dt * 1e6:8.1f} us  {by / dt / 1e9:7.1f} GB/s', flush=True)
