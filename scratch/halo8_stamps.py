"""per-barrier clock stamps of conv_halo8_kernel (experiment bits 512 + 1024), wave 0 of workgroup 0: arrival / release of every step's barrier"""
import sys, torch
sys.path.insert(0, '.')
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda:0'); lib = _lib.load()
for (n, c, r) in [(64, 128, 256), (64, 512, 64)]:
    x = torch.randn(n, c, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) / 30).to(torch.bfloat16)
    for _ in range(2): cg._conv_forward(x, w, (1, 1), (1, 1))
    lib.sbg_experiment_set(512 + 1024)
    y = cg._conv_forward(x, w, (1, 1), (1, 1))
    torch.cuda.synchronize(); lib.sbg_experiment_set(0)
    raw = y.permute(0, 2, 3, 1).contiguous().view(-1)[:1600].view(torch.int64).cpu().tolist()      # channel-minor memory order
    st = [v for v in raw[:400]]
    arr, rel = st[0::2], st[1::2]
    steps = [(rel[i] - rel[i - 1]) for i in range(1, len(rel))]
    waits = [(rel[i] - arr[i]) for i in range(len(rel))]
    k = c // 64 * 9
    print(f'C={c} steps/tile={k}; cycles between barrier releases (100 MHz ticks x?):')
    for t0 in range(0, min(len(steps), 3 * k + 9), 9):
        print('  ', t0, [s for s in steps[t0:t0 + 9]], ' wait@barrier', [w_ for w_ in waits[t0 + 1:t0 + 10]])
