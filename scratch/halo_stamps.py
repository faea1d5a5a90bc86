"""per-barrier clock stamps of conv_halo_ld_kernel (diagnosis build with -DSBG_K64_STAMPS: scratch/libsbg_hip_stamps.so), workgroup 0, waves 0 (X compute),
4 (Y compute), 8 (weight loader), 10 (halo loader): for every barrier the wave's arrival and the release.  Who arrives last at which barrier?"""
import os, sys
os.environ['SBG_HIP_LIBRARY'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libsbg_hip_stamps.so')
sys.path.insert(0, '.')
import torch
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda:0')
for (n, c, r) in [(64, 512, 64), (64, 128, 256)]:
    x = torch.randn(n, c, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) / 30).to(torch.bfloat16)
    for _ in range(3): y = cg._conv_forward(x, w, (1, 1), (1, 1))
    torch.cuda.synchronize()
    raw = y.permute(0, 2, 3, 1).contiguous().view(-1)[:6 * 4096 * 4].view(torch.int64).cpu()
    names = {0: 'X', 2: 'Y', 4: 'W', 5: 'H'}
    st = {names[k]: raw[k * 4096:(k + 1) * 4096].tolist() for k in names}
    # barrier index alignment: X misses barrier 0 (Y's extra first), so X's i-th barrier is global i + ... : use release times to align
    def pairs(v): return list(zip(v[0::2], v[1::2]))
    P = {k: pairs(v) for k, v in st.items()}
    # global barrier g: Y, W, H take part in all; X's barrier j is global j + 1
    print(f'C={c}: barrier g (steady state, from g=200): arrival offsets relative to the release (cycles before release), who is last')
    base = 200
    for g in range(base, base + 40):
        rel = P['Y'][g][1]
        arr = {'Y': P['Y'][g][0], 'W': P['W'][g][0], 'H': P['H'][g][0], 'X': P['X'][g - 1][0]}
        dur = rel - P['Y'][g - 1][1]
        print(f'  g={g} ({"even" if g % 2 == 0 else "odd "}) interval {dur:5d}  waits: ' + ' '.join(f'{k}:{rel - a:5d}' for k, a in arr.items()))
