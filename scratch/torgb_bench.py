"""ToRGB forward / backward kernels at the headline shapes: us and TB/s of algorithmic bytes (library launch log)"""
import sys
sys.path.insert(0, '.')
import torch
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.torch_utils.ops import torgb
dev = torch.device('cuda:0')
for (n, c, r) in [(64, 128, 256), (64, 256, 128), (64, 512, 64), (64, 512, 32)]:
    x = torch.randn(n, c, r, r, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wmod = torch.randn(n, 3, c, device=dev, requires_grad=True); b = torch.zeros(3, device=dev, requires_grad=True)
    def run():
        y = torgb.torgb(x, wmod, b, clamp=256.0)
        y.sum().backward()
    for _ in range(3): run()
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch()
    for _ in range(10): run()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    recs = [q for q in _lib.prof_fetch() if q['kind'] == 'torgb']
    for tag, sel in (('fwd', 0), ('bwd', 1)):
        rs = sorted(q['ms'] for q in recs if q['dims'][4] == sel)
        by = [q['bytes'] for q in recs if q['dims'][4] == sel][0]
        print(f'torgb {tag} [{n},{c},{r},{r}]: median {rs[len(rs) // 2] * 1e3:7.1f} us  {by / rs[len(rs) // 2] / 1e9:6.2f} TB/s', flush=True)
