"""grouped GEMM (style bank) timing at the headline sizes: forward, weight / bias gradients, w-slot gradients"""
import sys
sys.path.insert(0, '.')
import torch
import style_big_gan_amd
from style_big_gan_amd import _lib
from style_big_gan_amd.train_parts import generators
dev = torch.device('cuda:0')
syn = generators.SynthesisNetwork(w_dim=512, img_resolution=256, img_channels=3, channel_base=32768, channel_max=512, num_fp16_res=6).to(dev)
ws = torch.randn(64, syn.num_ws, 512, device=dev, requires_grad=True)
from style_big_gan_amd.torch_utils.ops import grouped_gemm
def run():
    bank = syn._style_bank(ws)
    outs = [s for v in bank.values() for s in v if s is not None]
    torch.autograd.grad(sum((o * o).sum() for o in outs), [ws] + [p for n, p in syn.named_parameters() if 'affine' in n])
for _ in range(3): run()
torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch()
for _ in range(10): run()
torch.cuda.synchronize(); _lib.prof_enable(False)
recs = [r for r in _lib.prof_fetch() if r['kind'] == 'grouped_gemm']
per = len(recs) // 10
for i in range(per):
    ms = sorted(r['ms'] for r in recs[i::per])
    print(f'launch {i}: problems {recs[i]["dims"][0]:2d} tiles {recs[i]["dims"][1]:4d}  median {ms[len(ms) // 2] * 1e3:7.1f} us  {recs[i]["flops"] / ms[len(ms) // 2] / 1e9:7.2f} TF')
