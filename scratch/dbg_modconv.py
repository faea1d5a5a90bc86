import sys, torch, numpy as np
sys.path.insert(0, '.')
import style_big_gan_amd
import torch.nn.functional as F
from style_big_gan_amd.torch_utils.ops import modconv, bias_act
from style_big_gan_amd.train_parts import generators as GN
dev = torch.device('cuda:0'); torch.manual_seed(11)
def rel(a, b): return float((a.double().cpu()-b.double().cpu()).abs().max() / (b.double().cpu().abs().max() + 1e-12))
def ref_layer(x, w, s, nz, b, act, gain, clamp):
    ws = w[None] * s[:, None, :, None, None]
    dco = (ws.square().sum([2, 3, 4]) + 1e-8).rsqrt()
    c = F.conv2d(x * s[:, :, None, None], w, padding=1)
    pre = c * dco[:, :, None, None] + (nz if nz is not None else 0) + b[None, :, None, None]
    t = F.leaky_relu(pre, 0.2) if act == 'lrelu' else pre
    y = t * gain
    return y.clamp(-clamp, clamp) if clamp is not None else y
for (n, cin, cout, r, act, clamp, noise_kind) in [(2, 16, 128, 16, "lrelu", 2.0, "per_sample"), (3, 24, 64, 8, "lrelu", None, "const"), (2, 8, 128, 16, "linear", 1.0, None), (4, 64, 128, 32, "lrelu", None, "per_sample")]:
    x0 = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16)
    w0 = torch.randn(cout, cin, 3, 3, device=dev).to(torch.bfloat16).float()
    s0 = torch.randn(n, cin, device=dev) + 1
    b0 = torch.randn(cout, device=dev)
    nz0 = None if noise_kind is None else torch.randn((n, 1, r, r) if noise_kind == "per_sample" else (r, r), device=dev)
    dy0 = None
    res = []
    for mode in ('fused', 'unfused', 'ref'):
        dt = torch.float64 if mode == 'ref' else None
        x = (x0.double() if mode == 'ref' else x0.clone()).requires_grad_(True); w = (w0.double() if mode == 'ref' else w0.clone()).requires_grad_(True)
        s = (s0.double() if mode == 'ref' else s0.clone()).requires_grad_(True); b = (b0.double() if mode == 'ref' else b0.clone()).requires_grad_(True)
        nz = None if nz0 is None else (nz0.double() if mode == 'ref' else nz0.clone()).requires_grad_(True)
        if mode == 'fused':
            y = modconv.modconv_bias_act(x, w.to(x.dtype), s, GN.demod_coefficients(w, s), nz, b, padding=1, act=act, gain=1.3, clamp=clamp)
        elif mode == 'unfused':
            y = GN.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, padding=1)
            y = bias_act.bias_act(y, b.to(y.dtype), act=act, gain=1.3, clamp=clamp)
        else:
            y = ref_layer(x, w, s, nz, b, act, 1.3, clamp)
        if dy0 is None: dy0 = torch.randn(y.shape, device=dev).to(torch.bfloat16)
        leaves = [x, w, s, b] + ([nz] if nz is not None else [])
        g = torch.autograd.grad((y.double() * dy0.double()).sum(), leaves)
        res.append((y.detach(), [t.detach() for t in g]))
    names = ["dx", "dw", "dstyles", "db", "dnoise"]
    for k in (0, 1):
        print(act, noise_kind, clamp, ['fused  ', 'unfused'][k], 'y %.4f' % rel(res[k][0], res[2][0]), ' '.join(f'{nm} {rel(a, b):.4f}' for nm, a, b in zip(names, res[k][1], res[2][1])))
