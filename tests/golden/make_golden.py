"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE on CPU.

Run in the dev container only (the reference checkout does not travel to the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

It imports the reference's own eager CPU path -- ``stylegan2ada.torch_utils.ops.*`` (pure-PyTorch ``_ref`` branches),
``stylegan2ada.training.networks`` (the vendored original whose hot functions are byte-identical to
``train_parts/generators.py`` / ``discriminators.py`` but import without omegaconf) and ``biggan.layers`` -- feeds seeded
inputs and stores inputs + outputs (+ gradients, + second-order gradients where the hot path needs them).  Fixtures are
data only: tensors as float32 arrays and JSON-encoded argument lists.
"""
import itertools
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SBG_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from stylegan2ada.torch_utils.ops import bias_act as R_bias_act            # noqa: E402
from stylegan2ada.torch_utils.ops import conv2d_resample as R_resample    # noqa: E402
from stylegan2ada.torch_utils.ops import fma as R_fma                      # noqa: E402
from stylegan2ada.torch_utils.ops import upfirdn2d as R_upfirdn2d          # noqa: E402
from stylegan2ada.training import networks as R_net                        # noqa: E402
import stylegan2ada.dnnlib as dnnlib                                        # noqa: E402

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148]


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32) if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, arrays, meta):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"{name}: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------------------------------------------------------- ops

def gen_upfirdn2d():
    torch.manual_seed(100)
    arrays, cases = {}, []
    filters = {"k4": ([1, 3, 3, 1], {}), "sym6": (SYM6, {}), "none": (None, {}), "k4_flipped_gain2": ([1, 3, 3, 1], dict(flip_filter=True, gain=2))}
    for fname, (taps, fkw) in filters.items():
        f = R_upfirdn2d.setup_filter(taps, **fkw) if taps is not None else None
        if f is not None:
            arrays[f"f/{fname}"] = npy(f)
    idx = 0
    for shape in [(2, 3, 8, 8), (1, 4, 9, 9)]:
        for fname in filters:
            f = torch.from_numpy(arrays[f"f/{fname}"]) if f"f/{fname}" in arrays else None
            for (up, down), padding, flip, gain in itertools.product([(1, 1), (2, 1), (1, 2), (2, 2)], [0, [2, 1, 2, 1], [-1, 2, 3, -1]], [False, True], [1, 4]):
                fw = 1 if f is None else f.shape[-1]
                p = [padding] * 4 if isinstance(padding, int) else padding
                if shape[3] * up + p[0] + p[1] - fw < 0 or shape[2] * up + p[2] + p[3] - fw < 0:
                    continue
                x = torch.randn(shape, requires_grad=True)
                y = R_upfirdn2d.upfirdn2d(x, f, up=up, down=down, padding=padding, flip_filter=flip, gain=gain, impl='ref')
                dy = torch.randn_like(y).requires_grad_(True)
                dx = torch.autograd.grad((y * dy).sum(), x, create_graph=True)[0]
                v = torch.randn_like(x)
                ddy = torch.autograd.grad((dx * v).sum(), dy)[0]       # grad-of-grad (w.r.t. dy)
                key = f"c{idx}"
                arrays.update({f"{key}/x": npy(x), f"{key}/y": npy(y), f"{key}/dy": npy(dy), f"{key}/dx": npy(dx), f"{key}/v": npy(v), f"{key}/ddy": npy(ddy)})
                cases.append(dict(key=key, filter=fname, up=up, down=down, padding=padding, flip_filter=flip, gain=gain))
                idx += 1
    # wrappers
    f = torch.from_numpy(arrays["f/k4"])
    x = torch.randn(2, 3, 16, 16)
    arrays["wrap/x"] = npy(x)
    for name in ["filter2d", "upsample2d", "downsample2d"]:
        arrays[f"wrap/{name}"] = npy(getattr(R_upfirdn2d, name)(x, f, impl='ref'))
    save("upfirdn2d", arrays, dict(cases=cases, sym6=SYM6))


def gen_bias_act():
    torch.manual_seed(101)
    arrays, cases = {}, []
    idx = 0
    for act, spec in R_bias_act.activation_funcs.items():
        for shape in [(2, 5, 4, 4), (3, 7)]:
            for use_b, clamp, gain in itertools.product([False, True], [None, 0.5], [None, 2.0]):
                x = torch.randn(shape, requires_grad=True)
                b = torch.randn(shape[1], requires_grad=True) if use_b else None
                y = R_bias_act.bias_act(x, b, dim=1, act=act, clamp=clamp, gain=gain, impl='ref')
                dy = torch.randn_like(y).requires_grad_(True)
                ins = [x] + ([b] if use_b else [])
                g = torch.autograd.grad((y * dy).sum(), ins, create_graph=True)
                key = f"c{idx}"
                arrays.update({f"{key}/x": npy(x), f"{key}/y": npy(y), f"{key}/dy": npy(dy), f"{key}/dx": npy(g[0])})
                if use_b:
                    arrays[f"{key}/b"] = npy(b); arrays[f"{key}/db"] = npy(g[1])
                # second order: derivative of sum(dx * v) w.r.t. dy (always defined) and x (when it exists)
                v = torch.randn_like(x)
                arrays[f"{key}/v"] = npy(v)
                g2 = torch.autograd.grad((g[0] * v).sum(), [dy, x], allow_unused=True)
                arrays[f"{key}/d_dy"] = npy(g2[0])
                if g2[1] is not None:
                    arrays[f"{key}/d_x"] = npy(g2[1])
                cases.append(dict(key=key, act=act, use_b=use_b, clamp=clamp, gain=gain))
                idx += 1
    table = {k: dict(def_alpha=float(v.def_alpha), def_gain=float(v.def_gain), cuda_idx=int(v.cuda_idx), ref=v.ref, has_2nd_grad=bool(v.has_2nd_grad))
             for k, v in R_bias_act.activation_funcs.items()}
    save("bias_act", arrays, dict(cases=cases, activation_funcs=table))


def gen_conv2d_resample():
    torch.manual_seed(102)
    arrays, cases = {}, []
    f = R_upfirdn2d.setup_filter([1, 3, 3, 1])
    idx = 0
    for k, (up, down), flip_weight, groups in itertools.product([1, 3], [(1, 1), (2, 1), (1, 2), (2, 2)], [True, False], [1, 2]):
        x = torch.randn(2, 8, 8, 8, requires_grad=True)
        w = (torch.randn(6, 8 // groups, k, k) / np.sqrt(8 * k * k)).requires_grad_(True)
        y = R_resample.conv2d_resample(x, w, f=f, up=up, down=down, padding=k // 2, groups=groups, flip_weight=flip_weight)
        dy = torch.randn_like(y)
        dx, dw = torch.autograd.grad((y * dy).sum(), [x, w])
        key = f"c{idx}"
        arrays.update({f"{key}/x": npy(x), f"{key}/w": npy(w), f"{key}/y": npy(y), f"{key}/dy": npy(dy), f"{key}/dx": npy(dx), f"{key}/dw": npy(dw)})
        cases.append(dict(key=key, k=k, up=up, down=down, flip_weight=flip_weight, groups=groups))
        idx += 1
    save("conv2d_resample", arrays, dict(cases=cases))


def gen_modulated_conv2d():
    torch.manual_seed(103)
    arrays, cases = {}, []
    f = R_upfirdn2d.setup_filter([1, 3, 3, 1])
    idx = 0
    for demodulate, fused, up, use_noise in itertools.product([True, False], [True, False], [1, 2], [False, True]):
        x = torch.randn(2, 8, 8, 8, requires_grad=True)
        w = torch.randn(6, 8, 3, 3, requires_grad=True)
        s = (torch.randn(2, 8) + 1).requires_grad_(True)
        noise = torch.randn(2, 1, 8 * up, 8 * up) if use_noise else None
        y = R_net.modulated_conv2d(x=x, weight=w, styles=s, noise=noise, up=up, padding=1, resample_filter=f, demodulate=demodulate,
                                   flip_weight=(up == 1), fused_modconv=fused)
        dy = torch.randn_like(y)
        g = torch.autograd.grad((y * dy).sum(), [x, w, s], create_graph=True)
        # path-length style second order: d/d(styles) of |d y / d styles|^2, and R1 style: d/dw of |dx|^2
        pl = torch.autograd.grad(g[2].square().sum() + g[0].square().sum(), [w, s])
        key = f"c{idx}"
        arrays.update({f"{key}/x": npy(x), f"{key}/w": npy(w), f"{key}/s": npy(s), f"{key}/y": npy(y), f"{key}/dy": npy(dy),
                       f"{key}/dx": npy(g[0]), f"{key}/dw": npy(g[1]), f"{key}/ds": npy(g[2]), f"{key}/d2w": npy(pl[0]), f"{key}/d2s": npy(pl[1])})
        if use_noise:
            arrays[f"{key}/noise"] = npy(noise)
        cases.append(dict(key=key, demodulate=demodulate, fused_modconv=fused, up=up, use_noise=use_noise))
        idx += 1
    # fma
    a, b, c = torch.randn(2, 4, 5, 5, requires_grad=True), torch.randn(2, 4, 1, 1, requires_grad=True), torch.randn(2, 1, 5, 5, requires_grad=True)
    y = R_fma.fma(a, b, c)
    g = torch.autograd.grad(y.square().sum(), [a, b, c])
    arrays.update({"fma/a": npy(a), "fma/b": npy(b), "fma/c": npy(c), "fma/y": npy(y), "fma/da": npy(g[0]), "fma/db": npy(g[1]), "fma/dc": npy(g[2])})
    save("modulated_conv2d", arrays, dict(cases=cases))


# ---------------------------------------------------------------------------------------------------------------- networks

def state_arrays(module, prefix):
    return {f"{prefix}/{k}": npy(v) for k, v in module.state_dict().items()}


def gen_networks():
    """whole G / D at 16x16 (channel_base 256, channel_max 32): forward (noise_mode const), and one
    Gmain + Dmain + R1 gradient set with softplus losses."""
    for tag, g_arch, d_arch, c_dim, clamp in [("skip_resnet", "skip", "resnet", 0, None), ("orig_orig_c3_clamp", "orig", "orig", 3, 4.0),
                                               ("resnet_skip", "resnet", "skip", 0, None)]:
        torch.manual_seed(200)
        res, cb, cm = 16, 256, 32
        G = R_net.Generator(z_dim=16, c_dim=c_dim, w_dim=24, img_resolution=res, img_channels=3,
                            mapping_kwargs=dnnlib.EasyDict(num_layers=2),
                            synthesis_kwargs=dnnlib.EasyDict(channel_base=cb, channel_max=cm, architecture=g_arch, conv_clamp=clamp))
        D = R_net.Discriminator(c_dim=c_dim, img_resolution=res, img_channels=3, architecture=d_arch, channel_base=cb, channel_max=cm,
                                conv_clamp=clamp, mapping_kwargs=dnnlib.EasyDict(num_layers=2), epilogue_kwargs=dnnlib.EasyDict(mbstd_group_size=2))
        G.train(); D.train()
        with torch.no_grad():       # non-trivial noise strengths / biases so every term is exercised
            for name, p in list(G.named_parameters()) + list(D.named_parameters()):
                if name.endswith("noise_strength"):
                    p.fill_(0.3)
                if name.endswith(".bias") and "affine" not in name:
                    p.copy_(torch.randn_like(p) * 0.1)
        n = 4
        z = torch.randn(n, 16)
        c = torch.nn.functional.one_hot(torch.arange(n) % max(c_dim, 1), max(c_dim, 1)).float()[:, :c_dim]
        real = torch.randn(n, 3, res, res)
        arrays = dict(z=npy(z), c=npy(c), real=npy(real))
        arrays.update(state_arrays(G, "G")); arrays.update(state_arrays(D, "D"))
        ws = G.mapping(z, c, skip_w_avg_update=True)
        img = G.synthesis(ws, noise_mode='const', fused_modconv=False)
        img_fused = G.synthesis(ws, noise_mode='const', fused_modconv=True)
        logits = D(img, c)
        arrays.update(ws=npy(ws), img=npy(img), img_fused=npy(img_fused), logits=npy(logits))
        # training-step gradients (softplus, R1 gamma 0.5), noise_mode const so the draw is reproducible
        for p in G.parameters(): p.requires_grad_(True)
        for p in D.parameters(): p.requires_grad_(False)
        fake = G.synthesis(G.mapping(z, c, skip_w_avg_update=True), noise_mode='const', fused_modconv=False)
        loss_g = torch.nn.functional.softplus(-D(fake, c)).mean()
        loss_g.backward()
        arrays["loss_g"] = npy(loss_g)
        for name, p in G.named_parameters():
            arrays[f"gradG/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p))
        for p in G.parameters(): p.requires_grad_(False)
        for p in D.parameters(): p.requires_grad_(True)
        z2 = torch.randn(n, 16)
        arrays["z2"] = npy(z2)
        with torch.no_grad():
            fake = G.synthesis(G.mapping(z2, c, skip_w_avg_update=True), noise_mode='const', fused_modconv=False)
        real_in = real.clone().requires_grad_(True)
        real_logits = D(real_in, c)
        loss_d = torch.nn.functional.softplus(-real_logits).mean() + torch.nn.functional.softplus(D(fake, c)).mean()
        loss_d.backward(retain_graph=True)
        arrays["loss_d"] = npy(loss_d)
        for name, p in D.named_parameters():
            arrays[f"gradD/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p)); p.grad = None
        r1 = torch.autograd.grad(real_logits.sum(), real_in, create_graph=True)[0]
        pen = (r1.square().sum([1, 2, 3]) * (0.5 / 2)).mean()
        pen.backward()
        arrays["r1_penalty"] = npy(pen)
        for name, p in D.named_parameters():
            arrays[f"gradR1/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p))
        meta = dict(z_dim=16, c_dim=c_dim, w_dim=24, img_resolution=res, img_channels=3, channel_base=cb, channel_max=cm, mapping_layers=2,
                    g_architecture=g_arch, d_architecture=d_arch, conv_clamp=clamp, mbstd_group_size=2, r1_gamma=0.5)
        save(f"networks_{tag}", arrays, meta)


# ---------------------------------------------------------------------------------------------------------------- BigGAN

def _import_train_parts():
    """train_parts.* imports omegaconf only for the MISSING sentinel (utils.py:91); a two-line in-memory stand-in for that
    absent package is enough to import the reference's model classes.  Nothing of it is written anywhere."""
    import types
    if 'omegaconf' not in sys.modules:
        stub = types.ModuleType('omegaconf')
        stub.MISSING = '???'
        sys.modules['omegaconf'] = stub
    import train_parts.generators as RG
    import train_parts.discriminators as RD
    return RG, RD


def gen_biggan():
    import biggan.layers as RL
    arrays = {}
    # -- power iteration / SN
    torch.manual_seed(300)
    W = torch.randn(12, 20)
    u = torch.randn(1, 12)
    arrays.update({"pi/W": npy(W), "pi/u": npy(u)})
    svs, us, vs = RL.power_iteration(W, [u.clone()], update=False, eps=1e-12)
    arrays.update({"pi/sigma": npy(svs[0]), "pi/u_new": npy(us[0]), "pi/v": npy(vs[0])})
    conv = RL.SNConv2d(8, 12, 3, padding=1)
    conv.train()
    x = torch.randn(2, 8, 6, 6, requires_grad=True)
    arrays.update({f"snconv/sd/{k}": npy(v) for k, v in conv.state_dict().items()})
    y = conv(x)
    gx, gw = torch.autograd.grad(y.square().sum(), [x, conv.weight])
    arrays.update({"snconv/x": npy(x), "snconv/y_train": npy(y), "snconv/dx": npy(gx), "snconv/dw": npy(gw),
                   "snconv/u0_after": npy(conv.u0), "snconv/sv0_after": npy(conv.sv0)})
    conv.eval()
    arrays["snconv/y_eval"] = npy(conv(x))
    lin = RL.SNLinear(10, 6)
    lin.train()
    arrays.update({f"snlin/sd/{k}": npy(v) for k, v in lin.state_dict().items()})
    xl = torch.randn(4, 10)
    arrays.update({"snlin/x": npy(xl), "snlin/y": npy(lin(xl))})
    # -- attention
    att = RL.Attention(16)
    att.train()
    with torch.no_grad():
        att.gamma.fill_(0.7)
    arrays.update({f"att/sd/{k}": npy(v) for k, v in att.state_dict().items()})
    xa = torch.randn(2, 16, 8, 8, requires_grad=True)
    ya = att(xa)
    ga = torch.autograd.grad(ya.square().sum(), [xa] + list(att.parameters()), create_graph=True)
    arrays.update({"att/x": npy(xa), "att/y": npy(ya), "att/dx": npy(ga[0])})
    for (name, _), g_ in zip(att.named_parameters(), ga[1:]):
        arrays[f"att/grad/{name}"] = npy(g_)
    g2 = torch.autograd.grad(ga[0].square().sum(), xa)[0]          # R1-style second order through attention
    arrays["att/d2x"] = npy(g2)
    # -- batch norms
    emb = torch.nn.Embedding
    cc = RL.ccbn(6, 10, emb)
    cc.train()
    arrays.update({f"ccbn/sd/{k}": npy(v) for k, v in cc.state_dict().items()})
    xb = (torch.randn(4, 6, 5, 5) * 2 + 1).requires_grad_(True)
    yb_idx = torch.tensor([1, 3, 3, 7])
    yb = cc(xb, yb_idx)
    gb = torch.autograd.grad(yb.square().sum(), [xb, cc.gain.weight, cc.bias.weight])
    arrays.update({"ccbn/x": npy(xb), "ccbn/y_idx": yb_idx.numpy().astype(np.float32), "ccbn/y_train": npy(yb), "ccbn/dx": npy(gb[0]),
                   "ccbn/dgain": npy(gb[1]), "ccbn/dbias": npy(gb[2]), "ccbn/mean_after": npy(cc.stored_mean), "ccbn/var_after": npy(cc.stored_var)})
    cc.eval()
    arrays["ccbn/y_eval"] = npy(cc(xb, yb_idx))
    b = RL.bn(6)
    b.train()
    with torch.no_grad():
        b.gain.copy_(torch.rand(6) + 0.5); b.bias.copy_(torch.randn(6))
    arrays.update({f"bn/sd/{k}": npy(v) for k, v in b.state_dict().items()})
    ybn = b(xb)
    arrays.update({"bn/y_train": npy(ybn), "bn/mean_after": npy(b.stored_mean), "bn/var_after": npy(b.stored_var)})
    # synchronized-BN formula (sync_batchnorm/batchnorm.py:147-158) on explicit sums
    s1, s2, cnt = xb.detach().sum([0, 2, 3]), xb.detach().square().sum([0, 2, 3]), 4 * 25
    mean = s1 / cnt; sumvar = s2 - s1 * mean
    arrays.update({"syncbn/mean": npy(mean), "syncbn/inv_std": npy(torch.rsqrt(sumvar / cnt + 1e-5)), "syncbn/unbias_var": npy(sumvar / (cnt - 1))})
    save("biggan_layers", arrays, dict())

    # -- whole tiny BigGAN G / D at 32x32 (ch 8), attention in both, hinge losses
    RG, RD = _import_train_parts()
    torch.manual_seed(301)
    cfg = dict(G_ch=8, D_ch=8, z_dim=16, n_classes=10, img_resolution=32, G_attn='16', D_attn='16')
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        G = RG.generators['big_gan'](G_ch=8, z_dim=16, c_dim=10, img_resolution=32, G_shared=False, G_attn='16', G_init='N02', n_classes=10)
        D = RD.discriminators['big_gan'](D_ch=8, c_dim=10, img_resolution=32, D_attn='16', D_init='N02', n_classes=10)
    with torch.no_grad():
        for m in list(G.modules()) + list(D.modules()):
            if isinstance(m, RL.Attention):
                m.gamma.fill_(0.5)
        for name, p_ in list(G.named_parameters()) + list(D.named_parameters()):
            if p_.ndim > 1:
                p_.mul_(8.0)        # N02 init is tiny at this width; scale up so every path carries signal
    G.train(); D.train()
    arrays = {}
    arrays.update(state_arrays(G, "G")); arrays.update(state_arrays(D, "D"))
    n = 4
    z = torch.randn(n, 16)
    c = torch.nn.functional.one_hot(torch.tensor([0, 3, 3, 9]), 10).float()
    real = torch.randn(n, 3, 32, 32)
    arrays.update(z=npy(z), c=npy(c), real=npy(real))
    img = G(z, c)
    logits_fake = D(img, c)
    loss_g = -logits_fake.mean()
    gG = torch.autograd.grad(loss_g, list(G.parameters()), allow_unused=True, retain_graph=True)
    arrays.update(img=npy(img), logits_fake=npy(logits_fake), loss_g=npy(loss_g))
    for (name, p_), g_ in zip(G.named_parameters(), gG):
        arrays[f"gradG/{name}"] = npy(g_ if g_ is not None else torch.zeros_like(p_))
    arrays.update({f"G_after/{k}": npy(v) for k, v in G.state_dict().items() if ('u0' in k or 'sv0' in k or 'stored' in k)})
    arrays.update({f"D_after1/{k}": npy(v) for k, v in D.state_dict().items() if ('u0' in k or 'sv0' in k)})
    logits_real = D(real, c)
    loss_d = torch.nn.functional.relu(1 - logits_real).mean() + torch.nn.functional.relu(1 + D(img.detach(), c)).mean()
    gD = torch.autograd.grad(loss_d, list(D.parameters()), allow_unused=True)
    arrays.update(logits_real=npy(logits_real), loss_d=npy(loss_d))
    for (name, p_), g_ in zip(D.named_parameters(), gD):
        arrays[f"gradD/{name}"] = npy(g_ if g_ is not None else torch.zeros_like(p_))
    save("biggan_networks", arrays, cfg)

# ---------------------------------------------------------------------------------------------------------------- hybrid + host step

class _ConstNoise(torch.nn.Module):
    """harness wrapper: the loss code calls ``G_synthesis(ws)``, whose default noise mode draws fresh noise from the device's
    generator on every call; the fixtures pin the synthesis network with its registered constant noise instead"""

    def __init__(self, synthesis):
        super().__init__()
        self.synthesis = synthesis

    def forward(self, ws):
        return self.synthesis(ws, noise_mode='const')


def _build_sg2(RG, RD, g_kw, d_kw, seed):
    import copy
    torch.manual_seed(seed)
    g_kw, d_kw = copy.deepcopy(g_kw), copy.deepcopy(d_kw)
    ed = dnnlib.EasyDict
    G = RG.generators['sg2_classic'](**{k: (ed(v) if isinstance(v, dict) else v) for k, v in g_kw.items()})
    D = RD.discriminators['sg2_classic'](**{k: (ed(v) if isinstance(v, dict) else v) for k, v in d_kw.items()})
    with torch.no_grad():
        for m in list(G.modules()) + list(D.modules()):
            if type(m).__name__ == 'Attention':
                m.gamma.fill_(0.6)
        for name, p in list(G.named_parameters()) + list(D.named_parameters()):
            if name.endswith("noise_strength"):
                p.fill_(0.3)
            if name.endswith(".bias") and "affine" not in name:
                p.copy_(torch.randn_like(p) * 0.1)
    return G.train(), D.train()


def gen_sg2attent():
    """configs/sg2attent.yaml's architecture at 16x16: the reference's ``train_parts`` StyleGAN2 blocks with the non-local attention
    hook at the end of the block (generators.py:443-445, discriminators.py:297-299): G attention at every resolution, D attention
    at 16 (applied to its 8x8 output) and 8.  Forward + one Gmain / Dmain / R1 gradient set, as in gen_networks."""
    RG, RD = _import_train_parts()
    res, cb, cm = 16, 256, 32
    g_kw = dict(z_dim=16, c_dim=0, w_dim=24, img_resolution=res, img_channels=3, attentions=[16, 8, 4], mapping_kwargs=dict(num_layers=2),
                synthesis_kwargs=dict(channel_base=cb, channel_max=cm, block_kwargs=dict(architecture='skip', layer_kwargs=dict())))
    d_kw = dict(c_dim=0, img_resolution=res, img_channels=3, attentions=[16, 8], architecture='orig', channel_base=cb, channel_max=cm,
                block_kwargs=dict(), mapping_kwargs=dict(), epilogue_kwargs=dict(mbstd_group_size=2))
    G, D = _build_sg2(RG, RD, g_kw, d_kw, seed=400)
    n = 4
    z, z2, real = torch.randn(n, 16), torch.randn(n, 16), torch.randn(n, 3, res, res)
    c = torch.zeros(n, 0)
    arrays = dict(z=npy(z), z2=npy(z2), real=npy(real))
    arrays.update(state_arrays(G, "G")); arrays.update(state_arrays(D, "D"))
    for p in D.parameters(): p.requires_grad_(False)
    ws = G.mapping(z, c, skip_w_avg_update=True)
    img = G.synthesis(ws, noise_mode='const')
    logits = D(img, c)
    loss_g = torch.nn.functional.softplus(-logits).mean()
    loss_g.backward()
    arrays.update(ws=npy(ws), img=npy(img), logits=npy(logits), loss_g=npy(loss_g))
    for name, p in G.named_parameters():
        arrays[f"gradG/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p))
    arrays.update({f"G_after/{k}": npy(v) for k, v in G.state_dict().items() if k.endswith(('u0', 'sv0'))})      # power iteration ran once per SN layer
    for p in G.parameters(): p.requires_grad_(False)
    for p in D.parameters(): p.requires_grad_(True)
    with torch.no_grad():
        fake = G.synthesis(G.mapping(z2, c, skip_w_avg_update=True), noise_mode='const')
    real_in = real.clone().requires_grad_(True)
    real_logits = D(real_in, c)
    loss_d = torch.nn.functional.softplus(-real_logits).mean() + torch.nn.functional.softplus(D(fake, c)).mean()
    loss_d.backward(retain_graph=True)
    arrays.update(loss_d=npy(loss_d), real_logits=npy(real_logits))
    for name, p in D.named_parameters():
        arrays[f"gradD/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p)); p.grad = None
    r1 = torch.autograd.grad(real_logits.sum(), real_in, create_graph=True)[0]
    pen = (r1.square().sum([1, 2, 3]) * (0.5 / 2)).mean()
    pen.backward()
    arrays["r1_penalty"] = npy(pen)
    for name, p in D.named_parameters():
        arrays[f"gradR1/{name}"] = npy(p.grad if p.grad is not None else torch.zeros_like(p))
    save("sg2attent", arrays, dict(g_kwargs=g_kw, d_kwargs=d_kw, r1_gamma=0.5))


def _run_reference_schedule(cfg, tag):
    """Drive the REFERENCE's loss / regulariser objects (train_parts/losses_base.py SG2Loss, regularizations.py R1reg / PPLreg) through the
    iteration body of its training loop.  ``train_parts/trainers.py`` itself does not import here (wandb), so its loop is re-enacted
    from the description in SURVEY.md section 8(c): phase table of :601-633 (main / reg slots per network when the interval is positive,
    optimizer rescaled by interval / (interval + 1)), and per iteration (:711-765) latents for every slot, then for each slot that is
    due: clear gradients -> requires_grad on -> accumulation rounds of accumulate_gradients(gain = interval, sync on the last) ->
    requires_grad off -> nan_to_num of the gradients -> optimizer step; then the generator average and the image counter.
    Everything random is captured: latents and reals are stored per iteration, the path-length direction is recorded as it is drawn."""
    import copy
    RG, RD = _import_train_parts()
    import train_parts.losses_base as R_lb
    from stylegan2ada.torch_utils import training_stats as R_stats
    from stylegan2ada.torch_utils import misc as R_misc
    G, D = _build_sg2(RG, RD, cfg['g_kwargs'], cfg['d_kwargs'], seed=cfg['seed'])
    G.requires_grad_(False); D.requires_grad_(False)
    G_ema = copy.deepcopy(G).eval()
    arrays = {}
    arrays.update(state_arrays(G, "G0")); arrays.update(state_arrays(D, "D0"))
    loss = R_lb.losses_arch['sg2'](device=torch.device('cpu'), loss=cfg['loss'], gen_regs=cfg['gen_regs'], dis_regs=cfg['dis_regs'], D=D,
                                   G_mapping=G.mapping, G_synthesis=_ConstNoise(G.synthesis), style_mixing_prob=0)
    slots = []
    for name, module, interval in [('G', G, cfg['g_reg_interval']), ('D', D, cfg['d_reg_interval'])]:
        kw = dict(cfg['opt'])
        if interval <= 0:
            opt = torch.optim.Adam(module.parameters(), **kw)
            slots.append(dict(name=name + 'both', module=module, opt=opt, interval=1))
        else:
            ratio = interval / (interval + 1)
            kw['lr'] = kw['lr'] * ratio
            kw['betas'] = [b ** ratio for b in kw['betas']]
            opt = torch.optim.Adam(module.parameters(), **kw)
            slots.append(dict(name=name + 'main', module=module, opt=opt, interval=1))
            slots.append(dict(name=name + 'reg', module=module, opt=opt, interval=interval))
    B, b = cfg['batch'], cfg['batch_gpu']
    collector = R_stats.Collector(regex='.*')
    collector.update()          # forget whatever earlier generators of this process reported
    pl_noise_log = []
    real_randn_like = torch.randn_like

    def recording_randn_like(t, *a, **k):       # the only randn_like on this path is the path-length direction (regularizations.py:26)
        out = real_randn_like(t, *a, **k)
        pl_noise_log.append(out.clone())
        return out

    torch.manual_seed(cfg['seed'] + 1)
    cur_nimg = 0
    for it in range(cfg['iterations']):
        reals = torch.randn(B, 3, cfg['g_kwargs']['img_resolution'], cfg['g_kwargs']['img_resolution']).clamp(-1, 1)
        all_z = torch.randn(len(slots) * B, G.z_dim)
        arrays[f"it{it}/real"], arrays[f"it{it}/all_gen_z"] = npy(reals), npy(all_z)
        c = torch.zeros(B, 0)
        for slot, slot_z in zip(slots, all_z.split(B)):
            if it % slot['interval'] != 0:
                continue
            slot['opt'].zero_grad(set_to_none=True)
            slot['module'].requires_grad_(True)
            rounds = B // b
            torch.randn_like = recording_randn_like
            try:
                for r in range(rounds):
                    sl = slice(r * b, (r + 1) * b)
                    loss.accumulate_gradients(phase=slot['name'], real_img=reals[sl], real_c=c[sl], gen_z=slot_z[sl], gen_c=c[sl],
                                              sync=(r == rounds - 1), gain=slot['interval'])
            finally:
                torch.randn_like = real_randn_like
            slot['module'].requires_grad_(False)
            for p in slot['module'].parameters():
                if p.grad is not None:
                    R_misc.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)
            slot['opt'].step()
        ema_nimg = cfg['ema_kimg'] * 1000
        if cfg['ema_rampup'] is not None:
            ema_nimg = min(ema_nimg, cur_nimg * cfg['ema_rampup'])
        beta = 0.5 ** (B / max(ema_nimg, 1e-8))
        with torch.no_grad():
            for pe, p in zip(G_ema.parameters(), G.parameters()):
                pe.copy_(p.lerp(pe, beta))
            for be, b_ in zip(G_ema.buffers(), G.buffers()):
                be.copy_(b_)
        cur_nimg += B
        arrays.update(state_arrays(G, f"it{it}/G")); arrays.update(state_arrays(D, f"it{it}/D")); arrays.update(state_arrays(G_ema, f"it{it}/G_ema"))
    for i, t in enumerate(pl_noise_log):
        arrays[f"pl_noise/{i}"] = npy(t)
    if loss.gen_regs is not None:
        arrays["pl_mean"] = npy(loss.gen_regs[0].pl_mean)
    collector.update()
    stats = {name: dict(mean=float(collector.mean(name)), num=int(collector.num(name))) for name in collector.names() if name.startswith('Loss/')}
    save(f"host_step_{tag}", arrays, dict(cfg=cfg, slots=[dict(name=s_['name'], interval=s_['interval']) for s_ in slots], stats=stats,
                                           n_pl_noise=len(pl_noise_log)))


def gen_host_step():
    res, cb, cm = 16, 256, 32
    base_g = dict(z_dim=16, c_dim=0, w_dim=24, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2),
                  synthesis_kwargs=dict(channel_base=cb, channel_max=cm, block_kwargs=dict(architecture='skip', layer_kwargs=dict())))
    base_d = dict(c_dim=0, img_resolution=res, img_channels=3, architecture='orig', channel_base=cb, channel_max=cm, block_kwargs=dict(),
                  mapping_kwargs=dict(), epilogue_kwargs=dict(mbstd_group_size=2))
    # Adam's eps is raised from the configs' 1e-8 to 1e-4: with beta1 = 0 the first update is lr * g / (|g| + eps), which at eps = 1e-8
    # amplifies the last bit of any gradient near zero to a full +-lr step; the schedule, gains and rescaling under test do not depend on it
    opt = dict(lr=0.0025, betas=[0, 0.99], eps=1e-4)
    # (1) sg2ada.yaml's schedule: lazy R1 on D, NO generator regulariser but a positive g_reg_interval (idle Greg slot, rescaled G optimizer);
    #     two accumulation rounds per phase
    _run_reference_schedule(dict(g_kwargs=base_g, d_kwargs=base_d, seed=500, loss='softplus', gen_regs=[], dis_regs=[['r1', dict(r1_gamma=0.5)]],
                                 g_reg_interval=4, d_reg_interval=2, opt=opt, batch=4, batch_gpu=2, iterations=3, ema_kimg=0.02, ema_rampup=0.5), "r1")
    # (2) ffhq_sg2.yaml's schedule: path-length regulariser on G (weight 2, batch shrink 2) + R1 on a resnet D, mapping depth 3
    import copy
    g2, d2 = copy.deepcopy(base_g), copy.deepcopy(base_d)
    g2['mapping_kwargs']['num_layers'] = 3
    d2['architecture'] = 'resnet'
    _run_reference_schedule(dict(g_kwargs=g2, d_kwargs=d2, seed=510, loss='softplus',
                                 gen_regs=[['ppl', dict(pl_batch_shrink=2, pl_decay=0.01, pl_weight=2.)]], dis_regs=[['r1', dict(r1_gamma=1.0)]],
                                 g_reg_interval=2, d_reg_interval=2, opt=opt, batch=4, batch_gpu=4, iterations=3, ema_kimg=0.02, ema_rampup=None), "ppl")

# ---------------------------------------------------------------------------------------------------------------- quality metrics

def gen_metrics():
    """the reference's metric arithmetic (stylegan2ada/metrics: FeatureStats moments, compute_fid / compute_kid / compute_is / compute_pr) on
    synthetic feature sets: its two feature loops are replaced by functions that hand back FeatureStats filled with the fixture's features
    (the detectors are URL fetches), everything downstream is the reference's own code on CPU."""
    _import_train_parts()
    from stylegan2ada.metrics import metric_utils as R_mu
    from stylegan2ada.metrics import frechet_inception_distance as R_fid, kernel_inception_distance as R_kid
    from stylegan2ada.metrics import inception_score as R_is, precision_recall as R_pr
    rng = np.random.RandomState(600)
    F = 24
    mix = rng.randn(F, F) * 0.3 + np.eye(F)
    real = (rng.randn(320, F) @ mix + 0.5).astype(np.float32)
    gen = (rng.randn(256, F) @ (mix * 0.8) + rng.randn(F) * 0.3).astype(np.float32)
    logits = rng.randn(200, 10) * 2
    probs = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    arrays = dict(real=real, gen=gen, probs=probs)

    def stats_of(x, **kw):
        st = R_mu.FeatureStats(**kw)
        for part in np.array_split(x, 5):       # several appends, like the feature loops
            st.append(part)
        return st

    st = stats_of(real, capture_mean_cov=True, capture_all=True, max_items=300)     # max_items clips the last append
    mean, cov = st.get_mean_cov()
    arrays.update(stats_mean=mean, stats_cov=cov, stats_all=st.get_all(), stats_num=np.asarray(st.num_items))

    opts = R_mu.MetricOptions(num_gpus=1, rank=0, device=torch.device('cpu'), cache=False)
    current = {}

    def fake_dataset(opts=None, max_items=None, **kw):
        flags = {k: v for k, v in kw.items() if k in ('capture_all', 'capture_mean_cov')}
        x = current['real']
        return stats_of(x, max_items=min(len(x), max_items) if max_items is not None else len(x), **flags)

    def fake_generator(opts=None, max_items=None, **kw):
        flags = {k: v for k, v in kw.items() if k in ('capture_all', 'capture_mean_cov')}
        return stats_of(current['gen'], max_items=max_items, **flags)

    saved = R_mu.compute_feature_stats_for_dataset, R_mu.compute_feature_stats_for_generator
    R_mu.compute_feature_stats_for_dataset, R_mu.compute_feature_stats_for_generator = fake_dataset, fake_generator
    half = torch.Tensor.to
    try:
        current.update(real=real, gen=gen)
        arrays['fid'] = np.asarray(R_fid.compute_fid(opts, max_real=None, num_gen=256))
        arrays['fid_maxreal'] = np.asarray(R_fid.compute_fid(opts, max_real=200, num_gen=128))
        np.random.seed(601)
        arrays['kid'] = np.asarray(R_kid.compute_kid(opts, max_real=1000, num_gen=256, num_subsets=7, max_subset_size=100))
        current['gen'] = probs
        arrays['is_mean_std'] = np.asarray(R_is.compute_is(opts, num_gen=200, num_splits=4))
        current.update(real=real, gen=gen)
        # compute_pr passes a misspelt keyword (datasetname=, precision_recall.py:41), which the stand-in absorbs; it casts features to fp16,
        # which torch.cdist does not take on the CPU: keep them in fp32 for this CPU run (the cast is a device-memory saving, not arithmetic)
        torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] is torch.float16) else half(self, *a, **k)
        arrays['pr'] = np.asarray(R_pr.compute_pr(opts, max_real=320, num_gen=256, nhood_size=3, row_batch_size=100, col_batch_size=64))
    finally:
        torch.Tensor.to = half
        R_mu.compute_feature_stats_for_dataset, R_mu.compute_feature_stats_for_generator = saved
    save("metrics", arrays, dict(kid_seed=601, kid_subsets=7, kid_subset_size=100, is_splits=4, pr=dict(nhood_size=3, row_batch_size=100, col_batch_size=64)))

# ---------------------------------------------------------------------------------------------------------------- input pipeline

def gen_datasets():
    """the reference's ImageFolderDataset (train_parts/datasets.py:160-248) and InfiniteSampler (torch_utils/misc.py:109-140) over the
    deterministic PNG folder of tests/golden_util.make_image_folder: item order, flips, labels, pixels, sampler index streams"""
    import tempfile
    _import_train_parts()
    sys.path.insert(0, os.path.dirname(HERE))
    from golden_util import make_image_folder
    import train_parts.datasets as R_ds
    from stylegan2ada.torch_utils import misc as R_misc
    arrays, cases = {}, []
    with tempfile.TemporaryDirectory() as d:
        root = make_image_folder(os.path.join(d, "data"))
        zroot = make_image_folder(os.path.join(d, "dataz"), as_zip=True)
        for idx, (src, kw) in enumerate([(root, dict(use_labels=True)), (zroot, dict(use_labels=True)), (root, dict()),
                                         (root, dict(use_labels=True, max_size=5, random_seed=3)), (root, dict(use_labels=True, max_size=6, xflip=True, random_seed=1)),
                                         (root, dict(xflip=True))]):
            ds = R_ds.datasets['image_folder'](path=src, **kw)
            imgs = np.stack([ds[i][0] for i in range(len(ds))])
            labs = np.stack([ds[i][1] for i in range(len(ds))])
            det = [ds.get_details(i) for i in range(len(ds))]
            arrays.update({f"c{idx}/images": imgs, f"c{idx}/labels": labs.astype(np.float32), f"c{idx}/raw_idx": np.asarray([int(x.raw_idx) for x in det]),
                           f"c{idx}/xflip": np.asarray([int(x.xflip) for x in det]),
                           f"c{idx}/get_label": np.stack([ds.get_label(i) for i in range(len(ds))]).astype(np.float32)})
            cases.append(dict(key=f"c{idx}", zip=src.endswith('.zip'), kwargs=kw, len=len(ds), image_shape=list(ds.image_shape), label_shape=list(ds.label_shape),
                              label_dim=int(ds.label_dim), has_labels=bool(ds.has_labels), has_onehot_labels=bool(ds.has_onehot_labels), resolution=int(ds.resolution),
                              num_channels=int(ds.num_channels), name=ds.name))
            ds.close()
        ds = R_ds.datasets['image_folder'](path=root)
        samplers = []
        # torch 2.10's Sampler.__init__ no longer takes the data source the reference passes up (misc.py:115, written for torch 1.7): accept it
        torch.utils.data.Sampler.__init__ = lambda self, *a, **k: None
        for j, kw in enumerate([dict(rank=0, num_replicas=1, seed=0), dict(rank=1, num_replicas=2, seed=5), dict(rank=0, num_replicas=2, seed=5),
                                dict(rank=2, num_replicas=4, seed=1, window_size=0.2), dict(rank=0, num_replicas=1, shuffle=False)]):
            it = iter(R_misc.InfiniteSampler(ds, **kw))
            arrays[f"sampler{j}"] = np.asarray([int(next(it)) for _ in range(60)])
            samplers.append(kw)
    save("datasets", arrays, dict(cases=cases, samplers=samplers))


# ---------------------------------------------------------------------------------------------------------------- ADA pipe

AUG_SPECS = {   # stylegan2ada/train.py:271-283
    "blit": dict(xflip=1, rotate90=1, xint=1),
    "geom": dict(scale=1, rotate=1, aniso=1, xfrac=1),
    "color": dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    "filter": dict(imgfilter=1),
    "noise": dict(noise=1),
    "cutout": dict(cutout=1),
}
AUG_SPECS["bgc"] = {**AUG_SPECS["blit"], **AUG_SPECS["geom"], **AUG_SPECS["color"]}
AUG_SPECS["bgcfnc"] = {**AUG_SPECS["bgc"], **AUG_SPECS["filter"], **AUG_SPECS["noise"], **AUG_SPECS["cutout"]}


def gen_augment():
    """the reference AugmentPipe (stylegan2ada/training/augment.py == train_parts/augmentations.py:121-433) on CPU: per case the
    seed that precedes the call, the input images, the output, and d(sum(out * weight))/d(images)."""
    from stylegan2ada.training import augment as R_aug
    arrays, cases = {}, []
    torch.manual_seed(104)
    inputs = {"rgb": torch.randn(4, 3, 32, 32).clamp(-1, 1), "gray": torch.randn(3, 1, 24, 28).clamp(-1, 1)}
    for k, v in inputs.items():
        arrays[f"x/{k}"] = npy(v)
    idx = 0
    for spec, p, dbg, which in [("blit", 1.0, None, "rgb"), ("geom", 1.0, None, "rgb"), ("color", 1.0, None, "rgb"), ("filter", 1.0, None, "rgb"),
                                ("noise", 1.0, None, "rgb"), ("cutout", 1.0, None, "rgb"), ("bgc", 1.0, None, "rgb"), ("bgc", 0.6, None, "rgb"),
                                ("bgc", 0.0, None, "rgb"), ("bgcfnc", 0.8, None, "rgb"), ("bgcfnc", 1.0, None, "gray"), ("bgc", 0.7, None, "gray"),
                                ("bgcfnc", 1.0, 0.3, "rgb"), ("bgc", 1.0, 0.85, "rgb")]:
        pipe = R_aug.AugmentPipe(**AUG_SPECS[spec])
        pipe.p.copy_(torch.as_tensor(p))
        x = inputs[which].clone().requires_grad_(True)
        seed = 5000 + idx
        torch.manual_seed(seed)
        y = pipe(x, debug_percentile=dbg)
        wgt = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1))
        (dx,) = torch.autograd.grad((y * wgt).sum(), x)
        arrays[f"y/{idx}"], arrays[f"dx/{idx}"], arrays[f"w/{idx}"] = npy(y), npy(dx), npy(wgt)
        cases.append(dict(idx=idx, spec=spec, kwargs=AUG_SPECS[spec], p=p, debug_percentile=dbg, input=which, seed=seed))
        idx += 1
    arrays["Hz_fbank"] = npy(R_aug.AugmentPipe(imgfilter=1).Hz_fbank)
    arrays["Hz_geom"] = npy(R_aug.AugmentPipe().Hz_geom)
    save("augment", arrays, dict(cases=cases))


if __name__ == "__main__":
    which = sys.argv[1:] or ["upfirdn2d", "bias_act", "conv2d_resample", "modulated_conv2d", "networks", "biggan", "augment", "sg2attent", "host_step", "metrics", "datasets"]
    for name in which:
        globals()["gen_" + name]()
