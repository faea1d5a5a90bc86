"""GPU parity of the BigGAN layers / models (spectral norm, attention, batch norms, residual blocks, whole G / D) against the
golden vectors captured from the reference's biggan/layers.py and train_parts models.

Tolerances: fp32 tensors throughout (six bf16 MFMA passes for the convolutions, exact-fp32 MFMA for attention):
2e-4 of the tensor's max magnitude for activations, 2e-3 for gradients, 1e-2 for the second-order attention gradient."""
import pytest
import torch
import torch.nn.functional as F

import style_big_gan_amd  # noqa: F401
from golden_util import Golden, max_rel
from style_big_gan_amd.biggan import layers as L
from style_big_gan_amd.train_parts import discriminators as PD
from style_big_gan_amd.train_parts import generators as PG

pytestmark = pytest.mark.gpu


def test_power_iteration_kernel(dev):
    g = Golden("biggan_layers")
    W = g.t("pi/W").to(dev).requires_grad_(True)
    sigma, u_new, v = L._SpectralSigma.apply(W, g.t("pi/u").to(dev), 1e-12)
    assert max_rel(sigma, g.t("pi/sigma")) < 1e-5 and max_rel(u_new, g.t("pi/u_new")) < 1e-5 and max_rel(v, g.t("pi/v")) < 1e-5
    (gw,) = torch.autograd.grad(sigma, W)
    assert max_rel(gw, torch.outer(g.t("pi/u_new")[0], g.t("pi/v")[0])) < 1e-5
    # a large matrix (the biggest BigGAN layer is [1024, 9216]) against torch on the device
    torch.manual_seed(0)
    Wb = torch.randn(1024, 4608, device=dev); ub = torch.randn(1, 1024, device=dev)
    s, un, vv = L._SpectralSigma.apply(Wb, ub, 1e-12)
    v_ref = F.normalize(ub @ Wb, eps=1e-12); u_ref = F.normalize(v_ref @ Wb.t(), eps=1e-12); s_ref = (v_ref @ Wb.t() @ u_ref.t()).squeeze()
    assert max_rel(s, s_ref) < 1e-4 and max_rel(un, u_ref) < 1e-4 and max_rel(vv, v_ref) < 1e-4


def test_sn_conv_and_linear(dev):
    g = Golden("biggan_layers")
    conv = L.SNConv2d(8, 12, 3, padding=1)
    conv.load_state_dict(g.state_dict("snconv/sd"))
    conv = conv.to(dev).train()
    x = g.t("snconv/x").to(dev).requires_grad_(True)
    y = conv(x)
    assert max_rel(y, g.t("snconv/y_train")) < 2e-4
    gx, gw = torch.autograd.grad(y.square().sum(), [x, conv.weight])
    assert max_rel(gx, g.t("snconv/dx")) < 2e-3 and max_rel(gw, g.t("snconv/dw")) < 2e-3
    assert max_rel(conv.u0, g.t("snconv/u0_after")) < 1e-5 and max_rel(conv.sv0, g.t("snconv/sv0_after")) < 1e-5
    conv.eval()
    assert max_rel(conv(x), g.t("snconv/y_eval")) < 2e-4
    lin = L.SNLinear(10, 6)
    lin.load_state_dict(g.state_dict("snlin/sd"))
    lin = lin.to(dev).train()
    assert max_rel(lin(g.t("snlin/x").to(dev)), g.t("snlin/y")) < 1e-5


def test_attention(dev):
    g = Golden("biggan_layers")
    att = L.Attention(16)
    att.load_state_dict(g.state_dict("att/sd"))
    att = att.to(dev).train()
    x = g.t("att/x").to(dev).requires_grad_(True)
    y = att(x)
    assert max_rel(y, g.t("att/y")) < 2e-4
    params = dict(att.named_parameters())
    names = [k[len("att/grad/"):] for k in g.keys("att/grad/")]
    grads = torch.autograd.grad(y.square().sum(), [x] + [params[k] for k in names], create_graph=True)
    assert max_rel(grads[0], g.t("att/dx")) < 2e-3
    for k, got in zip(names, grads[1:]):
        assert max_rel(got, g.t("att/grad/" + k)) < 2e-3, k
    assert max_rel(torch.autograd.grad(grads[0].square().sum(), x)[0], g.t("att/d2x")) < 1e-2


def test_attention_core_shapes(dev):
    torch.manual_seed(1)
    for (n, q, m, d, dv) in [(2, 64, 16, 4, 16), (3, 1024, 256, 16, 64), (2, 256, 64, 64, 256), (1, 1024, 256, 64, 256)]:
        t, p, gg = torch.randn(n, q, d, device=dev), torch.randn(n, m, d, device=dev), torch.randn(n, m, dv, device=dev)
        assert style_big_gan_amd._lib.load().sbg_attention_supported(q, m, d, dv)
        out = L.attention_core(t, p, gg)
        ref = L._attention_reference(t.double(), p.double(), gg.double()).float()
        assert max_rel(out, ref) < 1e-5, (n, q, m, d, dv)
    # spiky logits (|logit| ~ 1e3: exercises the max-subtraction; fp32 logits carry ~1e-4 absolute error at that magnitude)
    t = torch.randn(1, 64, 16, device=dev) * 30; p = torch.randn(1, 32, 16, device=dev) * 30; gg = torch.randn(1, 32, 16, device=dev)
    assert max_rel(L.attention_core(t, p, gg), L._attention_reference(t.double(), p.double(), gg.double()).float()) < 1e-3


def test_attention_backward_kernels(dev):
    """sbg_attention_bwd (two recompute passes: per query tile -> dtheta + row statistics, per key tile -> dphi, dg) against float64 autograd
    of the reference's bmm / softmax / bmm on the CPU: 1e-5 of each gradient's max magnitude (exact-fp32 MFMA).  Shapes: the attention
    blocks of big_gan.yaml (C = 128 at 32x32 -> D 16, DV 64), sg2attent.yaml (C = 512 at 32 / 16 / 8 / 4), the 16x16 fixtures (D 4), and a
    D that is not a multiple of 16.  Then the same gradients with create_graph=True (the library composition) and a second-order check."""
    torch.manual_seed(2)
    lib = style_big_gan_amd._lib.load()
    for (n, q, m, d, dv) in [(2, 64, 16, 4, 16), (3, 1024, 256, 16, 64), (2, 1024, 256, 64, 256), (2, 256, 64, 64, 256), (2, 64, 16, 64, 256), (2, 256, 64, 24, 32)]:
        assert lib.sbg_attention_bwd_supported(q, m, d, dv), (q, m, d, dv)
        ins = [torch.randn(n, q, d), torch.randn(n, m, d), torch.randn(n, m, dv)]
        dout = torch.randn(n, q, dv)
        ref_in = [t.double().requires_grad_(True) for t in ins]
        ref = torch.autograd.grad(L._attention_reference(*ref_in), ref_in, dout.double())
        dev_in = [t.to(dev).requires_grad_(True) for t in ins]
        got = torch.autograd.grad(L.attention_core(*dev_in), dev_in, dout.to(dev))
        for name, a, b in zip(("dtheta", "dphi", "dg"), got, ref):
            assert max_rel(a, b.float()) < 1e-5, (name, n, q, m, d, dv, max_rel(a, b.float()))
        # twice: the kernels are deterministic (sums over queries stay inside a wave, fixed order)
        again = torch.autograd.grad(L.attention_core(*dev_in), dev_in, dout.to(dev))
        assert all(torch.equal(a, b) for a, b in zip(got, again))
    # second order: gradient of |dtheta|^2 + |dg|^2 w.r.t. all three inputs and dout
    ins = [torch.randn(2, 64, 16), torch.randn(2, 16, 16), torch.randn(2, 16, 32)]
    dout = torch.randn(2, 64, 32)

    def second(tensors, dy, fn):
        g1 = torch.autograd.grad(fn(*tensors), tensors, dy, create_graph=True)
        return torch.autograd.grad(g1[0].square().sum() + g1[2].square().sum(), list(tensors) + [dy])

    ref = second([t.double().requires_grad_(True) for t in ins], dout.double().requires_grad_(True), L._attention_reference)
    got = second([t.to(dev).requires_grad_(True) for t in ins], dout.to(dev).requires_grad_(True), L.attention_core)
    for a, b in zip(got, ref):
        assert max_rel(a, b.float()) < 1e-4


def test_batch_norms(dev):
    g = Golden("biggan_layers")
    cc = L.ccbn(6, 10, torch.nn.Embedding)
    cc.load_state_dict(g.state_dict("ccbn/sd"))
    cc = cc.to(dev).train()
    x = g.t("ccbn/x").to(dev).requires_grad_(True)
    yi = g.t("ccbn/y_idx").long().to(dev)
    y = cc(x, yi)
    assert max_rel(y, g.t("ccbn/y_train")) < 1e-5
    gr = torch.autograd.grad(y.square().sum(), [x, cc.gain.weight, cc.bias.weight])
    assert max_rel(gr[0], g.t("ccbn/dx")) < 1e-4 and max_rel(gr[1], g.t("ccbn/dgain")) < 1e-4 and max_rel(gr[2], g.t("ccbn/dbias")) < 1e-4
    assert max_rel(cc.stored_mean, g.t("ccbn/mean_after")) < 1e-5 and max_rel(cc.stored_var, g.t("ccbn/var_after")) < 1e-5
    cc.eval()
    assert max_rel(cc(x, yi), g.t("ccbn/y_eval")) < 1e-5
    b = L.bn(6)
    b.load_state_dict(g.state_dict("bn/sd"))
    b = b.to(dev).train()
    assert max_rel(b(x), g.t("bn/y_train")) < 1e-5
    assert max_rel(b.stored_mean, g.t("bn/mean_after")) < 1e-5 and max_rel(b.stored_var, g.t("bn/var_after")) < 1e-5
    # the synchronized-statistics formula used when cross_replica=True (single rank here)
    mean, var, unb, cnt = L.batch_stats(x.detach(), cross_replica=True)
    assert max_rel(mean, g.t("syncbn/mean")) < 1e-5 and max_rel(torch.rsqrt(var + 1e-5), g.t("syncbn/inv_std")) < 1e-5 and max_rel(unb, g.t("syncbn/unbias_var")) < 1e-5
    # channel-minor bf16 activations take the vector kernels
    xb = torch.randn(4, 16, 8, 8, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    m2, v2, _, _ = L.batch_stats(xb)
    assert max_rel(m2, xb.float().mean([0, 2, 3])) < 1e-4 and max_rel(v2, xb.float().var([0, 2, 3], unbiased=False)) < 1e-3


def test_norm_style_variants_and_mybn(dev):
    """ccbn norm_style 'in' / 'gn' / 'nonorm' and the `mybn` path (reference layers.py:212-253, 257-268, 310-325) on the device against the
    reference's own formulas evaluated by torch on the CPU: outputs, input gradients, running buffers (instance norm: batch mean of the
    per-instance statistics, unbiased variance; myBN: BIASED variance; standing statistics), train and eval mode."""
    torch.manual_seed(4)
    x = torch.randn(4, 32, 12, 10); y = torch.randn(4, 7); dy = torch.randn(4, 32, 12, 10)

    def run(mod, eval_mode=False):
        mod = mod.to(dev)
        mod.train(not eval_mode)
        xg = x.to(dev).requires_grad_(True)
        out = mod(xg, y.to(dev))
        (dx,) = torch.autograd.grad((out * dy.to(dev)).sum(), xg)
        return out.detach().cpu(), dx.cpu()

    for style in ("in", "gn", "gn_grp_4", "gn_ch_8", "nonorm"):
        m = L.ccbn(32, 7, torch.nn.Linear, norm_style=style)
        gain = (1 + m.gain(y)).view(4, -1, 1, 1).detach(); bias = m.bias(y).view(4, -1, 1, 1).detach()
        xr = x.clone().requires_grad_(True)
        rm, rv = torch.zeros(32), torch.ones(32)
        if style == "in":
            ref = F.instance_norm(xr, rm, rv, None, None, True, 0.1, 1e-5) * gain + bias
        elif style == "nonorm":
            ref = xr * gain + bias
        else:
            ref = F.group_norm(xr, L.group_norm_groups(style, 32)) * gain + bias
        (dxr,) = torch.autograd.grad((ref * dy).sum(), xr)
        out, dx = run(m)
        assert max_rel(out, ref) < 1e-5 and max_rel(dx, dxr) < 1e-4, style
        if style == "in":
            assert max_rel(m.stored_mean, rm) < 1e-5 and max_rel(m.stored_var, rv) < 1e-5
            out_e, _ = run(m, eval_mode=True)
            assert max_rel(out_e, F.instance_norm(x, rm, rv, None, None, False, 0.1, 1e-5) * gain + bias) < 1e-5
    with pytest.raises(NotImplementedError):
        L.ccbn(32, 7, torch.nn.Linear, norm_style="layer")
    # myBN: mean-of-squares statistics, running average of the biased variance, standing statistics
    m = L.bn(32, mybn=True)
    assert sorted(m.state_dict()) == ["bias", "bn.accumulation_counter", "bn.stored_mean", "bn.stored_var", "gain"]
    mean = x.mean([0, 2, 3]); var = (x ** 2).mean([0, 2, 3]) - mean ** 2
    out, _ = run(m)
    assert max_rel(out, (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var + 1e-5).view(1, -1, 1, 1)) < 1e-5
    assert max_rel(m.bn.stored_mean, 0.1 * mean) < 1e-5 and max_rel(m.bn.stored_var, 0.9 + 0.1 * var) < 1e-5
    m.bn.reset_stats(); m.bn.accumulate_standing = True
    run(m); run(m)
    assert float(m.bn.accumulation_counter) == 2.0 and max_rel(m.bn.stored_var, 2 * var) < 1e-5
    out_e, _ = run(m, eval_mode=True)
    assert max_rel(out_e, (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var + 1e-5).view(1, -1, 1, 1)) < 1e-5
    mc = L.ccbn(32, 7, torch.nn.Linear, mybn=True)
    gain = (1 + mc.gain(y)).view(4, -1, 1, 1).detach(); bias = mc.bias(y).view(4, -1, 1, 1).detach()
    out, _ = run(mc)
    assert max_rel(out, (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var + 1e-5).view(1, -1, 1, 1) * gain + bias) < 1e-5


def test_pooling_and_upsampling_as_fir(dev):
    torch.manual_seed(2)
    x = torch.randn(2, 8, 6, 10, device=dev)
    assert max_rel(L.nearest_upsample2x(x), F.interpolate(x, scale_factor=2)) < 1e-6
    assert max_rel(L.avg_pool2x(x), F.avg_pool2d(x, 2)) < 1e-6


def test_biggan_networks(dev):
    g = Golden("biggan_networks")
    G = PG.generators["big_gan"](G_ch=8, z_dim=16, c_dim=10, img_resolution=32, G_shared=False, G_attn="16", G_init="N02", n_classes=10)
    D = PD.discriminators["big_gan"](D_ch=8, c_dim=10, img_resolution=32, D_attn="16", D_init="N02", n_classes=10)
    G.load_state_dict(g.state_dict("G"), strict=True)       # state_dicts interchange with the reference
    D.load_state_dict(g.state_dict("D"), strict=True)
    G, D = G.to(dev).train(), D.to(dev).train()
    z, c, real = g.t("z").to(dev), g.t("c").to(dev), g.t("real").to(dev)
    img = G(z, c)
    assert max_rel(img, g.t("img")) < 5e-4
    lf = D(img, c)
    assert max_rel(lf, g.t("logits_fake")) < 2e-3
    gnames = [k for k, _ in G.named_parameters()]
    gg = torch.autograd.grad(-lf.mean(), list(G.parameters()), allow_unused=True, retain_graph=True)
    scale = max(float(g.t("gradG/" + k).abs().max()) for k in gnames)
    bad = []
    for k, got in zip(gnames, gg):
        ref = g.t("gradG/" + k)
        got = got if got is not None else torch.zeros_like(ref, device=dev)
        if max_rel(got, ref) >= 5e-3 and float(ref.abs().max()) >= 1e-4 * scale:
            bad.append((k, max_rel(got, ref)))
    assert not bad, bad
    for k in g.keys("G_after/"):
        assert max_rel(G.state_dict()[k[len("G_after/"):]], g.t(k)) < 1e-3, k
    lr = D(real, c)
    assert max_rel(lr, g.t("logits_real")) < 2e-3
    loss_d = F.relu(1 - lr).mean() + F.relu(1 + D(img.detach(), c)).mean()
    assert abs(float(loss_d) - float(g.t("loss_d"))) < 2e-3
    dnames = [k for k, _ in D.named_parameters()]
    dg = torch.autograd.grad(loss_d, list(D.parameters()), allow_unused=True)
    scale = max(float(g.t("gradD/" + k).abs().max()) for k in dnames)
    bad = []
    for k, got in zip(dnames, dg):
        ref = g.t("gradD/" + k)
        got = got if got is not None else torch.zeros_like(ref, device=dev)
        if max_rel(got, ref) >= 5e-3 and float(ref.abs().max()) >= 1e-4 * scale:
            bad.append((k, max_rel(got, ref)))
    assert not bad, bad
