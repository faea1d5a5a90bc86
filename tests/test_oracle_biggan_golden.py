"""CPU: oracle/biggan.py replayed against golden vectors captured from the reference's biggan/layers.py and
train_parts BigGAN generator / discriminator (tests/golden/biggan_*.npz).  Tolerance 2e-5 of the tensor's max."""
import torch
import torch.nn.functional as F

from golden_util import Golden, max_rel
from oracle import biggan as OB

TOL = 2e-5


def _sd(g, prefix):
    return g.state_dict(prefix)


def test_power_iteration_and_sn_layers():
    g = Golden("biggan_layers")
    sigma, u_new, v = OB.power_iteration(g.t("pi/W"), g.t("pi/u"))
    assert max_rel(sigma, g.t("pi/sigma")) < TOL and max_rel(u_new, g.t("pi/u_new")) < TOL and max_rel(v, g.t("pi/v")) < TOL
    sd = {"c." + k: v for k, v in _sd(g, "snconv/sd").items()}
    sd["c.weight"].requires_grad_(True)
    x = g.t("snconv/x").requires_grad_(True)
    upd = {}
    y = OB.sn_conv(sd, "c", x, 1, True, upd)
    assert max_rel(y, g.t("snconv/y_train")) < TOL
    gx, gw = torch.autograd.grad(y.square().sum(), [x, sd["c.weight"]])
    assert max_rel(gx, g.t("snconv/dx")) < TOL and max_rel(gw, g.t("snconv/dw")) < 1e-4
    assert max_rel(upd["c.u0"], g.t("snconv/u0_after")) < TOL and max_rel(upd["c.sv0"], g.t("snconv/sv0_after")) < TOL
    sd2 = dict(sd); sd2.update(upd)
    assert max_rel(OB.sn_conv(sd2, "c", x, 1, False, None), g.t("snconv/y_eval")) < TOL
    sl = {"l." + k: v for k, v in _sd(g, "snlin/sd").items()}
    assert max_rel(OB.sn_linear(sl, "l", g.t("snlin/x"), True, {}), g.t("snlin/y")) < TOL


def test_attention():
    g = Golden("biggan_layers")
    sd = {"a." + k: v.clone().requires_grad_(v.is_floating_point() and "u0" not in k and "sv0" not in k) for k, v in _sd(g, "att/sd").items()}
    x = g.t("att/x").requires_grad_(True)
    y = OB.attention(sd, "a", x, True, {})
    assert max_rel(y, g.t("att/y")) < TOL
    names = [k for k in g.keys("att/grad/")]
    grads = torch.autograd.grad(y.square().sum(), [x] + [sd["a." + k[len("att/grad/"):]] for k in names], create_graph=True)
    assert max_rel(grads[0], g.t("att/dx")) < 1e-4
    for k, got in zip(names, grads[1:]):
        assert max_rel(got, g.t(k)) < 1e-4, k
    assert max_rel(torch.autograd.grad(grads[0].square().sum(), x)[0], g.t("att/d2x")) < 1e-3


def test_batch_norms():
    g = Golden("biggan_layers")
    sd = {"b." + k: v.clone() for k, v in _sd(g, "ccbn/sd").items()}
    for k in ["b.gain.weight", "b.bias.weight"]:
        sd[k].requires_grad_(True)
    x = g.t("ccbn/x").requires_grad_(True)
    yi = g.t("ccbn/y_idx").long()
    upd = {}
    y = OB.ccbn(sd, "b", x, yi, True, upd)
    assert max_rel(y, g.t("ccbn/y_train")) < TOL
    gr = torch.autograd.grad(y.square().sum(), [x, sd["b.gain.weight"], sd["b.bias.weight"]])
    assert max_rel(gr[0], g.t("ccbn/dx")) < 1e-4 and max_rel(gr[1], g.t("ccbn/dgain")) < 1e-4 and max_rel(gr[2], g.t("ccbn/dbias")) < 1e-4
    assert max_rel(upd["b.stored_mean"], g.t("ccbn/mean_after")) < TOL and max_rel(upd["b.stored_var"], g.t("ccbn/var_after")) < TOL
    sd2 = {k: v.detach() for k, v in sd.items()}; sd2.update(upd)
    assert max_rel(OB.ccbn(sd2, "b", x, yi, False, None), g.t("ccbn/y_eval")) < TOL
    sb = {"n." + k: v for k, v in _sd(g, "bn/sd").items()}
    upd = {}
    assert max_rel(OB.bn(sb, "n", x, True, upd), g.t("bn/y_train")) < TOL
    assert max_rel(upd["n.stored_mean"], g.t("bn/mean_after")) < TOL and max_rel(upd["n.stored_var"], g.t("bn/var_after")) < TOL
    xd = x.detach()
    mean, inv_std, unb = OB.synchronized_stats(xd.sum([0, 2, 3]), xd.square().sum([0, 2, 3]), 100)
    assert max_rel(mean, g.t("syncbn/mean")) < TOL and max_rel(inv_std, g.t("syncbn/inv_std")) < TOL and max_rel(unb, g.t("syncbn/unbias_var")) < TOL


def test_biggan_networks():
    g = Golden("biggan_networks")
    gsd = {k: v.clone().requires_grad_(v.is_floating_point() and not any(t in k for t in ("u0", "sv0", "stored_"))) for k, v in _sd(g, "G").items()}
    dsd = {k: v.clone().requires_grad_(v.is_floating_point() and not any(t in k for t in ("u0", "sv0"))) for k, v in _sd(g, "D").items()}
    z, c, real = g.t("z"), g.t("c"), g.t("real")
    gu, du = {}, {}
    img = OB.generator(gsd, z, c, True, gu)
    assert max_rel(img, g.t("img")) < TOL
    lf = OB.discriminator(dsd, img, c, OB.D_DOWNSAMPLE[32], True, du)
    assert max_rel(lf, g.t("logits_fake")) < 1e-4
    names = [k for k, v in gsd.items() if v.requires_grad]
    gg = torch.autograd.grad(-lf.mean(), [gsd[k] for k in names], allow_unused=True, retain_graph=True)
    scale = max(float(g.t("gradG/" + k).abs().max()) for k in names)
    for k, got in zip(names, gg):
        ref = g.t("gradG/" + k)
        got = got if got is not None else torch.zeros_like(ref)
        # a conv bias in front of a batch norm has an exactly-zero gradient: what both sides hold there is rounding noise
        assert max_rel(got, ref) < 2e-4 or float(ref.abs().max()) < 1e-5 * scale, k
    for k in g.keys("G_after/"):
        assert max_rel(gu[k[len("G_after/"):]], g.t(k)) < 1e-4, k
    dsd1 = dict(dsd); dsd1.update({k: v for k, v in du.items()})
    du2 = {}
    lr = OB.discriminator(dsd1, real, c, OB.D_DOWNSAMPLE[32], True, du2)
    assert max_rel(lr, g.t("logits_real")) < 1e-4
    dsd2 = dict(dsd1); dsd2.update(du2)
    loss_d = F.relu(1 - lr).mean() + F.relu(1 + OB.discriminator(dsd2, img.detach(), c, OB.D_DOWNSAMPLE[32], True, {})).mean()
    assert abs(float(loss_d) - float(g.t("loss_d"))) < 1e-4
    dn = [k for k, v in dsd.items() if v.requires_grad]
    dg = torch.autograd.grad(loss_d, [dsd[k] for k in dn], allow_unused=True)
    scale = max(float(g.t("gradD/" + k).abs().max()) for k in dn)
    for k, got in zip(dn, dg):
        ref = g.t("gradD/" + k)
        got = got if got is not None else torch.zeros_like(ref)
        assert max_rel(got, ref) < 5e-4 or float(ref.abs().max()) < 1e-5 * scale, k
