"""CPU: the oracle (oracle/) replayed against the golden vectors captured from the reference.

This is the parity pin of the oracle: the reference ships no tests of its own, so these fixtures -- outputs of the
reference's eager CPU path on seeded inputs -- are what the oracle is held to.  Tolerance: 1e-5 relative to the
tensor's max magnitude (both sides are fp32 CPU PyTorch; only the order of a few reductions differs).
"""
import pytest
import torch

from golden_util import Golden, max_rel
from oracle import networks as ON
from oracle import ops as O

TOL = 1e-5


def test_upfirdn2d_golden():
    g = Golden("upfirdn2d")
    assert len(g.meta["cases"]) > 300
    for case in g.meta["cases"]:
        k = case["key"]
        f = g.t("f/" + case["filter"]) if ("f/" + case["filter"]) in g else None
        x = g.t(k + "/x").requires_grad_(True)
        y = O.upfirdn2d(x, f, up=case["up"], down=case["down"], padding=case["padding"], flip_filter=case["flip_filter"], gain=case["gain"])
        assert max_rel(y, g.t(k + "/y")) < TOL, case
        dy = g.t(k + "/dy").requires_grad_(True)
        dx = torch.autograd.grad((y * dy).sum(), x, create_graph=True)[0]
        assert max_rel(dx, g.t(k + "/dx")) < TOL, case
        ddy = torch.autograd.grad((dx * g.t(k + "/v")).sum(), dy)[0]
        assert max_rel(ddy, g.t(k + "/ddy")) < TOL, case
    f = g.t("f/k4")
    for name in ["filter2d", "upsample2d", "downsample2d"]:
        assert max_rel(getattr(O, name)(g.t("wrap/x"), f), g.t("wrap/" + name)) < TOL
    assert torch.equal(O.setup_filter([1, 3, 3, 1]), f)
    assert max_rel(O.setup_filter(g.meta["sym6"]), g.t("f/sym6")) < 1e-7
    assert max_rel(O.setup_filter([1, 3, 3, 1], flip_filter=True, gain=2), g.t("f/k4_flipped_gain2")) < 1e-7


def test_bias_act_golden():
    g = Golden("bias_act")
    for name, spec in g.meta["activation_funcs"].items():
        assert abs(O.ACTIVATIONS[name][1] - spec["def_alpha"]) < 1e-12 and abs(O.ACTIVATIONS[name][2] - spec["def_gain"]) < 1e-7
    for case in g.meta["cases"]:
        k = case["key"]
        x = g.t(k + "/x").requires_grad_(True)
        b = g.t(k + "/b").requires_grad_(True) if case["use_b"] else None
        y = O.bias_act(x, b, dim=1, act=case["act"], clamp=case["clamp"], gain=case["gain"])
        assert max_rel(y, g.t(k + "/y")) < TOL, case
        dy = g.t(k + "/dy").requires_grad_(True)
        grads = torch.autograd.grad((y * dy).sum(), [x] + ([b] if b is not None else []), create_graph=True)
        assert max_rel(grads[0], g.t(k + "/dx")) < TOL, case
        if b is not None:
            assert max_rel(grads[1], g.t(k + "/db")) < TOL, case
        g2 = torch.autograd.grad((grads[0] * g.t(k + "/v")).sum(), [dy, x], allow_unused=True)
        assert max_rel(g2[0], g.t(k + "/d_dy")) < TOL, case
        if (k + "/d_x") in g:
            assert max_rel(g2[1], g.t(k + "/d_x")) < 1e-4, case


def test_conv2d_resample_golden():
    g = Golden("conv2d_resample")
    f = O.setup_filter([1, 3, 3, 1])
    for case in g.meta["cases"]:
        k = case["key"]
        x = g.t(k + "/x").requires_grad_(True); w = g.t(k + "/w").requires_grad_(True)
        y = O.conv2d_resample(x, w, f=f, up=case["up"], down=case["down"], padding=case["k"] // 2, groups=case["groups"], flip_weight=case["flip_weight"])
        assert max_rel(y, g.t(k + "/y")) < TOL, case
        dx, dw = torch.autograd.grad((y * g.t(k + "/dy")).sum(), [x, w])
        assert max_rel(dx, g.t(k + "/dx")) < TOL and max_rel(dw, g.t(k + "/dw")) < TOL, case


def test_modulated_conv2d_golden():
    g = Golden("modulated_conv2d")
    f = O.setup_filter([1, 3, 3, 1])
    for case in g.meta["cases"]:
        k = case["key"]
        x = g.t(k + "/x").requires_grad_(True); w = g.t(k + "/w").requires_grad_(True); s = g.t(k + "/s").requires_grad_(True)
        noise = g.t(k + "/noise") if case["use_noise"] else None
        y = O.modulated_conv2d(x, w, s, noise=noise, up=case["up"], padding=1, resample_filter=f, demodulate=case["demodulate"],
                               flip_weight=(case["up"] == 1), fused_modconv=case["fused_modconv"])
        assert max_rel(y, g.t(k + "/y")) < TOL, case
        gr = torch.autograd.grad((y * g.t(k + "/dy")).sum(), [x, w, s], create_graph=True)
        for got, name in zip(gr, ["dx", "dw", "ds"]):
            assert max_rel(got, g.t(f"{k}/{name}")) < 2e-5, (case, name)
        g2 = torch.autograd.grad(gr[2].square().sum() + gr[0].square().sum(), [w, s])
        assert max_rel(g2[0], g.t(k + "/d2w")) < 1e-4 and max_rel(g2[1], g.t(k + "/d2s")) < 1e-4, case
    a, b, c = g.t("fma/a").requires_grad_(True), g.t("fma/b").requires_grad_(True), g.t("fma/c").requires_grad_(True)
    y = O.fma(a, b, c)
    assert max_rel(y, g.t("fma/y")) < TOL
    for got, name in zip(torch.autograd.grad(y.square().sum(), [a, b, c]), ["da", "db", "dc"]):
        assert max_rel(got, g.t("fma/" + name)) < TOL


@pytest.mark.parametrize("tag", ["skip_resnet", "orig_orig_c3_clamp", "resnet_skip"])
def test_networks_golden(tag):
    g = Golden("networks_" + tag)
    cfg = ON.default_cfg(**{k: v for k, v in g.meta.items() if k != "r1_gamma"})
    gsd, dsd = g.state_dict("G"), g.state_dict("D")
    z, c, real = g.t("z"), g.t("c"), g.t("real")
    ws = ON.mapping(gsd, "mapping", z, c, cfg, num_ws=ON.synthesis_num_ws(cfg))
    assert max_rel(ws, g.t("ws")) < TOL
    img = ON.synthesis(gsd, "synthesis", ws, cfg, noise_mode="const")
    assert max_rel(img, g.t("img")) < 2e-5
    img_fused = ON.synthesis(gsd, "synthesis", ws, cfg, noise_mode="const", fused_modconv=True)
    assert max_rel(img_fused, g.t("img_fused")) < 2e-5
    assert max_rel(ON.discriminator(dsd, img, c, cfg), g.t("logits")) < 2e-5
    # training-step gradients (Gmain, Dmain, R1)
    loss_g, grads_g, loss_d, grads_d, grads_r1 = ON.gd_step_grads(gsd, dsd, cfg, z, g.t("z2"), real, r1_gamma=g.meta["r1_gamma"], c=c)
    assert abs(float(loss_g) - float(g.t("loss_g"))) < 1e-5 and abs(float(loss_d) - float(g.t("loss_d"))) < 1e-5
    for name, got in grads_g.items():
        assert max_rel(got, g.t("gradG/" + name)) < 1e-4 or float(g.t("gradG/" + name).abs().max()) < 1e-9, name
    for name, got in grads_d.items():
        assert max_rel(got, g.t("gradD/" + name)) < 1e-4 or float(g.t("gradD/" + name).abs().max()) < 1e-9, name
    for name, got in grads_r1.items():
        assert max_rel(got, g.t("gradR1/" + name)) < 2e-4 or float(g.t("gradR1/" + name).abs().max()) < 1e-9, name


def test_sg2_attention_hybrid_oracle():
    """oracle/networks.py with the attention hook (oracle/biggan.py attention at the end of the listed blocks) against the reference's
    train_parts models of tests/golden/sg2attent.npz: image, logits, and the spectral-norm buffers the first forward leaves behind"""
    from golden_util import Golden
    from oracle import networks as ON
    g = Golden("sg2attent")
    gk, dk = g.meta["g_kwargs"], g.meta["d_kwargs"]
    cfg = ON.default_cfg(z_dim=gk["z_dim"], w_dim=gk["w_dim"], c_dim=0, img_resolution=gk["img_resolution"], channel_base=gk["synthesis_kwargs"]["channel_base"],
                         channel_max=gk["synthesis_kwargs"]["channel_max"], mapping_layers=gk["mapping_kwargs"]["num_layers"], g_architecture="skip",
                         d_architecture=dk["architecture"], mbstd_group_size=dk["epilogue_kwargs"]["mbstd_group_size"],
                         g_attentions=tuple(gk["attentions"]), d_attentions=tuple(dk["attentions"]))
    gsd, dsd = g.state_dict("G"), g.state_dict("D")
    z, c = g.t("z"), torch.zeros(g.t("z").shape[0], 0)
    upd = {}
    with torch.no_grad():
        img = ON.generator(gsd, z, c, cfg, noise_mode="const", sn_updates=upd)
        logits = ON.discriminator(dsd, g.t("img"), c, cfg)
    assert float((img - g.t("img")).abs().max() / g.t("img").abs().max()) < 1e-5
    assert float((logits - g.t("logits")).abs().max() / g.t("logits").abs().max()) < 1e-4
    for key in g.keys("G_after/"):
        name = key[len("G_after/"):]
        if name in upd:
            assert torch.allclose(upd[name].reshape(g.t(key).shape), g.t(key), atol=1e-5), name
