"""CPU: host logic and the C-ABI surface (no compute calls -- there is no GPU here).

* libsbg_hip.so loads and exports every symbol include/sbg_hip.h declares;
* the op modules refuse CPU tensors / impl='ref' loudly (no CPU fallback in the product);
* registry / kwargs dataclasses, model construction, state_dict key parity with the golden (reference) state_dicts;
* padding arithmetic of the upfirdn2d wrappers and conv2d_resample against the oracle's shapes.
"""
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import style_big_gan_amd  # noqa: E402
from style_big_gan_amd import _lib  # noqa: E402
from style_big_gan_amd.torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, upfirdn2d  # noqa: E402
from golden_util import Golden  # noqa: E402


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sbg_hip.h")).read()
    declared = set(re.findall(r"\b(sbg_[a-z0-9_]+)\s*\(", header))
    declared -= {"sbg_stream_t"}
    lib = _lib.load()
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, f"header vs ctypes binding mismatch: {declared ^ bound}"
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sbg_version() >= 1
    assert lib.sbg_prof_enable(0) in (0, 1)


def test_no_cpu_fallback():
    x = torch.randn(2, 4, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path|CPU"):
        bias_act.bias_act(x, act="lrelu")
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(x, upfirdn2d.setup_filter([1, 3, 3, 1]))
    with pytest.raises(RuntimeError):
        conv2d_gradfix.conv2d(x, torch.randn(4, 4, 3, 3))
    with pytest.raises(RuntimeError):
        bias_act.bias_act(x, act="lrelu", impl="ref")
    with pytest.raises(AssertionError):
        bias_act.bias_act(x, act="lrelu", impl="bogus")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsbg_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_activation_table_matches_reference():
    g = Golden("bias_act")
    for name, spec in g.meta["activation_funcs"].items():
        ours = bias_act.activation_funcs[name]
        assert ours.cuda_idx == spec["cuda_idx"] and ours.ref == spec["ref"] and ours.has_2nd_grad == spec["has_2nd_grad"]
        assert abs(ours.def_alpha - spec["def_alpha"]) < 1e-12 and abs(ours.def_gain - spec["def_gain"]) < 1e-7


def test_setup_filter_matches_reference():
    g = Golden("upfirdn2d")
    assert torch.equal(upfirdn2d.setup_filter([1, 3, 3, 1]), g.t("f/k4"))
    assert torch.allclose(upfirdn2d.setup_filter(g.meta["sym6"]), g.t("f/sym6"), atol=1e-8)
    assert torch.allclose(upfirdn2d.setup_filter([1, 3, 3, 1], flip_filter=True, gain=2), g.t("f/k4_flipped_gain2"), atol=1e-8)
    assert upfirdn2d.setup_filter(None).shape == (1, 1)
    assert upfirdn2d._parse_padding([1, 2]) == (1, 1, 2, 2) and upfirdn2d._parse_padding(3) == (3, 3, 3, 3)


def test_registry_and_state_dict_keys():
    from style_big_gan_amd.train_parts import discriminators as PD, generators as PG, losses, losses_base, optimizers, regularizations
    assert {"sg2_classic", "cnn32_dcgan", "cnn48_dcgan"} <= set(PG.generators.classes)
    assert {"sg2_classic", "cnn32_dcgan", "cnn48_dcgan"} <= set(PD.discriminators.classes)
    assert set(losses.losses.classes) == {"bcew", "hinge", "wasserstein", "softplus"}
    assert set(losses_base.losses_arch.classes) == {"base", "sg2"}
    assert set(regularizations.generator_regs.classes) == {"ppl"} and set(regularizations.discriminator_regs.classes) == {"r1", "grad_pen"}
    assert "adam" in optimizers.optimizers.classes
    args = PG.generators.args["sg2_classic"]()          # dataclass synthesised from __init__ (reference utils.py:88-118)
    assert args.z_dim == 128 and args.c_dim is None and args.img_resolution is None
    for tag in ["skip_resnet", "orig_orig_c3_clamp", "resnet_skip"]:
        g = Golden("networks_" + tag)
        m = g.meta
        G = PG.generators["sg2_classic"](z_dim=m["z_dim"], c_dim=m["c_dim"], w_dim=m["w_dim"], img_resolution=16, img_channels=3,
                                         mapping_kwargs=dict(num_layers=2),
                                         synthesis_kwargs=dict(channel_base=m["channel_base"], channel_max=m["channel_max"],
                                                               block_kwargs=dict(architecture=m["g_architecture"])))
        D = PD.discriminators["sg2_classic"](c_dim=m["c_dim"], img_resolution=16, img_channels=3, architecture=m["d_architecture"],
                                             channel_base=m["channel_base"], channel_max=m["channel_max"], mapping_kwargs=dict(num_layers=2))
        assert set(G.state_dict().keys()) == set(g.state_dict("G").keys())
        assert set(D.state_dict().keys()) == set(g.state_dict("D").keys())
        for k, v in g.state_dict("G").items():
            assert G.state_dict()[k].shape == v.shape, k


def test_dcgan_plumbing_runs_on_cpu():
    """configs/dcgan.yaml: 32x32, batch 16, stock torch.nn layers, bcew loss through the 'base' loss arch (whose constructor
    is broken in the reference by a `__int__` typo -- here it works)."""
    from style_big_gan_amd.train_parts import trainers
    eng = trainers.StepEngine("cpu", generator="cnn32_dcgan", discriminator="cnn32_dcgan",
                              gen_kwargs=dict(z_dim=100, c_dim=0, img_resolution=32), disc_kwargs=dict(),
                              loss_arch="base", loss="bcew", gen_regs=[], dis_regs=[],
                              optim_gen=("adam", dict(lr=2e-4, betas=[0.5, 0.9])), optim_disc=("adam", dict(lr=2e-4, betas=[0.5, 0.9])),
                              g_reg_interval=0, d_reg_interval=0, batch=16, batch_gpu=16, use_ema=False)
    assert [p.name for p in eng.phases] == ["Gboth", "Dboth"]
    before = [p.detach().clone() for p in eng.G.parameters()]
    eng.train_iteration(torch.rand(16, 3, 32, 32) * 2 - 1, None)
    assert any(not torch.equal(a, b) for a, b in zip(before, eng.G.parameters()))
    assert all(torch.isfinite(p).all() for p in eng.D.parameters())


def test_lazy_regularisation_phases():
    from style_big_gan_amd.train_parts import trainers
    kw = trainers.lazy_reg_opt_kwargs(dict(lr=0.0025, betas=[0, 0.99]), 4)
    assert abs(kw["lr"] - 0.0025 * 0.8) < 1e-12 and abs(kw["betas"][1] - 0.99 ** 0.8) < 1e-12


def test_phase_construction_follows_the_interval_alone():
    """reference trainers.py:615-627 branches on reg_interval only.  sg2ada.yaml: g_reg_interval 16 and NO generator regulariser still
    gives Gmain + an (idle) Greg slot and scales G's Adam to lr * 16/17, beta2 ** (16/17); D: interval 4 with R1.  n_dis stretches only
    an un-split 'Gboth' phase (:609-610, :618); with interval 0 the phases are 'Gboth' / 'Dboth'."""
    from style_big_gan_amd.train_parts import trainers
    tiny = dict(generator="cnn32_dcgan", discriminator="cnn32_dcgan", gen_kwargs=dict(z_dim=8, c_dim=0, img_resolution=32), disc_kwargs=dict(),
                loss_arch="base", loss="softplus", batch=4, batch_gpu=4, use_ema=False,
                optim_gen=("adam", dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)), optim_disc=("adam", dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)))
    eng = trainers.StepEngine("cpu", gen_regs=[], dis_regs=[("r1", dict(r1_gamma=0.01))], g_reg_interval=16, d_reg_interval=4, n_dis=3, **tiny)
    assert [(p.name, p.interval, p.idle) for p in eng.phases] == [("Gmain", 1, False), ("Greg", 16, True), ("Dmain", 1, False), ("Dreg", 4, False)]
    g_opt, d_opt = eng.phases[0].opt.param_groups[0], eng.phases[2].opt.param_groups[0]
    assert eng.phases[0].opt is eng.phases[1].opt and eng.phases[2].opt is eng.phases[3].opt
    assert abs(g_opt["lr"] - 0.0025 * 16 / 17) < 1e-15 and abs(g_opt["lr"] - 0.002353) < 1e-6
    assert tuple(g_opt["betas"]) == (0.0, 0.99 ** (16 / 17)) and abs(g_opt["betas"][1] - 0.99058) < 1e-5
    assert abs(d_opt["lr"] - 0.0025 * 4 / 5) < 1e-15 and tuple(d_opt["betas"]) == (0.0, 0.99 ** (4 / 5))
    # the idle slot neither steps the optimizer nor touches its moments, but four phases' worth of latents are drawn per iteration
    drawn = []
    randn = torch.randn
    torch.randn = lambda *a, **k: (drawn.append(a[0]) if a and isinstance(a[0], (list, tuple)) and len(a[0]) == 2 else None, randn(*a, **k))[1]
    try:
        eng.train_iteration(torch.rand(4, 3, 32, 32) * 2 - 1, None)
    finally:
        torch.randn = randn
    assert [4 * 4, 8] in [list(d) for d in drawn]
    assert all(int(st["step"]) == 1 for st in eng.phases[0].opt.state.values())      # Gmain stepped once, the idle Greg not at all
    eng = trainers.StepEngine("cpu", gen_regs=[], dis_regs=[], g_reg_interval=0, d_reg_interval=0, n_dis=3, **tiny)
    assert [(p.name, p.interval) for p in eng.phases] == [("Gboth", 3), ("Dboth", 1)]
    assert eng.phases[0].opt.param_groups[0]["lr"] == 0.0025
