"""CPU: the config / registry surface -- structured defaults < yaml < CLI dot-list (reference arguments.py:146-159), run on
yaml written in the shape of the reference's configs/*.yaml, and the dcgan plumbing config end to end through starter."""
import os
import sys

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import style_big_gan_amd  # noqa: E402
from style_big_gan_amd import arguments, starter  # noqa: E402

SG2ADA_LIKE = {
    "exp": {"trainer": "sg2"},
    "gen": {"kimg": 20000, "batch": 64, "disc_regs": ["r1"]},
    "log": {"output": "./logs"},
    "data": {"dataset_path": "./data/cifar10.zip"},
    "datasets_args": {"image_folder": {"use_labels": True, "max_size": 50000}},
    "gens_args": {"sg2_classic": {"z_dim": 512, "w_dim": 512, "mapping_kwargs": {"num_layers": 2}}},
    "discs_args": {"sg2_classic": {"epilogue_kwargs": {"mbstd_group_size": 32}, "architecture": "orig"}},
    "optim_gen_args": {"adam": {"lr": 0.0025, "betas": [0, 0.99]}},
    "optim_disc_args": {"adam": {"lr": 0.0025, "betas": [0, 0.99]}},
    "ema": {"kimg": 500, "ramp": 0.05},
    "losses_arch_args": {"sg2": {"style_mixing_prob": 0}},
    "disc_regs_all": {"r1": {"r1_gamma": 0.01}},
}

DCGAN_LIKE = {
    "exp": {"trainer": "base"},
    "gen": {"kimg": 3200, "batch": 128, "loss_arch": "base", "loss": "bcew", "generator": "cnn32_dcgan", "discriminator": "cnn32_dcgan",
            "g_reg_interval": 0, "d_reg_interval": 0},
    "gens_args": {"cnn32_dcgan": {"z_dim": 100}},
    "optim_gen_args": {"adam": {"lr": 0.0002, "betas": [0.5, 0.9]}},
    "optim_disc_args": {"adam": {"lr": 0.0002, "betas": [0.5, 0.9]}},
    "ema": {"use_ema": False},
    "aug": {"aug": "noaug"},
}


def _write(tmp_path, name, cfg):
    with open(os.path.join(tmp_path, name), "w") as fh:
        yaml.safe_dump(cfg, fh)
    return ["exp.config_dir=" + str(tmp_path), "exp.config=" + name, "exp.name=test"]


def test_precedence_and_registry_groups(tmp_path):
    argv = _write(tmp_path, "sg2ada.yaml", SG2ADA_LIKE) + ["gen.batch=32", "gens_args.sg2_classic.synthesis_kwargs.channel_base=16384", "perf.gpus=2"]
    cfg = arguments.load_config(argv)
    assert cfg.exp.trainer == "sg2" and cfg.exp.name == "test"
    assert cfg.gen.batch == 32                               # CLI beats yaml (64)
    assert cfg.gen.batch_gpu == 32 and cfg.gen.loss == "softplus"   # structured defaults survive
    assert cfg.gens_args.sg2_classic.z_dim == 512            # yaml beats the class default (128)
    assert cfg.gens_args.sg2_classic.mapping_kwargs.num_layers == 2 and cfg.gens_args.sg2_classic.mapping_kwargs.lr_multiplier == 0.01
    assert cfg.gens_args.sg2_classic.synthesis_kwargs.channel_base == 16384 and cfg.gens_args.sg2_classic.synthesis_kwargs.channel_max == 512
    assert cfg.discs_args.sg2_classic.architecture == "orig" and cfg.discs_args.sg2_classic.epilogue_kwargs.mbstd_group_size == 32
    assert cfg.disc_regs_all.r1.r1_gamma == 0.01 and cfg.disc_regs_all.grad_pen.alpha == 10.0
    assert cfg.optim_gen_args.adam.betas == [0, 0.99] and cfg.perf.gpus == 2
    assert set(arguments.missing_keys(cfg)) >= {"optim_gen_args.adam.params"}
    with pytest.raises(KeyError):
        arguments.merge(arguments.structured_defaults(), {"no_such_group": {"x": 1}})
    assert arguments.parse_dotlist(["a.b=[1, 2]", "a.c=true", "d=hello"]) == {"a": {"b": [1, 2], "c": True}, "d": "hello"}


def test_trainer_validation(tmp_path):
    from style_big_gan_amd.train_parts.trainers import trainers
    argv = _write(tmp_path, "sg2ada.yaml", SG2ADA_LIKE)
    cfg = arguments.load_config(argv)
    with pytest.raises(NotImplementedError, match="synthetic"):   # dataset loading is out of scope; must be switched explicitly
        trainers["sg2"]().setup_arguments(cfg)
    ada = trainers["sg2"]().setup_arguments(arguments.load_config(argv + ["data.dataset=synthetic", "data.resolution=64"])).aug
    assert ada["ada_target"] == 0.6 and ada["augment_p"] == 0.0 and ada["ada_interval"] == 4 and ada["augment_kwargs"]["hue"] == 1
    assert sum(v == 1 for k, v in ada["augment_kwargs"].items() if k not in ("rotate_max", "hue_max", "saturation_std", "imgfilter_std")) == 12 and ada["augment_kwargs"]["xint_max"] == 0.125 and ada["augment_type"] == "sg2_ada"       # 'bgc' = blit + geom + color
    fixed = trainers["sg2"]().setup_arguments(arguments.load_config(argv + ["data.dataset=synthetic", "aug.aug=fixed", "aug.p=0.3", "aug.augpipe=bg"])).aug
    assert fixed["ada_target"] is None and fixed["augment_p"] == 0.3 and fixed["augment_kwargs"]["hue"] == 0 and fixed["augment_kwargs"]["xflip"] == 1
    with pytest.raises(ValueError):
        trainers["sg2"]().setup_arguments(arguments.load_config(argv + ["data.dataset=synthetic", "aug.p=0.3"]))
    cfg = arguments.load_config(argv + ["aug.aug=noaug", "data.dataset=synthetic", "data.resolution=64"])
    tr = trainers["sg2"]().setup_arguments(cfg)
    assert tr.batch_size == 64 and tr.batch_gpu == 32 and tr.dis_regs == [("r1", {"r1_gamma": 0.01})]
    assert tr.G_kwargs["img_resolution"] == 64 and tr.G_kwargs["z_dim"] == 512 and tr.D_kwargs["architecture"] == "orig"
    assert tr.G_opt == ("adam", {"lr": 0.0025, "betas": [0, 0.99], "eps": 1e-08, "weight_decay": 0, "amsgrad": False})
    cfg = arguments.load_config(argv + ["aug.aug=noaug", "data.dataset=synthetic", "perf.gpus=3"])
    with pytest.raises(ValueError):
        trainers["sg2"]().setup_arguments(cfg)


def test_dcgan_config_runs_on_cpu(tmp_path):
    """configs/dcgan.yaml-shaped run: 32x32, batch 16, CPU eager, two iterations through starter.main"""
    argv = _write(tmp_path, "dcgan.yaml", DCGAN_LIKE) + ["gen.batch=16", "gen.batch_gpu=16", "data.dataset=synthetic", "data.resolution=32", "gen.kimg=1"]
    if torch.cuda.is_available():
        pytest.skip("plumbing test is for the CPU container")
    trainer = starter.main(argv, max_iterations=2)
    assert trainer.engine.batch_idx == 2 and trainer.engine.cur_nimg == 32
    assert [p.name for p in trainer.engine.phases] == ["Gmain", "Dmain"]
    assert all(torch.isfinite(p).all() for p in trainer.engine.G.parameters())
    dry = starter.main(argv + ["exp.dry_run=true"])
    assert not hasattr(dry, "engine")


def test_snapshot_and_resume(tmp_path):
    """save_snapshot -> a weights-only loadable file + training_options.json; a fresh trainer resumed from it holds the same networks,
    optimizer moments and counters, and continues from there"""
    if torch.cuda.is_available():
        pytest.skip("plumbing test is for the CPU container")
    import json
    argv = _write(tmp_path, "dcgan.yaml", DCGAN_LIKE) + ["gen.batch=16", "gen.batch_gpu=16", "data.dataset=synthetic", "data.resolution=32", "gen.kimg=1"]
    a = starter.main(argv, max_iterations=2)
    path = a.save_snapshot(run_dir=str(tmp_path / "run"))
    assert os.path.basename(path) == "network-snapshot-000000.pt"
    opts = json.load(open(tmp_path / "run" / "training_options.json"))
    assert opts["start_options"] == {"cur_nimg": 32, "batch_idx": 2} and opts["snapshot"] == os.path.basename(path)
    state = torch.load(path, weights_only=True)
    assert set(state) >= {"G", "D", "optimizers", "progress"} and set(state["optimizers"]) == {"G", "D"}
    b = starter.main(argv + [f"trans.resume={path}"], max_iterations=0)
    assert b.engine.batch_idx == 2 and b.engine.cur_nimg == 32
    for pa, pb in zip(list(a.engine.G.parameters()) + list(a.engine.D.parameters()), list(b.engine.G.parameters()) + list(b.engine.D.parameters())):
        assert torch.equal(pa, pb)
    sa, sb = a.engine.phases[0].opt.state_dict()["state"], b.engine.phases[0].opt.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)
    b.training_loop(max_iterations=1)
    assert b.engine.batch_idx == 3 and b.engine.cur_nimg == 48
    with pytest.raises(ValueError):
        starter.main(argv + ["trans.resume=/nonexistent/file.pt"], max_iterations=1)
