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
    with pytest.raises(IOError):                                  # data.dataset=image_folder with a path that does not exist
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
    assert [p.name for p in trainer.engine.phases] == ["Gboth", "Dboth"]      # interval 0 -> un-split phases (reference trainers.py:615-618)
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


def _make_image_folder(root, n=12, res=32, labels=True, as_zip=False):
    import json, zipfile
    import numpy as np
    import PIL.Image
    rng = np.random.RandomState(0)
    os.makedirs(root / "00000", exist_ok=True)
    names, arrays = [], []
    for i in range(n):
        a = rng.randint(0, 256, [res, res, 3], dtype=np.uint8)
        name = f"00000/img{i:05d}.png"
        PIL.Image.fromarray(a).save(root / name)
        names.append(name); arrays.append(a)
    if labels:
        json.dump({"labels": [[nm, i % 3] for i, nm in enumerate(names)]}, open(root / "dataset.json", "w"))
    if as_zip:
        zp = str(root) + ".zip"
        with zipfile.ZipFile(zp, "w") as z:
            for nm in names + (["dataset.json"] if labels else []):
                z.write(root / nm, nm)
        return zp, arrays
    return str(root), arrays


def test_image_folder_dataset_and_sampler(tmp_path):
    """directory and zip sources, labels from dataset.json (one-hot), max_size subset, x-flip doubling, rank-sharded endless sampler, loader"""
    import numpy as np
    from style_big_gan_amd.torch_utils import misc
    from style_big_gan_amd.train_parts.dataloaders import dataloaders
    from style_big_gan_amd.train_parts.datasets import datasets
    path, arrays = _make_image_folder(tmp_path / "data")
    zpath, _ = _make_image_folder(tmp_path / "dataz", as_zip=True)
    for src in (path, zpath):
        ds = datasets["image_folder"](path=src, use_labels=True)
        assert len(ds) == 12 and ds.image_shape == [3, 32, 32] and ds.resolution == 32 and ds.num_channels == 3
        assert ds.has_labels and ds.has_onehot_labels and ds.label_dim == 3 and ds.label_shape == [3]
        img, lab = ds[4]
        assert img.dtype == np.uint8 and np.array_equal(img, arrays[4].transpose(2, 0, 1)) and lab.tolist() == [0.0, 1.0, 0.0]
        assert ds.get_details(4).raw_idx == 4 and not ds.get_details(4).xflip
        ds.close()
    nolab = datasets["image_folder"](path=path)
    assert not nolab.has_labels and nolab.label_dim == 0 and nolab[0][1].shape == (0,)
    sub = datasets["image_folder"](path=path, max_size=5, xflip=True, random_seed=3)
    assert len(sub) == 10 and np.array_equal(sub[7][0], sub[2][0][:, :, ::-1]) and sub.get_details(7).xflip
    with pytest.raises(IOError):
        datasets["image_folder"](path=path, resolution=64)
    with pytest.raises(IOError):
        datasets["image_folder"](path=str(tmp_path / "missing"))
    # sampler: two ranks see disjoint, jointly exhaustive index streams over a period
    streams = []
    for rank in range(2):
        it = iter(misc.InfiniteSampler(nolab, rank=rank, num_replicas=2, seed=1, window_size=0))
        streams.append([int(next(it)) for _ in range(6)])
    assert sorted(streams[0] + streams[1]) == list(range(12))
    loader = dataloaders["basic"](dataset=sub, sampler=misc.InfiniteSampler(sub, seed=0), batch_size=4, num_workers=0)
    imgs, labs = next(iter(loader))
    assert imgs.dtype == torch.uint8 and tuple(imgs.shape) == (4, 3, 32, 32) and tuple(labs.shape) == (4, 0)


def test_training_on_an_image_folder(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("plumbing test is for the CPU container")
    path, _ = _make_image_folder(tmp_path / "data", n=20)
    argv = _write(tmp_path, "dcgan.yaml", DCGAN_LIKE) + ["gen.batch=8", "gen.batch_gpu=8", "data.dataset=image_folder", f"data.dataset_path={path}",
                                                         "data.mirror=true", "dataloaders_args.basic.num_workers=0", "gen.kimg=1", "log.metrics=[]"]
    t = starter.main(argv, max_iterations=2)
    assert t.engine.batch_idx == 2 and len(t.dataset) == 40 and t.dataset.resolution == 32 and t.training_set_kwargs["xflip"] is True
    assert all(torch.isfinite(p).all() for p in t.engine.G.parameters())


def test_image_folder_and_sampler_against_the_reference(tmp_path):
    """tests/golden/datasets.npz: the REFERENCE's ImageFolderDataset and InfiniteSampler over the deterministic PNG folder of
    golden_util.make_image_folder (directory and zip, labels, max_size subset with its seed, x-flip doubling) -- item order, flip flags,
    raw indices, pixels, one-hot labels, shape accessors and the samplers' index streams must agree exactly"""
    import numpy as np
    from golden_util import Golden, make_image_folder
    from style_big_gan_amd.torch_utils import misc
    from style_big_gan_amd.train_parts.datasets import datasets
    g = Golden("datasets")
    root = make_image_folder(str(tmp_path / "data"))
    zroot = make_image_folder(str(tmp_path / "dataz"), as_zip=True)
    for case in g.meta["cases"]:
        k = case["key"]
        ds = datasets["image_folder"](path=zroot if case["zip"] else root, **case["kwargs"])
        assert len(ds) == case["len"] and list(ds.image_shape) == case["image_shape"] and list(ds.label_shape) == case["label_shape"]
        assert ds.label_dim == case["label_dim"] and ds.has_labels == case["has_labels"] and ds.has_onehot_labels == case["has_onehot_labels"]
        assert ds.resolution == case["resolution"] and ds.num_channels == case["num_channels"] and ds.name == case["name"]
        imgs = np.stack([ds[i][0] for i in range(len(ds))])
        labs = np.stack([ds[i][1] for i in range(len(ds))]).astype(np.float32)
        assert imgs.dtype == np.uint8 and np.array_equal(imgs, g.npz[f"{k}/images"]), k
        assert np.array_equal(labs, g.npz[f"{k}/labels"]) and np.array_equal(np.stack([ds.get_label(i) for i in range(len(ds))]).astype(np.float32), g.npz[f"{k}/get_label"])
        det = [ds.get_details(i) for i in range(len(ds))]
        assert [int(d.raw_idx) for d in det] == g.npz[f"{k}/raw_idx"].tolist() and [int(d.xflip) for d in det] == g.npz[f"{k}/xflip"].tolist()
        ds.close()
    ds = datasets["image_folder"](path=root)
    for j, kw in enumerate(g.meta["samplers"]):
        it = iter(misc.InfiniteSampler(ds, **kw))
        assert [int(next(it)) for _ in range(60)] == g.npz[f"sampler{j}"].tolist(), kw


REFERENCE_CONFIGS = "/root/reference/configs"


@pytest.mark.parametrize("name", ["dcgan", "sg2ada", "big_gan", "ffhq_sg2", "sg2attent"])
def test_reference_yaml_files_load_unchanged(name, tmp_path):
    """the reference's OWN configs/*.yaml (the five BASELINE.json names) through arguments.load_config + trainer.setup_arguments, as they are:
    container-only (the reference does not travel to the GPU box).  Data comes from the synthetic stand-in -- the yaml's data paths
    (./data/*.zip) do not exist here -- and wandb logging keys are carried but unused."""
    path = os.path.join(REFERENCE_CONFIGS, name + ".yaml")
    if not os.path.isfile(path):
        pytest.skip("reference checkout not present (GPU box)")
    from style_big_gan_amd.train_parts.trainers import trainers
    raw = yaml.safe_load(open(path))
    argv = ["exp.config_dir=" + REFERENCE_CONFIGS, f"exp.config={name}.yaml", "exp.name=t", "data.dataset=synthetic", "data.num_classes=10",
            "data.resolution=" + {"dcgan": "32", "sg2ada": "256", "big_gan": "128", "ffhq_sg2": "1024", "sg2attent": "32"}[name]]
    cfg = arguments.load_config(argv)
    assert cfg.exp.trainer == raw["exp"]["trainer"] and cfg.gen.batch == raw["gen"]["batch"]
    for group in ("gens_args", "discs_args", "optim_gen_args", "optim_disc_args", "disc_regs_all", "gen_regs_all", "losses_arch_args"):
        for key, sub in (raw.get(group) or {}).items():
            for leaf, val in sub.items():
                got = cfg[group][key][leaf]
                assert (dict(got) if isinstance(val, dict) else got) == val or isinstance(val, dict), (group, key, leaf)
    tr = trainers[cfg.exp.trainer]().setup_arguments(cfg)
    assert tr.batch_size == raw["gen"]["batch"] and tr.G_kwargs["img_resolution"] == int(argv[-1].split("=")[1])
    gen_name = cfg.gen.generator
    assert gen_name == raw["gen"].get("generator", gen_name)
    if name == "ffhq_sg2":
        assert [n for n, _ in tr.gen_regs] == ["ppl"] and [n for n, _ in tr.dis_regs] == ["r1"] and tr.G_kwargs["mapping_kwargs"]["num_layers"] == raw["gens_args"]["sg2_classic"]["mapping_kwargs"]["num_layers"]
    if name == "sg2attent":
        assert list(tr.G_kwargs["attentions"]) == list(raw["gens_args"]["sg2_classic"]["attentions"])
