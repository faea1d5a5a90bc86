"""Experiment configuration: structured defaults < yaml file < CLI dot-list (host code).

Counterpart of the reference's ``arguments.py``: the same eight hand-written groups (``exp, data, log, gen, perf, ema, aug,
trans`` -- :19-109) plus the registry-derived groups (``gens_args, discs_args, optim_gen_args, optim_disc_args,
losses_arch_args, gen_regs_all, disc_regs_all`` -- :112-143, each holding one kwargs block per registered class, synthesised
from its ``__init__`` signature), and the same precedence as ``load_config`` (:146-159), so the reference's ``configs/*.yaml``
and ``key.sub=value`` overrides (Readme.md:28-30) load unchanged.  The reference builds this on OmegaConf; that package is
optional here -- this module is a small yaml + dot-list implementation of the same contract.  Groups for components that are
out of scope are accepted and carried verbatim.
"""
import copy
import dataclasses
import os
import sys

import yaml

from . import utils
from .utils import EasyDict, MISSING

args = utils.ClassRegistry()


def _group(group_name, **defaults):
    args.classes[group_name] = lambda d=defaults: EasyDict(copy.deepcopy(d))


_group("exp", config_dir=MISSING, config=MISSING, name=MISSING, project="gan-collections", notes="empty notes", dry_run=False, trainer="base")
_group("data", dataset="image_folder", dataloader="basic", dataset_path="./data", cond=False, subset=0, mirror=False)
_group("log", snap=50, output="./outputs", metrics=["fid50k_full", "is50k"], kimg_per_tick=4, wandb=True)
_group("gen", kimg=-1, batch=-1, batch_gpu=32, seed=0, generator="sg2_classic", discriminator="sg2_classic", optim_gen="adam",
       optim_disc="adam", gen_regs=[], disc_regs=[], loss_arch="sg2", loss="softplus", g_reg_interval=16, d_reg_interval=4, n_dis=1)
_group("perf", fp32=False, nhwc=False, allow_tf32=False, nobench=False, gpus=1)
_group("ema", use_ema=True, kimg=20, ramp=-1)
_group("aug", aug="ada", aug_type="sg2_ada", p=-1, target=-1, augpipe="bgc")
_group("trans", resume="noresume", resume_url="", freezed=-1, resume_model="", resume_dir="", args_name="training_options.json")


def _to_plain(obj):
    """kwargs dataclass instance -> nested EasyDict"""
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return EasyDict({f.name: _to_plain(getattr(obj, f.name)) for f in dataclasses.fields(obj)})
    if isinstance(obj, dict):
        return EasyDict({k: _to_plain(v) for k, v in obj.items()})
    if isinstance(obj, tuple):
        return list(obj)
    return obj


def _registry_group(registry):
    return lambda: EasyDict({name: _to_plain(cls()) for name, cls in registry.args.items()})


def _register_model_groups():
    from .train_parts.discriminators import discriminators
    from .train_parts.generators import generators
    from .train_parts.losses_base import losses_arch
    from .train_parts.optimizers import optimizers
    from .train_parts.regularizations import discriminator_regs, generator_regs
    args.classes["gens_args"] = _registry_group(generators)
    args.classes["discs_args"] = _registry_group(discriminators)
    args.classes["optim_gen_args"] = _registry_group(optimizers)
    args.classes["optim_disc_args"] = _registry_group(optimizers)
    args.classes["losses_arch_args"] = _registry_group(losses_arch)
    args.classes["gen_regs_all"] = _registry_group(generator_regs)
    args.classes["disc_regs_all"] = _registry_group(discriminator_regs)
    from .train_parts.augmentations import augmentations
    args.classes["augpipe_specs"] = _registry_group(augmentations)
    from .train_parts.datasets import datasets
    from .train_parts.dataloaders import dataloaders
    args.classes["datasets_args"] = _registry_group(datasets)
    args.classes["dataloaders_args"] = _registry_group(dataloaders)


def structured_defaults():
    if "gens_args" not in args.classes:
        _register_model_groups()
    return EasyDict({name: factory() for name, factory in args.classes.items()})


def merge(base, override, path=""):
    """recursive merge of `override` into `base` (in place); unknown top-level groups are rejected like a structured config"""
    for k, v in (override or {}).items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            merge(base[k], v, f"{path}{k}.")
        else:
            if path == "" and k not in base:
                raise KeyError(f"unknown config group '{k}'")
            base[k] = _to_plain(v) if isinstance(v, dict) else v
    return base


def parse_dotlist(items):
    """['gen.batch=50', 'exp.name=run'] -> nested dict; values parsed as yaml scalars (ints, floats, bools, lists, strings)"""
    out = {}
    for item in items:
        if "=" not in item:
            raise ValueError(f"override '{item}' is not of the form key=value")
        key, value = item.split("=", 1)
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(value) if value != "" else ""
    return out


def load_config(argv=None):
    """structured defaults, then exp.config_dir/exp.config (yaml), then the command line"""
    cli = parse_dotlist(sys.argv[1:] if argv is None else argv)
    config = structured_defaults()
    exp = cli.get("exp", {})
    if "config" not in exp or "config_dir" not in exp:
        raise ValueError("exp.config_dir=<dir> and exp.config=<file.yaml> are required")
    with open(os.path.join(exp["config_dir"], exp["config"])) as fh:
        merge(config, yaml.safe_load(fh) or {})
    merge(config, cli)
    return config


def missing_keys(config, prefix=""):
    out = []
    for k, v in config.items():
        if isinstance(v, dict):
            out += missing_keys(v, f"{prefix}{k}.")
        elif isinstance(v, str) and v == MISSING:
            out.append(prefix + k)
    return out
