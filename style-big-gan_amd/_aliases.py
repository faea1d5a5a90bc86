"""Register this package's modules under the reference's import paths (drop-in for code written against the reference)."""
import importlib
import sys
import types

_MAP = {
    "stylegan2ada.torch_utils.ops.bias_act": "torch_utils.ops.bias_act",
    "stylegan2ada.torch_utils.ops.upfirdn2d": "torch_utils.ops.upfirdn2d",
    "stylegan2ada.torch_utils.ops.conv2d_resample": "torch_utils.ops.conv2d_resample",
    "stylegan2ada.torch_utils.ops.conv2d_gradfix": "torch_utils.ops.conv2d_gradfix",
    "stylegan2ada.torch_utils.ops.fma": "torch_utils.ops.fma",
    "stylegan2ada.torch_utils.ops.grid_sample_gradfix": "torch_utils.ops.grid_sample_gradfix",
    "stylegan2ada.training.augment": "train_parts.augmentations",
    "train_parts.augmentations": "train_parts.augmentations",
    "train_parts.datasets": "train_parts.datasets",
    "train_parts.dataloaders": "train_parts.dataloaders",
    "stylegan2ada.torch_utils.custom_ops": "torch_utils.custom_ops",
    "stylegan2ada.torch_utils.misc": "torch_utils.misc",
    "stylegan2ada.torch_utils.training_stats": "torch_utils.training_stats",
    "train_parts.generators": "train_parts.generators",
    "train_parts.discriminators": "train_parts.discriminators",
    "train_parts.losses": "train_parts.losses",
    "train_parts.losses_base": "train_parts.losses_base",
    "train_parts.regularizations": "train_parts.regularizations",
    "train_parts.optimizers": "train_parts.optimizers",
    "train_parts.trainers": "train_parts.trainers",
    "biggan.layers": "biggan.layers",
    "stylegan2ada.metrics.metric_main": "metrics.metric_main",
    "stylegan2ada.metrics.metric_utils": "metrics.metric_utils",
    "utils": "utils",
}


def _ensure_parents(name):
    parts = name.split(".")
    for i in range(1, len(parts)):
        parent = ".".join(parts[:i])
        if parent not in sys.modules:
            mod = types.ModuleType(parent)
            mod.__path__ = []
            sys.modules[parent] = mod


def install(overwrite=False):
    pkg = __name__.rsplit(".", 1)[0]
    for ref_name, ours in _MAP.items():
        if ref_name in sys.modules and not overwrite:
            continue
        try:
            mod = importlib.import_module(f"{pkg}.{ours}")
        except ModuleNotFoundError:
            continue
        _ensure_parents(ref_name)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, mod)
