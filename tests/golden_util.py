"""Loader for the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.npz = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(bytes(self.npz["__meta__"]).decode())

    def __contains__(self, key):
        return key in self.npz.files

    def t(self, key):
        return torch.from_numpy(self.npz[key].copy())

    def keys(self, prefix):
        return [k for k in self.npz.files if k.startswith(prefix)]

    def state_dict(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self.t(k) for k in self.npz.files if k.startswith(p)}


def max_rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
