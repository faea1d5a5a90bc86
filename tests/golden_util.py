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


def make_image_folder(root, n=14, res=16, seed=7, grey_every=None, as_zip=False):
    """deterministic PNG folder (two sub-directories, dataset.json with class indices) -- written identically by make_golden.py, which runs
    the REFERENCE's ImageFolderDataset over it, and by the tests, which run this package's"""
    import zipfile
    import PIL.Image
    rng = np.random.RandomState(seed)
    names = []
    for i in range(n):
        sub = "00000" if i % 2 == 0 else "00001"
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        name = f"{sub}/img{i:05d}.png"
        PIL.Image.fromarray(rng.randint(0, 256, [res, res, 3], dtype=np.uint8)).save(os.path.join(root, name))
        names.append(name)
    with open(os.path.join(root, "dataset.json"), "w") as f:
        json.dump({"labels": [[nm, int(rng.randint(0, 4))] for nm in names]}, f)
    if as_zip:
        zp = root.rstrip("/") + ".zip"
        with zipfile.ZipFile(zp, "w") as z:
            for nm in names + ["dataset.json"]:
                z.write(os.path.join(root, nm), nm)
        return zp
    return root
