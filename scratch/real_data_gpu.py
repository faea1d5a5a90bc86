"""GPU sanity of the real-data path: a generated PNG folder -> image_folder dataset -> sg2 trainer at 32x32, three iterations."""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_config_cpu as T
from style_big_gan_amd import starter
tmp = pathlib.Path(tempfile.mkdtemp())
path, _ = T._make_image_folder(tmp / "data", n=40)
argv = T._write(tmp, "sg2ada.yaml", T.SG2ADA_LIKE) + ["gen.batch=8", "gen.batch_gpu=4", "data.dataset=image_folder", f"data.dataset_path={path}",
                                                    "data.mirror=true", "aug.aug=ada", "gen.kimg=1", "dataloaders_args.basic.num_workers=2"]
t = starter.main(argv, max_iterations=3)
print("iterations", t.engine.batch_idx, "dataset", len(t.dataset), t.dataset.resolution, "ada p", float(t.engine.augment_pipe.p),
      "finite", all(torch.isfinite(p).all().item() for p in t.engine.G.parameters()))
