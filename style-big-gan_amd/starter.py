"""Entry point: ``python -m style_big_gan_amd.starter exp.config_dir=<dir> exp.config=<file.yaml> exp.name=<run> [key=value ...]``

Counterpart of the reference's ``starter.py`` (``main`` :12-30, ``multiprocesses_main`` :32-45): load the structured config,
let the registered trainer validate it, then run one process per GPU through the same lifecycle calls.
"""
import tempfile

import torch

from . import arguments
from .train_parts.trainers import trainers


def main(argv=None, max_iterations=None):
    config = arguments.load_config(argv)
    trainer = trainers[config.exp.trainer]()
    trainer.setup_arguments(config)
    if config.exp.dry_run:
        print("Dry run; exiting.")
        return trainer
    with tempfile.TemporaryDirectory() as temp_dir:
        if config.perf.gpus == 1:
            multiprocesses_main(0, trainer, temp_dir, max_iterations)
        else:
            torch.multiprocessing.set_start_method("spawn", force=True)
            torch.multiprocessing.spawn(fn=multiprocesses_main, args=(trainer, temp_dir, max_iterations), nprocs=config.perf.gpus)
    return trainer


# the trainer lifecycle the reference drives on every rank (starter.py:32-45); a trainer class may override any stage
LIFECYCLE = ("setup_logs", "distribute_torch", "init_params", "setup_dataset", "setup_networks", "setup_augmentations", "distrib_acrros_gpu",
             "setup_training_phases", "export_sample_images", "training_loop")
_STAGE_ARGS = {"distribute_torch": lambda temp_dir, max_iterations: (temp_dir,), "training_loop": lambda temp_dir, max_iterations: (max_iterations,)}


def multiprocesses_main(rank, trainer, temp_dir, max_iterations=None):
    """one rank: every lifecycle stage in order"""
    trainer.rank = rank
    for stage in LIFECYCLE:
        getattr(trainer, stage)(*_STAGE_ARGS.get(stage, lambda *_: ())(temp_dir, max_iterations))


if __name__ == "__main__":
    main()
