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


def multiprocesses_main(rank, trainer, temp_dir, max_iterations=None):
    trainer.rank = rank
    trainer.setup_logs()
    trainer.distribute_torch(temp_dir)
    trainer.init_params()
    trainer.setup_dataset()
    trainer.setup_networks()
    trainer.setup_augmentations()
    trainer.distrib_acrros_gpu()
    trainer.setup_training_phases()
    trainer.export_sample_images()
    trainer.training_loop(max_iterations)


if __name__ == "__main__":
    main()
