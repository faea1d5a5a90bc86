"""Data loaders (reference ``train_parts/dataloaders.py:7-12``): stock ``torch.utils.data.DataLoader`` under the name 'basic' with the
reference's defaults (pinned host buffers, 3 workers, prefetch 2); fed by ``misc.InfiniteSampler`` it is an endless stream of
``(uint8 [B, C, H, W], float32 [B, label_dim])`` batches sharded by rank."""
import torch

from .. import utils

dataloaders = utils.ClassRegistry()


@dataloaders.add_to_registry("basic")
class BasicDataloader(torch.utils.data.DataLoader):
    def __init__(self, pin_memory=True, num_workers=3, prefetch_factor=2, **args):
        if num_workers == 0:
            prefetch_factor = None          # torch rejects a prefetch factor without workers
        super().__init__(pin_memory=pin_memory and torch.cuda.is_available(), num_workers=num_workers, prefetch_factor=prefetch_factor, **args)
