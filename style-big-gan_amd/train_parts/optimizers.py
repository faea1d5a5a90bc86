"""Optimizer registry (reference ``train_parts/optimizers.py:7-11``): stock torch Adam under the name 'adam'."""
import torch

from .. import utils

optimizers = utils.ClassRegistry()


@optimizers.add_to_registry("adam")
class Adam(torch.optim.Adam):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=tuple(float(b) for b in betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
