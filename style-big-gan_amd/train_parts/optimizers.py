"""Optimizer registry (reference ``train_parts/optimizers.py:7-11``): stock torch Adam under the name 'adam'."""
import torch

from .. import utils

optimizers = utils.ClassRegistry()


@optimizers.add_to_registry("adam")
class Adam(torch.optim.Adam):
    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        params = list(params)
        # on the GPU the whole update (moments, bias correction, step) is one multi-tensor kernel per chunk of parameters instead of the
        # six foreach passes of the default implementation; same arithmetic
        flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
        fused = len(flat) > 0 and all(isinstance(p, torch.Tensor) and p.device.type == "cuda" and p.is_floating_point() for p in flat)
        super().__init__(params, lr=lr, betas=tuple(float(b) for b in betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                         **(dict(fused=True) if fused else {}))
