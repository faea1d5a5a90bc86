"""Fused multiply-add ``a * b + c`` with broadcast-aware gradients.

Host-side mirror of ``stylegan2ada/torch_utils/ops/fma.py`` (``fma`` :15).  The shape the models use --
``a`` [N, C, H, W] activations, ``b`` [N, C, 1, 1] per-sample channel scale, ``c`` [N, 1, H, W] per-pixel addend
(train_parts/generators.py:84) -- runs in the HIP kernel ``sbg_scale_nc`` with reduction gradients from ``sbg_dot_hw``
(see modulate.py); other broadcast patterns use torch.addcmul on the device with the reference's un-broadcasting
gradient rule (:49-58).
"""
import torch

from . import modulate


def fma(a, b, c):   # => a * b + c
    if (a.ndim == 4 and b.ndim == 4 and c.ndim == 4 and a.device.type == "cuda"
            and tuple(b.shape) == (a.shape[0], a.shape[1], 1, 1)
            and tuple(c.shape) == (a.shape[0], 1, a.shape[2], a.shape[3])):
        return modulate.scale_nc(a, b.reshape(a.shape[0], a.shape[1]), c)
    return _FusedMultiplyAdd.apply(a, b, c)


def _sum_to_shape(t, shape):
    """reduce `t` over the dimensions that were broadcast from `shape`"""
    lead = t.ndim - len(shape)
    assert lead >= 0
    dims = [i for i in range(t.ndim) if t.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if dims:
        t = t.sum(dim=dims, keepdim=True)
    if lead:
        t = t.reshape(-1, *t.shape[lead + 1:])
    assert t.shape == shape
    return t


class _FusedMultiplyAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _sum_to_shape(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _sum_to_shape(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _sum_to_shape(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc
