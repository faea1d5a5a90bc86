"""Convolution with the bias + activation + gain + clamp tail fused into the matrix-core kernel's epilogue.

What the reference's layers compute as ``conv2d_resample(...)`` followed by ``bias_act.bias_act(...)``
(train_parts/discriminators.py:115-124, generators.py:176-185) becomes one launch: the pre-activation tensor is never written.
Exact for every derivative order with the piecewise-linear activations ('linear', 'relu', 'lrelu'): the backward is
``bias_act``'s own gradient Function on the saved OUTPUT followed by the differentiable data / weight gradient Functions of
``conv2d_gradfix``, so R1's double backward goes through unchanged.
"""
import torch

from . import bias_act as _ba
from . import conv2d_gradfix as _cg


class _ConvBiasAct(torch.autograd.Function):
    """cfg = (stride, padding, act name, alpha, gain, clamp, wgain); w in x's dtype, or the fp32 master weight (see conv2d_gradfix._Conv)"""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        stride, padding, act, alpha, gain, clamp, wgain = cfg
        epi = _cg.Epilogue(bias=b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        y = _cg._conv_forward(x, w, stride, padding, epi=epi, wgain=wgain)
        ctx.save_for_backward(x, w, y)
        ctx.cfg = cfg
        ctx.has_bias = b is not None
        ctx.b_dtype = b.dtype if b is not None else None       # fp32 parameters may be passed as they are: the epilogue reads fp32
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, padding, act, alpha, gain, clamp, wgain = ctx.cfg
        bcfg = _ba._Cfg((1, act, float(alpha), float(gain), float(clamp)))
        fmt = torch.channels_last
        d1 = dy.contiguous(memory_format=fmt)
        db_fused = None
        if not bcfg.trivial:
            fused = _ba._grad_and_bias_sum(d1, y, bcfg) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            if fused is not None:           # first order: activation gradient and bias gradient in one pass over (dy, y)
                d1, db_fused = fused
            else:
                d1 = _ba._BiasActGrad.apply(d1, None, None, y, bcfg, fmt)
        dx = dw = db = None
        ccfg = (False, stride, padding, (0, 0), wgain)
        if ctx.needs_input_grad[0]:
            op = _cg._output_padding_for(False, stride, padding, x.shape[2:], d1.shape[2:], w.shape[2:])
            dx = _cg._Conv.apply(d1, w, (True, stride, padding, op, wgain))
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            dw = _cg._ConvWgrad.apply(d1, x, ccfg, tuple(w.shape), _cg.wmeta_of(x, w))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = (db_fused if db_fused is not None else _ba._sum_to_bias(d1, 1)).to(ctx.b_dtype)
        return dx, dw, db, None


def fusable(x, w, act, groups=1):
    return (groups == 1 and x.device.type == "cuda" and _cg.epilogue_fusable(x) and act in ("linear", "relu", "lrelu")
            and (x.dtype == w.dtype or _cg.is_mixed(x, w)))


def conv2d_bias_act(x, w, b=None, stride=1, padding=0, act="linear", alpha=None, gain=None, clamp=None, wgain=1.0):
    """bias_act(conv2d(x, w, stride, padding), b, act=act, alpha=alpha, gain=gain, clamp=clamp) in one kernel (16-bit tensors)"""
    spec = _ba.activation_funcs[act]
    cfg = (_cg._pair(stride), _cg._pair(padding), act,
           float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain),
           float(clamp if clamp is not None else -1), float(wgain))
    return _ConvBiasAct.apply(x, w, b, cfg)
