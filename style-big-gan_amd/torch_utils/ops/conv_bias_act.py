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
from . import upfirdn2d as _up

# Set by the loss orchestration (train_parts/losses_base.py) for passes that differentiate the discriminator ONCE (like ops/fromrgb.enabled):
# layer pairs may then run as first-order-only fused Functions (_ConvBiasActFir).  Off by default: direct callers get arbitrary-order autograd.
first_order = False


def _backward_from_dy(ctx, x, w, y, d1, db_fused, needs):
    """data / weight / bias gradients of conv_bias_act given d1 = gradient w.r.t. the convolution's output (pre-activation)"""
    stride, padding, act, alpha, gain, clamp, wgain = ctx.cfg
    dx = dw = db = None
    ccfg = (False, stride, padding, (0, 0), wgain)
    if needs[0]:
        op = _cg._output_padding_for(False, stride, padding, x.shape[2:], d1.shape[2:], w.shape[2:])
        dx = _cg._Conv.apply(d1, w, (True, stride, padding, op, wgain))
    if needs[1] and not _cg.weight_gradients_disabled:
        dw = _cg._ConvWgrad.apply(d1, x, ccfg, tuple(w.shape), _cg.wmeta_of(x, w))
    if ctx.has_bias and needs[2]:
        db = (db_fused if db_fused is not None else _ba._sum_to_bias(d1, 1)).to(ctx.b_dtype)
    return dx, dw, db


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
        dx, dw, db = _backward_from_dy(ctx, x, w, y, d1, db_fused, ctx.needs_input_grad)
        return dx, dw, db, None


class _ConvBiasActFir(torch.autograd.Function):
    """F = upfirdn2d(bias_act(conv2d(x, w) + b), f, padding) -- a discriminator block's conv0 followed by the low-pass of its down-sampling conv1
    (reference train_parts/discriminators.py:286-291 -> Conv2dLayer -> conv2d_resample.py:110-112).  Forward: the convolution with its fused epilogue,
    then the low-pass.  Backward, FIRST ORDER ONLY: the transposed low-pass multiplies its result by the activation's slope at the saved conv output and
    adds up the bias gradient in the same launch (upfirdn2d.fir_transposed_dact), so the gradient w.r.t. the activation's output -- a full-resolution
    tensor the unfused chain writes, then reads again together with the output -- never exists.
    cfg as _ConvBiasAct; fcfg = (padx0, padx1, pady0, pady1, flip_filter, gain) of the low-pass."""

    @staticmethod
    def forward(ctx, x, w, b, f, cfg, fcfg):
        stride, padding, act, alpha, gain, clamp, wgain = cfg
        epi = _cg.Epilogue(bias=b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        y = _cg._conv_forward(x, w, stride, padding, epi=epi, wgain=wgain)
        padx0, padx1, pady0, pady1, flip, fgain = fcfg
        out = _up._launch(y, f, 1, 1, 1, 1, padx0, padx1, pady0, pady1, flip, fgain)
        ctx.save_for_backward(x, w, y, f)
        ctx.cfg, ctx.fcfg = cfg, fcfg
        ctx.has_bias = b is not None
        ctx.b_dtype = b.dtype if b is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        if torch.is_grad_enabled():
            raise RuntimeError("conv_bias_act_fir: first-order only; leave torch_utils.ops.conv_bias_act.first_order = False for graphs that are differentiated twice")
        x, w, y, f = ctx.saved_tensors
        stride, padding, act, alpha, gain, clamp, wgain = ctx.cfg
        padx0, padx1, pady0, pady1, flip, fgain = ctx.fcfg
        ucfg = (1, 1, 1, 1, padx0, padx1, pady0, pady1, flip, fgain)
        fused = _up.fir_transposed_dact(dout, f, ucfg, (y.shape[2], y.shape[3]), y, act, alpha, gain, clamp)
        if fused is not None:
            d1, db_fused = fused
        else:       # the composition: transposed low-pass, then the activation gradient (+ bias sum) from (dy, y)
            fw, fh = _up._get_filter_size(f)
            gcfg = (1, 1, 1, 1, fw - padx0 - 1, y.shape[3] - dout.shape[3] + padx0, fh - pady0 - 1, y.shape[2] - dout.shape[2] + pady0, not flip, fgain)
            dy = _up._Upfirdn2d.apply(dout.to(y.dtype), f, gcfg).contiguous(memory_format=torch.channels_last)
            bcfg = _ba._Cfg((1, act, float(alpha), float(gain), float(clamp)))
            both = _ba._grad_and_bias_sum(dy, y, bcfg) if not bcfg.trivial else (dy, None)
            d1, db_fused = both if both is not None else (_ba._BiasActGrad.apply(dy, None, None, y, bcfg, torch.channels_last), None)
        dx, dw, db = _backward_from_dy(ctx, x, w, y, d1, db_fused, ctx.needs_input_grad)
        return dx, dw, db, None, None, None


def fir_fusable(x, w, act, f, groups=1):
    """may conv2d_bias_act(x, w) followed by a 4 x 4 low-pass f run as _ConvBiasActFir?  (16-bit channel-minor activations, 64-channel blocks,
    taps exact in the dtype: what the sliding-window matrix-core FIR needs; smaller images fall back inside the backward)"""
    return (first_order and fusable(x, w, act, groups) and x.dtype in (torch.bfloat16, torch.float16) and w.shape[0] % 64 == 0
            and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4) and x.shape[2] >= 16 and _up._taps_exact(f, x.dtype))


def conv2d_bias_act_fir(x, w, b, f, fpad, stride=1, padding=0, act="linear", alpha=None, gain=None, clamp=None, wgain=1.0, flip_filter=False, fgain=1.0):
    """upfirdn2d(conv2d_bias_act(x, w, b, ...), f, padding=fpad) as one first-order Function (see _ConvBiasActFir; check fir_fusable first)"""
    spec = _ba.activation_funcs[act]
    cfg = (_cg._pair(stride), _cg._pair(padding), act,
           float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain),
           float(clamp if clamp is not None else -1), float(wgain))
    padx0, padx1, pady0, pady1 = _up._parse_padding(fpad)
    return _ConvBiasActFir.apply(x, w, b, f, cfg, (padx0, padx1, pady0, pady1, bool(flip_filter), float(fgain)))


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
