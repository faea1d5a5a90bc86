"""The discriminator's fromRGB layer as two streaming kernels.

``DiscriminatorBlock.fromrgb`` in the reference (train_parts/discriminators.py:270-277) is a ``Conv2dLayer`` with a 1x1 kernel on the
3 image channels: ``bias_act(conv2d(img.to(dtype), w * weight_gain), b, act='lrelu', clamp)``.  As matrix-core work K = 3 is ~95 %
padding and the image needs a cast + layout pass first; here the fp32 planar image is read as it is and the channel-minor 16-bit
activation is written once (csrc/fromrgb.hip); the backward pass produces the weight / bias gradients (and the image gradient when
the generator needs it) in one pass over ``(dy, y)``.

First order only.  R1 and the gradient penalty differentiate the discriminator twice with respect to its input, so the switch
``enabled`` is OFF by default (the layer then runs the arbitrarily differentiable ``conv2d_resample`` + ``bias_act`` composition) and
the loss code turns it on for the phases that carry no discriminator regulariser (train_parts/losses_base.py).
"""
import torch

from ... import _lib
from . import bias_act as _ba

enabled = False

_ACT = {"linear": 1, "relu": 2, "lrelu": 3}


def usable(img, weight, act, out_dtype):
    return (enabled and img.device.type == "cuda" and img.dtype == torch.float32 and img.ndim == 4 and out_dtype in (torch.bfloat16, torch.float16)
            and weight.ndim == 4 and weight.shape[2] == 1 and weight.shape[3] == 1 and act in _ACT
            and bool(_lib.load().sbg_fromrgb_supported(weight.shape[1], weight.shape[0], _ACT[act])))


class _FromRGB(torch.autograd.Function):
    """(img fp32 [N, Ci, H, W], w [Co, Ci] fp32, bias [Co] | None, cfg = (act, alpha, gain, clamp, out_dtype)) -> y [N, Co, H, W] channels_last"""

    @staticmethod
    def forward(ctx, img, w, bias, cfg):
        act, alpha, gain, clamp, out_dtype = cfg
        n, ci, h, wd = img.shape
        co = w.shape[0]
        ic = img.contiguous()
        wc = w.detach().to(torch.float32).contiguous()
        bc = bias.detach().to(torch.float32).contiguous() if bias is not None else None
        y = torch.empty([n, co, h, wd], dtype=out_dtype, device=img.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().sbg_fromrgb_fwd(ic.data_ptr(), wc.data_ptr(), _lib.ptr(bc), y.data_ptr(), _lib.dtype_code(out_dtype), n, ci, co, h * wd,
                                               _ACT[act], float(alpha), float(gain), float(clamp), _lib.stream_ptr(img.device)), "sbg_fromrgb_fwd")
        ctx.save_for_backward(ic, wc, y)
        ctx.cfg, ctx.has_bias = cfg, bias is not None
        ctx.dtypes = (w.dtype, bias.dtype if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        ic, wc, y = ctx.saved_tensors
        act, alpha, gain, clamp, out_dtype = ctx.cfg
        if torch.is_grad_enabled():
            raise RuntimeError("fromrgb: first-order only; set torch_utils.ops.fromrgb.enabled = False for double backward (R1, gradient penalty)")
        lib = _lib.load()
        n, ci, h, wd = ic.shape
        co = wc.shape[0]
        dyc = dy.to(y.dtype).contiguous(memory_format=torch.channels_last)
        dimg = torch.empty_like(ic) if ctx.needs_input_grad[0] else None
        nb = lib.sbg_fromrgb_bwd_blocks(n, h * wd)
        part = torch.empty([n * nb, co * ci + co], dtype=torch.float32, device=ic.device)
        _lib.check(lib.sbg_fromrgb_bwd(ic.data_ptr(), wc.data_ptr(), dyc.data_ptr(), y.data_ptr(), _lib.ptr(dimg), part.data_ptr(), _lib.dtype_code(y.dtype),
                                       n, ci, co, h * wd, _ACT[act], float(alpha), float(gain), float(clamp), _lib.stream_ptr(ic.device)), "sbg_fromrgb_bwd")
        sums = part.sum(0)                                              # fixed order
        dw = sums[:co * ci].reshape(co, ci).to(ctx.dtypes[0]) if ctx.needs_input_grad[1] else None
        db = sums[co * ci:].to(ctx.dtypes[1]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dimg, dw, db, None


def fromrgb(img, weight, bias, weight_gain, act, gain=None, clamp=None, out_dtype=torch.bfloat16):
    """clamp(act(conv1x1(img, weight * weight_gain) + bias) * gain) -> `out_dtype`, channels_last; weight: the layer's [Co, Ci, 1, 1] parameter"""
    spec = _ba.activation_funcs[act]
    w2d = weight.reshape(weight.shape[0], weight.shape[1]) * weight_gain
    cfg = (act, float(spec.def_alpha), float(gain if gain is not None else spec.def_gain), float(clamp if clamp is not None else -1), out_dtype)
    return _FromRGB.apply(img, w2d, bias, cfg)
