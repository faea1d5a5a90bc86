"""The ToRGB layer's image work as two streaming kernels.

``ToRGBLayer.forward`` in the reference (train_parts/generators.py:344-348) is ``modulated_conv2d(x, weight, styles, demodulate=False)``
with a 1x1 kernel and 3 output channels, then a linear ``bias_act`` with clamp.  As matrix-core work that is ~97 % padding plus a
modulated copy of x; here x streams through once with per-sample weights ``wmod[n, o, c] = weight[o, c] * styles[n, c]`` (formed by
the caller with a framework op, so autograd splits ``d wmod`` into the weight and style gradients).  Output: fp32 planar, which is
what the skip-connection sum consumes (SynthesisBlock, :448-455).

First order only (like ops/modconv.py): the backward kernel is not itself differentiable; trainers switch ``modconv.enabled`` off
when a generator regulariser differentiates twice and the layer then runs the differentiable composition.
"""
import torch

from ... import _lib
from . import modconv


def usable(x, weight):
    return (modconv.enabled and x.device.type == "cuda" and x.dtype in (torch.bfloat16, torch.float16) and x.ndim == 4
            and weight.ndim == 4 and weight.shape[2] == 1 and weight.shape[3] == 1
            and bool(_lib.load().sbg_torgb_supported(weight.shape[1], weight.shape[0])))


class _ToRGB(torch.autograd.Function):
    """(x [N, C, H, W] 16-bit, wmod [N, O, C] fp32, bias [O] fp32 | None, clamp) -> y [N, O, H, W] fp32"""

    @staticmethod
    def forward(ctx, x, wmod, bias, clamp):
        n, c, h, w = x.shape
        o = wmod.shape[1]
        assert wmod.shape == (n, o, c)
        xc = x.contiguous(memory_format=torch.channels_last)
        wm = wmod.detach().to(torch.float32).contiguous()
        b32 = bias.detach().to(torch.float32).contiguous() if bias is not None else None
        y = torch.empty([n, o, h, w], dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().sbg_torgb_fwd(xc.data_ptr(), wm.data_ptr(), _lib.ptr(b32), y.data_ptr(), _lib.dtype_code(x.dtype), n, c, o, h * w,
                                             float(clamp), _lib.stream_ptr(x.device)), "sbg_torgb_fwd")
        ctx.save_for_backward(xc, wm, y)
        ctx.clamp, ctx.has_bias = float(clamp), bias is not None
        ctx.dtypes = (wmod.dtype, bias.dtype if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wm, y = ctx.saved_tensors
        if torch.is_grad_enabled():
            raise RuntimeError("torgb: first-order only; set torch_utils.ops.modconv.enabled = False for double backward")
        lib = _lib.load()
        n, c, h, w = xc.shape
        o = wm.shape[1]
        dy = dy.to(torch.float32).contiguous()
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        nb = lib.sbg_torgb_bwd_blocks(n, c, h * w)
        part = torch.empty([n, nb, o * c + o], dtype=torch.float32, device=xc.device)
        _lib.check(lib.sbg_torgb_bwd(xc.data_ptr(), wm.data_ptr(), dy.data_ptr(), y.data_ptr(), _lib.ptr(dx), part.data_ptr(), _lib.dtype_code(xc.dtype),
                                     n, c, o, h * w, ctx.clamp, _lib.stream_ptr(xc.device)), "sbg_torgb_bwd")
        sums = part.sum(1)                                              # fixed order
        dwmod = sums[:, :o * c].reshape(n, o, c).to(ctx.dtypes[0]) if ctx.needs_input_grad[1] else None
        db = sums[:, o * c:].sum(0).to(ctx.dtypes[1]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dwmod, db, None


def torgb(x, wmod, bias=None, clamp=None):
    return _ToRGB.apply(x, wmod, bias, -1.0 if clamp is None else float(clamp))
