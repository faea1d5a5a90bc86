"""Training-time modulated convolution with its whole tail in the convolution kernel's epilogue.

The reference's synthesis layer runs ``modulated_conv2d`` (``x * styles`` -> convolution -> ``fma(x, dcoefs, noise)``,
train_parts/generators.py:79-88) and then ``bias_act`` (:328): three full passes over the activation besides the convolution.
Here the forward is ``scale_nc`` + ONE convolution launch whose epilogue applies demodulation, noise, bias, activation, gain and
clamp (csrc/conv_k64.hip), and the backward starts with ONE pass over ``(dy, y)`` (``sbg_modconv_bwd``, csrc/modulate.hip) that
produces the gradient fed to the data / weight gradient convolutions together with the bias, demodulation and noise gradients;
the pre-activation tensor is never materialised (the activation is piecewise linear, so it is recovered from the saved output).

First order only: path-length regularisation differentiates twice through the synthesis network, so trainers switch ``enabled``
off when a generator regulariser is configured and the layer falls back to the differentiable composition.
"""
import torch

from ... import _lib
from . import bias_act as _ba
from . import conv2d_gradfix as _cg
from . import modulate as _mod
from . import upfirdn2d as _up

enabled = True      # module switch: False -> SynthesisLayer uses modulated_conv2d + bias_act (arbitrarily differentiable)
import os as _os
chain_heads = _os.environ.get("SBG_CHAIN_HEADS", "1") != "0"       # a layer's backward also runs the backward head of the layer that feeds it (x_sole_consumer)

_ACT = {"linear": 1, "relu": 2, "lrelu": 3}


def usable(x, weight, act, up):
    return (enabled and up == 1 and act in _ACT and x.device.type == "cuda" and x.dtype in (torch.bfloat16, torch.float16)
            and weight.shape[2] == weight.shape[3] and bool(_lib.load().sbg_modconv_bwd_supported(weight.shape[0])))


class _ModConvBiasAct(torch.autograd.Function):
    """y = clamp(act(conv(x * s, w) * dcoefs + noise + b) * gain);  cfg = (padding, act, alpha, gain, clamp)"""

    @staticmethod
    def forward(ctx, x, w, styles, dcoefs, noise, b, cfg, x_tail=None):
        ctx.x_tail = x_tail          # upfirdn2d.TailHandle of the layer that produced x, when this layer is x's ONLY consumer (see backward)
        padding, act, alpha, gain, clamp = cfg
        xs = _mod._scale_nc_launch(x, styles, None)
        epi = _cg.Epilogue(oscale=dcoefs, noise=noise, bias=b, act=act, alpha=alpha, gain=gain, clamp=clamp)
        y = _cg._conv_forward(xs, w, (1, 1), (padding, padding), epi=epi)
        ctx.save_for_backward(x, xs, w, styles, dcoefs, noise, b, y)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xs, w, styles, dcoefs, noise, b, y = ctx.saved_tensors
        padding, act, alpha, gain, clamp = ctx.cfg
        if torch.is_grad_enabled():
            raise RuntimeError("modconv: the fused layer is first-order only; set torch_utils.ops.modconv.enabled = False "
                               "before building a graph that is differentiated twice (path-length regularisation)")
        n, cout, h, wd = y.shape
        want_dn = noise is not None and ctx.needs_input_grad[4]
        d2, sums, dn, dc32 = _up.backward_head(dy, y, dcoefs, noise, b, act, alpha, gain, clamp, want_dn)      # [2, N, Cout] sums, fixed order
        dx = dw = dstyles = ddcoefs = dnoise = db = None
        if ctx.needs_input_grad[3]:
            ddcoefs = (sums[1] / dc32).to(dcoefs.dtype).reshape(dcoefs.shape)
        if b is not None and ctx.needs_input_grad[5]:
            db = sums[0].sum(0).to(b.dtype)
        if want_dn:
            dnoise = dn if noise.numel() != h * wd else dn.sum(0, keepdim=True)
            dnoise = dnoise.reshape(noise.shape).to(noise.dtype)
        ccfg = (False, (1, 1), (padding, padding), (0, 0))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            dxs = _cg._Conv.apply(d2, w, (True, (1, 1), (padding, padding), (0, 0)))
            tail = ctx.x_tail
            both = None
            if (tail is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and dxs.dtype == x.dtype
                    and _lib.load().sbg_modconv_bwd_supported(x.shape[1])):
                # x = clamp(act(...)) is the output of an up-sampling layer's fused tail and feeds nothing but this convolution: `dxs * s`, `sum dxs * x`
                # and THAT layer's backward head (activation slope at x, bias / demodulation / noise sums) are one pass over (dxs, x).  What goes back
                # as "the gradient of x" is already the gradient of that layer's pre-activation; its backward recognises the tensor (TailHandle).
                hd2, hsums, hdn, hdc32, dst = _up.backward_head(dxs, x, tail.dcoefs, tail.noise, tail.b, tail.act, tail.alpha, tail.act_gain, tail.clamp,
                                                                tail.noise is not None, prescale=styles)
                tail.result = dict(ptr=hd2.data_ptr(), sums=hsums, dn=hdn, dc32=hdc32)
                dx, dstyles = hd2, dst.to(styles.dtype).reshape(styles.shape)
            else:
                if ctx.needs_input_grad[0] and ctx.needs_input_grad[2]:
                    both = _mod._dot_hw_scale_launch(dxs, x, styles)
                if both is not None:                         # dx = dxs * s and sum_hw dxs * x from one pass over (dxs, x)
                    dx, dstyles = both[0], both[1].to(styles.dtype).reshape(styles.shape)
                else:
                    if ctx.needs_input_grad[0]:
                        dx = _mod._scale_nc_launch(dxs, styles, None)
                    if ctx.needs_input_grad[2]:
                        dstyles = _mod._dot_hw_launch(dxs, x).to(styles.dtype).reshape(styles.shape)
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            dw = _cg._ConvWgrad.apply(d2, xs, ccfg, tuple(w.shape), _cg.wmeta_of(xs, w))
        return dx, dw, dstyles, ddcoefs, dnoise, db, None, None


class _DemodCoefs(torch.autograd.Function):
    """dcoefs[n, o] = rsqrt(sum_{i,kh,kw} (w[o,i,kh,kw] * s[n,i])^2 + 1e-8) from the fp32 parameter (generators.py:71-76): the sum over
    taps comes out of the (cached) operand-packing pass, the rest is one small kernel; backward = two small kernels + the
    gradient's `2 w dw2` written in the parameter's layout.  First order only, like the fused layers that consume it."""

    @staticmethod
    def forward(ctx, weight, styles, act_dtype):
        assert weight.dtype == torch.float32 and weight.ndim == 4
        o, i = weight.shape[0], weight.shape[1]
        n = styles.shape[0]
        _, w2 = _cg._packed_weight(weight, 0, act_dtype, (i + 7) // 8 * 8, 1.0)
        s32 = styles.detach().to(torch.float32).contiguous()
        d = torch.empty([n, o], dtype=torch.float32, device=weight.device)
        _lib.check(_lib.load().sbg_demod_coefs(s32.data_ptr(), w2.data_ptr(), d.data_ptr(), n, o, i, 1e-8, _lib.stream_ptr(weight.device)), "sbg_demod_coefs")
        ctx.save_for_backward(weight, s32, w2, d)
        ctx.s_dtype = styles.dtype
        return d

    @staticmethod
    def backward(ctx, g):
        weight, s32, w2, d = ctx.saved_tensors
        if torch.is_grad_enabled():
            raise RuntimeError("demod_coefs: first-order only; set torch_utils.ops.modconv.enabled = False for double backward")
        n, o = d.shape
        i = s32.shape[1]
        g = g.to(torch.float32).contiguous()
        ds = torch.empty_like(s32) if ctx.needs_input_grad[1] else None
        dw2 = torch.empty_like(w2) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().sbg_demod_coefs_bwd(g.data_ptr(), d.data_ptr(), s32.data_ptr(), w2.data_ptr(), _lib.ptr(ds), _lib.ptr(dw2), n, o, i,
                                                   _lib.stream_ptr(g.device)), "sbg_demod_coefs_bwd")
        dw = None
        if dw2 is not None:
            dw = _cg._unpack_wgrad(None, o, i, tuple(weight.shape), tuple(weight.stride()), 0, 1.0, w=weight.detach(), dw2=dw2)
        return dw, (ds.to(ctx.s_dtype) if ds is not None else None), None


def demod_coefs(weight, styles, act_dtype):
    """fused demodulation coefficients (see _DemodCoefs); weight: the layer's fp32 parameter, act_dtype: dtype of the layer's activations"""
    return _DemodCoefs.apply(weight, styles, act_dtype)


def modconv_bias_act(x, weight, styles, dcoefs, noise, bias, padding, act="lrelu", alpha=None, gain=None, clamp=None, x_sole_consumer=False):
    """Fused SynthesisLayer body (up = 1): x [N, Cin, H, W] 16-bit, weight [Cout, Cin, k, k] same dtype or the fp32 parameter, styles [N, Cin],
    dcoefs [N, Cout] (demodulation coefficients, differentiable), noise None / [N, 1, H, W] / [H, W], bias [Cout].
    `x_sole_consumer`: the caller guarantees that nothing else reads x; if x is the output of upfirdn2d.fir_bias_act (an up-sampling layer's fused tail),
    this layer's backward then also runs that layer's backward head, in the same pass as its own input gradients."""
    spec = _ba.activation_funcs[act]
    cfg = (int(padding), act, float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain),
           float(clamp if clamp is not None else -1))
    x_tail = getattr(x, "_sbg_tail", None) if (x_sole_consumer and chain_heads) else None
    return _ModConvBiasAct.apply(x, weight, styles, dcoefs, noise, bias, cfg, x_tail)
