"""Fused bias + activation + gain + clamp on MI355X.

Host-side mirror of the reference op ``stylegan2ada/torch_utils/ops/bias_act.py`` (public names
``activation_funcs`` :23-33 and ``bias_act`` :55-89 keep their meaning); the arithmetic runs in the HIP kernel
``sbg_bias_act`` (csrc/bias_act.hip).  First and second derivatives are autograd Functions built on the same
kernel (grad = 1, 2), so R1 / path-length double-backward works as in the reference (:129-210).
"""
import numpy as np
import torch

from ... import _lib


class _ActSpec(dict):
    """dict with attribute access (the reference exposes dnnlib.EasyDict entries)."""
    __getattr__ = dict.__getitem__


def _spec(cuda_idx, def_alpha, def_gain, ref, has_2nd_grad):
    return _ActSpec(def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref, has_2nd_grad=has_2nd_grad)


_SQRT2 = float(np.sqrt(2))
activation_funcs = {
    "linear":   _spec(1, 0.0, 1.0,    "",  False),
    "relu":     _spec(2, 0.0, _SQRT2, "y", False),
    "lrelu":    _spec(3, 0.2, _SQRT2, "y", False),
    "tanh":     _spec(4, 0.0, 1.0,    "y", True),
    "sigmoid":  _spec(5, 0.0, 1.0,    "y", True),
    "elu":      _spec(6, 0.0, 1.0,    "y", True),
    "selu":     _spec(7, 0.0, 1.0,    "y", True),
    "softplus": _spec(8, 0.0, 1.0,    "y", True),
    "swish":    _spec(9, 0.0, _SQRT2, "x", True),
}


class _Cfg(tuple):
    """(dim, act name, alpha, gain, clamp) -- hashable op configuration carried through autograd."""
    dim = property(lambda s: s[0]); act = property(lambda s: s[1]); alpha = property(lambda s: s[2])
    gain = property(lambda s: s[3]); clamp = property(lambda s: s[4])

    @property
    def spec(self):
        return activation_funcs[self.act]

    @property
    def trivial(self):
        return self.act == "linear" and self.gain == 1 and self.clamp < 0

    @property
    def needs_y(self):
        # The gradient's clamp mask is evaluated on the forward output.  The reference's CUDA plugin does not keep `y` for
        # 'linear' (ref=''), which makes its gradient ignore the clamp there; the reference's eager CPU path (the parity
        # target) zeroes the gradient of clamped elements for every activation, so `y` is kept for linear + clamp too.
        return "y" in self.spec.ref or (self.act == "linear" and self.clamp >= 0)


def _dense_like_format(x):
    """The memory format the kernel runs in: channels_last when x is 4-D channel-minor, else contiguous."""
    if x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1:
        return torch.channels_last
    return torch.contiguous_format


def _run(x, b, xref, yref, dy, grad, cfg, fmt):
    """One launch of sbg_bias_act. x is dense in `fmt`; aux tensors are brought to the same layout."""
    _lib.require_cuda(x, "bias_act")
    lib = _lib.load()
    aux = [t.contiguous(memory_format=fmt) if t is not None else None for t in (xref, yref, dy)]
    for t in aux:
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype):
            raise RuntimeError("bias_act: xref/yref/dy must have the same shape and dtype as x")
    y = torch.empty_like(x, memory_format=torch.preserve_format)
    if b is not None:
        if b.ndim != 1 or b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError("bias_act: b must be a 1-D tensor with the same dtype and device as x")
        if not (0 <= cfg.dim < x.ndim) or b.shape[0] != x.shape[cfg.dim]:
            raise RuntimeError("bias_act: b has wrong number of elements")
        b = b.contiguous()
        size_b, step_b = b.shape[0], x.stride(cfg.dim)
    else:
        size_b, step_b = 0, 1
    if x.numel() == 0:          # empty batch: nothing to launch (an empty tensor has no device pointer)
        return y
    status = lib.sbg_bias_act(_lib.ptr(x), _lib.ptr(b), _lib.ptr(aux[0]), _lib.ptr(aux[1]), _lib.ptr(aux[2]), _lib.ptr(y),
                              _lib.dtype_code(x.dtype), grad, cfg.spec.cuda_idx, cfg.alpha, cfg.gain, cfg.clamp,
                              x.numel(), size_b, max(step_b, 1), _lib.stream_ptr(x.device))
    _lib.check(status, "sbg_bias_act")
    return y


def _sum_to_bias(t, dim):
    """db = t summed over every dimension but `dim` (reference: bias_act.py:172-173); 4-D / dim 1 goes through the
    HIP reduction sbg_dot_hw (fp32 accumulation), everything else is small and uses torch."""
    if t.ndim == 4 and dim == 1 and t.device.type == "cuda":
        from . import modulate
        return modulate.dot_hw(t).sum(0).to(t.dtype)
    return t.sum([i for i in range(t.ndim) if i != dim])


_FUSED_ACTS = {"linear": 1, "relu": 2, "lrelu": 3}


def _grad_and_bias_sum(dy, y, cfg):
    """First-order backward of a piecewise-linear bias_act in ONE pass over (dy, y): returns (dx, db) with
    dx = dy * gain * (y > 0 ? 1 : alpha) * [|y| < clamp] and db = dx summed over N, H, W (fp32 accumulation, fixed order).
    Uses sbg_modconv_bwd with unit demodulation (csrc/modulate.hip); None when the layout / activation does not fit, in which
    case the caller takes the generic differentiable path (bias_act.py:159-210 of the reference)."""
    if (torch.is_grad_enabled() or cfg.act not in _FUSED_ACTS or cfg.dim != 1 or y.ndim != 4 or y.device.type != "cuda"
            or y.dtype not in (torch.bfloat16, torch.float16) or not y.is_contiguous(memory_format=torch.channels_last) or y.shape[1] == 1):
        return None
    lib = _lib.load()
    n, c, h, w = y.shape
    if not lib.sbg_modconv_bwd_supported(c):
        return None
    dy = dy.to(y.dtype).contiguous(memory_format=torch.channels_last)
    ones = torch.ones([n, c], dtype=torch.float32, device=y.device)
    ns = lib.sbg_dot_hw_splits(1, n, c, h * w)
    part = torch.empty([2, ns, n, c], dtype=torch.float32, device=y.device)
    dx = torch.empty_like(y)
    _lib.check(lib.sbg_modconv_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(ones), None, None, _lib.ptr(dx), _lib.ptr(part), None,
                                   _lib.dtype_code(y.dtype), n, c, h * w, 0, _FUSED_ACTS[cfg.act], float(cfg.alpha), float(cfg.gain),
                                   float(cfg.clamp), _lib.stream_ptr(y.device)), "sbg_modconv_bwd")
    return dx, part[0].sum([0, 1])


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, cfg):
        fmt = _dense_like_format(x)
        x = x.contiguous(memory_format=fmt)
        y = x if (cfg.trivial and b is None) else _run(x, b, None, None, None, 0, cfg, fmt)
        spec = cfg.spec
        keep_x = ("x" in spec.ref) or spec.has_2nd_grad
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if cfg.needs_y else None)
        ctx.cfg, ctx.fmt = cfg, fmt
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        cfg = ctx.cfg
        dx = db = None
        if ctx.needs_input_grad[1] and not cfg.trivial and y is not None and b is not None:
            fused = _grad_and_bias_sum(dy, y, cfg)          # first order, piecewise-linear activation: dx and db in one pass
            if fused is not None:
                return fused[0], fused[1].to(b.dtype), None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy.contiguous(memory_format=ctx.fmt)
            if not cfg.trivial:
                dx = _BiasActGrad.apply(dx, x, b, y, cfg, ctx.fmt)
        if ctx.needs_input_grad[1]:
            db = _sum_to_bias(dx, cfg.dim)
        return dx, db, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, cfg, fmt):
        dx = _run(dy, b, x, y, None, 1, cfg, fmt)
        ctx.save_for_backward(dy if cfg.spec.has_2nd_grad else None, x, b, y)
        ctx.cfg, ctx.fmt = cfg, fmt
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        cfg = ctx.cfg
        d_dx = d_dx.contiguous(memory_format=ctx.fmt)
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, cfg, ctx.fmt)
        if cfg.spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _run(d_dx, b, x, y, dy, 2, cfg, ctx.fmt)
            if ctx.needs_input_grad[2]:
                d_b = _sum_to_bias(d_x, cfg.dim)
        return d_dy, d_x, d_b, None, None, None


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="cuda"):
    """y = clamp(act(x + b) * gain).  Same arguments and defaults as the reference (bias_act.py:55).

    `impl` is accepted for signature compatibility; only the HIP implementation exists here ('cuda' selects it on a
    ROCm device exactly as in the reference).  `impl='ref'` and CPU tensors raise: the product has no CPU path."""
    assert isinstance(x, torch.Tensor)
    assert impl in ["ref", "cuda"]
    assert clamp is None or clamp >= 0
    if impl == "ref":
        raise RuntimeError("bias_act: impl='ref' is not part of the MI355X build (the CPU restatement is oracle/, test-only)")
    _lib.require_cuda(x, "bias_act")
    spec = activation_funcs[act]
    cfg = _Cfg((int(dim), act,
                float(alpha if alpha is not None else spec.def_alpha),
                float(gain if gain is not None else spec.def_gain),
                float(clamp if clamp is not None else -1)))
    return _BiasAct.apply(x, b, cfg)
