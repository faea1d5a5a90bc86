"""Per-sample channel scaling and its reduction gradient (the elementwise half of the modulated convolution).

``scale_nc(x, a, z)`` = ``x * a[:, :, None, None] (+ z)`` covers both `x * styles` and `fma(x, dcoefs, noise)` of the
reference's training-time modulated convolution (train_parts/generators.py:79-88); ``dot_hw(u, v)`` =
``(u * v).sum([2, 3])`` is its gradient w.r.t. the scale and, with ``v=None``, the bias-gradient reduction of bias_act.
The two Functions are each other's derivatives, so they are closed under differentiation (path-length regularisation
differentiates twice through the modulation).  Kernels: csrc/modulate.hip.
"""
import torch

from ... import _lib


def _layout(x):
    """(dense tensor, layout code): 1 = channel-minor, 0 = planar"""
    if x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1:
        return x.contiguous(memory_format=torch.channels_last), 1
    return x.contiguous(), 0


def _scale_nc_launch(x, a, z):
    lib = _lib.load()
    _lib.require_cuda(x, "scale_nc")
    x, layout = _layout(x)
    n, c, h, w = x.shape
    a = a.reshape(n, c).to(torch.float32).contiguous()
    zsn = 0
    if z is not None:
        z = z.to(torch.float32)
        if z.numel() == h * w:
            z = z.reshape(1, h * w).contiguous(); zsn = 0
        else:
            z = z.reshape(n, h * w).contiguous(); zsn = h * w
    y = torch.empty_like(x)
    if y.numel() == 0:          # empty batch: nothing to launch
        return y
    _lib.check(lib.sbg_scale_nc(_lib.ptr(x), _lib.ptr(a), _lib.ptr(z), _lib.ptr(y), _lib.dtype_code(x.dtype), layout,
                                n, c, h * w, zsn, _lib.stream_ptr(x.device)), "sbg_scale_nc")
    return y


def _dot_hw_launch(u, v):
    lib = _lib.load()
    _lib.require_cuda(u, "dot_hw")
    u, layout = _layout(u)
    if v is not None:
        v = v.to(u.dtype).contiguous(memory_format=torch.channels_last if layout == 1 else torch.contiguous_format)
    n, c, h, w = u.shape
    if u.numel() == 0:          # empty batch / image: the sum over nothing
        return torch.zeros([n, c], dtype=torch.float32, device=u.device)
    ns = lib.sbg_dot_hw_splits(layout, n, c, h * w)
    part = torch.empty([ns, n, c], dtype=torch.float32, device=u.device)
    _lib.check(lib.sbg_dot_hw(_lib.ptr(u), _lib.ptr(v), _lib.ptr(part), _lib.dtype_code(u.dtype), layout,
                              n, c, h * w, _lib.stream_ptr(u.device)), "sbg_dot_hw")
    return part.sum(0) if ns > 1 else part[0]


def _dot_hw_scale_launch(u, v, a):
    """(u * a[n, c], sum_hw u * v) from one pass over u and v (channel-minor, same shape), or None when the shapes do not fit that kernel"""
    lib = _lib.load()
    u, layout = _layout(u)
    n, c, h, w = u.shape
    if layout != 1 or u.dtype != v.dtype or u.numel() == 0 or not lib.sbg_dot_hw_scale_supported(c):
        return None
    v = v.contiguous(memory_format=torch.channels_last)
    a32 = a.detach().to(torch.float32).reshape(n, c).contiguous()
    y = torch.empty_like(u)
    ns = lib.sbg_dot_hw_splits(1, n, c, h * w)
    part = torch.empty([ns, n, c], dtype=torch.float32, device=u.device)
    _lib.check(lib.sbg_dot_hw_scale(_lib.ptr(u), _lib.ptr(v), _lib.ptr(a32), _lib.ptr(y), _lib.ptr(part), _lib.dtype_code(u.dtype),
                                    n, c, h * w, _lib.stream_ptr(u.device)), "sbg_dot_hw_scale")
    return y, (part[0] if ns == 1 else part.sum(0))


class _ScaleNC(torch.autograd.Function):
    """y = x * a[n, c] (+ z[n or 1, 1, h, w]);  a: fp32 [N, C]"""

    @staticmethod
    def forward(ctx, x, a, z):
        y = _scale_nc_launch(x, a, z)
        ctx.save_for_backward(x, a)
        ctx.z_shape = None if z is None else z.shape
        ctx.z_dtype = None if z is None else z.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        dx = da = dz = None
        both = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not torch.is_grad_enabled() and dy.dtype == x.dtype:
            both = _dot_hw_scale_launch(dy, x, a)            # first order: both gradients from one pass over (dy, x)
        if both is not None:
            dx, da = both[0], both[1].to(a.dtype).reshape(a.shape)
        else:
            if ctx.needs_input_grad[0]:
                dx = _ScaleNC.apply(dy, a, None)
            if ctx.needs_input_grad[1]:
                da = _DotHW.apply(dy, x).to(a.dtype).reshape(a.shape)
        if ctx.needs_input_grad[2]:
            dz = dy.sum(dim=1, keepdim=True, dtype=torch.float32)
            if ctx.z_shape[0] == 1 or len(ctx.z_shape) == 2:
                dz = dz.sum(dim=0, keepdim=True)
            dz = dz.reshape(ctx.z_shape).to(ctx.z_dtype)
        return dx, da, dz


class _DotHW(torch.autograd.Function):
    """r[n, c] = sum_{h,w} u * v  (fp32)"""

    @staticmethod
    def forward(ctx, u, v):
        ctx.save_for_backward(u, v)
        return _dot_hw_launch(u, v)

    @staticmethod
    def backward(ctx, dr):
        u, v = ctx.saved_tensors
        du = dv = None
        if ctx.needs_input_grad[0]:
            du = _ScaleNC.apply(v, dr, None) if v is not None else dr.to(u.dtype)[:, :, None, None].expand_as(u)
        if v is not None and ctx.needs_input_grad[1]:
            dv = _ScaleNC.apply(u, dr, None)
        return du, dv


class _MomentsHW(torch.autograd.Function):
    """(sum x, sum x^2) over (H, W) of a planar tensor in one pass: fp32 [N, C] each.  d/dx = ds1[n, c] + 2 x ds2[n, c] (a scale_shift_nc pass,
    itself differentiable)"""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _lib.require_cuda(x, "moments_hw")
        n, c, h, w = x.shape
        r = torch.empty([2, n, c], dtype=torch.float32, device=x.device)
        _lib.check(lib.sbg_moments_hw(_lib.ptr(x), _lib.ptr(r), _lib.dtype_code(x.dtype), n, c, h * w, _lib.stream_ptr(x.device)), "sbg_moments_hw")
        ctx.save_for_backward(x)
        return r[0], r[1]

    @staticmethod
    def backward(ctx, ds1, ds2):
        (x,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None
        return _ScaleShiftNC.apply(x, 2.0 * ds2, ds1)


def moments_hw(x):
    """x: [N, C, H, W] -> (x.sum([2, 3]), x.square().sum([2, 3])) in fp32; one pass for planar (contiguous NCHW) tensors, two dot_hw passes otherwise"""
    if x.is_contiguous() and x.numel() > 0:
        return _MomentsHW.apply(x)
    return dot_hw(x), dot_hw(x, x)


class _ScaleShiftNC(torch.autograd.Function):
    """y = x * a[n, c] + b[n, c];  a, b: fp32 [N, C]  (normalise-and-modulate step of the batch-norm layers)"""

    @staticmethod
    def forward(ctx, x, a, b):
        lib = _lib.load()
        _lib.require_cuda(x, "scale_shift_nc")
        xd, layout = _layout(x)
        n, c, h, w = xd.shape
        a32 = a.reshape(n, c).to(torch.float32).contiguous()
        b32 = b.reshape(n, c).to(torch.float32).contiguous()
        y = torch.empty_like(xd)
        _lib.check(lib.sbg_scale_shift_nc(_lib.ptr(xd), _lib.ptr(a32), _lib.ptr(b32), _lib.ptr(y), _lib.dtype_code(xd.dtype), layout,
                                          n, c, h * w, _lib.stream_ptr(xd.device)), "sbg_scale_shift_nc")
        ctx.save_for_backward(x, a)
        ctx.b_meta = (b.shape, b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        dx = da = db = None
        if ctx.needs_input_grad[0]:
            dx = _ScaleNC.apply(dy, a, None)
        if ctx.needs_input_grad[1]:
            da = _DotHW.apply(dy, x).to(a.dtype).reshape(a.shape)
        if ctx.needs_input_grad[2]:
            db = _DotHW.apply(dy, None).to(ctx.b_meta[1]).reshape(ctx.b_meta[0])
        return dx, da, db


def scale_shift_nc(x, a, b):
    """x: [N, C, H, W]; a, b: [N, C] -> x * a + b (per sample and channel)"""
    n, c = x.shape[0], x.shape[1]
    return _ScaleShiftNC.apply(x, a.reshape(n, c), b.reshape(n, c))


def scale_nc(x, a, z=None):
    """x: [N, C, H, W]; a: [N, C] (any float dtype, used in fp32); z: None, [N, 1, H, W] or [H, W] -> x * a (+ z)"""
    assert x.ndim == 4 and a.numel() == x.shape[0] * x.shape[1]
    return _ScaleNC.apply(x, a.reshape(x.shape[0], x.shape[1]), z)


def dot_hw(u, v=None):
    """u, v: [N, C, H, W] -> fp32 [N, C] = (u * v).sum([2, 3])  (v=None: u.sum([2, 3]))"""
    return _DotHW.apply(u, v)
