"""FIR resampling (pad -> zero-insert upsample -> filter -> decimate) on MI355X.

Host-side mirror of ``stylegan2ada/torch_utils/ops/upfirdn2d.py`` (``setup_filter`` :72, ``upfirdn2d`` :120,
``filter2d`` :272, ``upsample2d`` :308, ``downsample2d`` :347): same names, arguments, padding arithmetic and
gradient definition (:214-268 -- the gradient of an upfirdn is the upfirdn with up/down swapped, the filter
flipped and the complementary padding, hence gradients of any order).  The arithmetic runs in ``sbg_upfirdn2d``
(csrc/upfirdn2d.hip) for both memory layouts.
"""
import numpy as np
import torch

from ... import _lib


def _pair(v):
    """int or [x, y] -> (x, y)"""
    if isinstance(v, int):
        v = [v, v]
    assert isinstance(v, (list, tuple)) and len(v) == 2 and all(isinstance(i, int) for i in v)
    sx, sy = v
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_scaling(scaling):
    return _pair(scaling)


def _parse_padding(padding):
    """int | [x, y] | [x0, x1, y0, y1] -> (x0, x1, y0, y1)"""
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(i, (int, np.integer)) for i in padding)
    padding = [int(i) for i in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    x0, x1, y0, y1 = padding
    return x0, x1, y0, y1


def _get_filter_size(f):
    """-> (fw, fh); None is the 1x1 identity."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare a float32 FIR filter: [fh, fw] (non-separable) or [taps] (separable; chosen automatically for
    1-D inputs with >= 8 taps).  None -> identity.  Same contract as the reference's ``setup_filter`` (:72-116)."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, tail=None, probe=False, dact=None):
    """One sbg_upfirdn2d launch on a rank-2 filter; output keeps x's memory format (reference: upfirdn2d.cpp:35).
    `tail` = dict(oscale, noise, bias, act, alpha, gain, clamp): fused demodulation / noise / bias_act epilogue (matrix-core FIR path only);
    `probe=True` returns whether that path would take this launch with a tail, without launching.
    `dact` = dict(y, act, alpha, gain, clamp): backward tail -- the result is multiplied by the slope of that bias_act at its saved output y and summed
    per channel; returns (result, bias gradient [C] fp32), or None when the sliding-window matrix-core FIR does not take the launch."""
    lib = _lib.load()
    if x.ndim != 4:
        raise RuntimeError("upfirdn2d: x must be rank 4")
    if f2d.dtype != torch.float32 or f2d.ndim != 2 or f2d.device != x.device:
        raise RuntimeError("upfirdn2d: f must be a float32 rank-2 tensor on the same device as x")
    n, c, ih, iw = x.shape
    fh, fw = f2d.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError("upfirdn2d: output must be at least 1x1")
    cl = x.stride(1) == 1 and c > 1
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if cl else torch.contiguous_format)
    if y.numel() == 0 and not probe:         # empty batch: nothing to launch
        return y
    p = _lib.UpfirdnParams()
    p.x, p.f, p.y = x.data_ptr(), f2d.data_ptr(), y.data_ptr()
    p.dtype = _lib.dtype_code(x.dtype)
    p.upx, p.upy, p.downx, p.downy, p.padx0, p.pady0 = upx, upy, downx, downy, padx0, pady0
    p.flip, p.gain = int(bool(flip)), float(gain)
    p.inSize[:] = [iw, ih, c, n]
    p.inStride[:] = [x.stride(3), x.stride(2), x.stride(1), x.stride(0)]
    p.filterSize[:] = [fw, fh]
    p.filterStride[:] = [f2d.stride(1), f2d.stride(0)]
    p.outSize[:] = [ow, oh, c, n]
    p.outStride[:] = [y.stride(3), y.stride(2), y.stride(1), y.stride(0)]
    p.filter_exact16 = int(x.dtype != torch.float32 and upx == upy == downx == downy == 1 and (fh, fw) == (4, 4) and _taps_exact(f2d, x.dtype))
    if probe:
        return bool(cl and lib.sbg_upfirdn2d_tail_supported(p))
    if dact is not None:
        ys = dact["y"]
        rows = int(lib.sbg_upfirdn2d_dact_rows(p)) if cl else -1
        if rows <= 0 or ys.shape != y.shape or ys.dtype != y.dtype or ys.stride() != y.stride():
            return None
        part = torch.empty([rows, 64], dtype=torch.float32, device=x.device)
        p.dact_y, p.dact_partial = ys.data_ptr(), part.data_ptr()
        p.dact_act = {"linear": 1, "relu": 2, "lrelu": 3}[dact["act"]]
        p.dact_alpha, p.dact_gain, p.dact_clamp = float(dact["alpha"]), float(dact["gain"]), float(dact["clamp"])
        _lib.check(lib.sbg_upfirdn2d(p, _lib.stream_ptr(x.device)), "sbg_upfirdn2d")
        return y, part.reshape(-1, c // 64, 64).sum(0).reshape(c)       # rows: (sample, segment, strip) x channel block; fixed order
    keep = []
    if tail is not None:
        def f32(t, shape):
            t = t.detach().to(torch.float32).reshape(shape).contiguous()
            keep.append(t)
            return t.data_ptr()
        if tail.get("oscale") is not None:
            p.oscale = f32(tail["oscale"], [n, c])
        if tail.get("bias") is not None:
            p.bias = f32(tail["bias"], [c])
        if tail.get("noise") is not None:
            nz = tail["noise"]
            per_sample = nz.numel() != oh * ow
            p.noise = f32(nz, [n if per_sample else 1, oh * ow])
            p.noise_stride_n = oh * ow if per_sample else 0
        p.act = {"linear": 1, "relu": 2, "lrelu": 3}[tail["act"]]
        p.alpha, p.act_gain, p.clamp = float(tail["alpha"]), float(tail["gain"]), float(tail["clamp"])
        if tail.get("post") is not None:            # (inference) the next layer's style modulation applied on the way out
            p.post_scale = f32(tail["post"], [n, c])
    _lib.check(lib.sbg_upfirdn2d(p, _lib.stream_ptr(x.device)), "sbg_upfirdn2d")
    return y


def _separable_fused_ok(x, f, upx, upy, downx, downy):
    """planar dense fp32 image, same factor on both axes: both passes of a rank-1 filter run in one launch (sbg_upfirdn2d_separable)"""
    return (x.dtype == torch.float32 and x.is_contiguous() and upx == upy and downx == downy and f.dtype == torch.float32
            and bool(_lib.load().sbg_upfirdn2d_separable_supported(upx, downx, f.numel())))


def _launch_separable(x, f, up, down, padx0, padx1, pady0, pady1, flip, gain):
    n, c, ih, iw = x.shape
    t = f.numel()
    ow = (iw * up + padx0 + padx1 - t) // down + 1
    oh = (ih * up + pady0 + pady1 - t) // down + 1
    if ow < 1 or oh < 1:
        raise RuntimeError("upfirdn2d: output would be empty")
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    fc = f.contiguous()
    if y.numel():
        _lib.check(_lib.load().sbg_upfirdn2d_separable(x.data_ptr(), fc.data_ptr(), y.data_ptr(), n * c, ih, iw, oh, ow, t, up, down,
                                                       padx0, pady0, int(bool(flip)), float(gain), _lib.stream_ptr(x.device)), "sbg_upfirdn2d_separable")
    return y


import weakref

_filter_facts = {}      # id(filter tensor) -> [weakref, version, sightings, {fact: value}]


def _filter_fact(f, name, compute, default, eager=False):
    """Cached property of a filter tensor that needs a device -> host read to establish.  Keyed on the tensor OBJECT (weak reference +
    version counter: a recycled id or an in-place edit starts over), not on its address -- the allocator hands the address of a dead
    temporary to the next one.  Unless `eager`, the read is made from the second sighting of an object on (module buffers); a filter built
    on the fly is a new object every call and gets `default` (the conservative path) without synchronising."""
    e = _filter_facts.get(id(f))
    if e is None or e[0]() is not f or e[1] != f._version:
        if len(_filter_facts) > 1024:
            for k in [k for k, v in _filter_facts.items() if v[0]() is None]:
                del _filter_facts[k]
        e = _filter_facts[id(f)] = [weakref.ref(f), f._version, 1, {}]
        if not eager:
            return default
    if name not in e[3]:
        e[3][name] = compute(f)
    return e[3][name]


def _taps_exact(f, dtype):
    """are all taps of f exactly representable in `dtype`?  (the matrix-core FIR path takes the filter in the tensor dtype.)"""
    # eager: asked only for 4x4 filters of 16-bit up = down = 1 launches (the resampling low-pass, a module buffer)
    return _filter_fact(f, ("exact", dtype), lambda t: bool((t.to(dtype).to(torch.float32) == t).all().item()), False, eager=True)


def _rank1_factor(f):
    """f [k, k] == outer(g, g)?  -> g (float32, on f's device) or None.  The reference keeps short filters 2-D (`setup_filter`: separable only
    from 8 taps, upfirdn2d.py:98-100), but [1, 3, 3, 1] x [1, 3, 3, 1] is an outer product: the planar fp32 launches (the RGB skip branch) can
    then take the fused separable kernel."""
    def compute(t):
        if t.ndim != 2 or t.shape[0] != t.shape[1] or t.shape[0] < 2:
            return None
        h = t.detach().to("cpu", torch.float64)
        j = int(torch.argmax(h.diagonal().abs()))
        if float(h[j, j]) <= 0:
            return None
        cand = h[:, j] / h[j, j].sqrt()
        if float((torch.outer(cand, cand) - h).abs().max()) > 1e-7 * float(h.abs().max()):
            return None
        return cand.to(torch.float32).to(t.device)
    return _filter_fact(f, "rank1", compute, None)


class _Upfirdn2d(torch.autograd.Function):
    """cfg = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)"""

    @staticmethod
    def forward(ctx, x, f, cfg):
        upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain = cfg
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        assert f.ndim in [1, 2]
        g1 = None
        if f.ndim == 2 and x.dtype == torch.float32 and x.is_contiguous() and upx == upy and downx == downy and (upx > 1 or downx > 1):
            g1 = _rank1_factor(f)
        if g1 is not None and _separable_fused_ok(x, g1, upx, upy, downx, downy):
            y = _launch_separable(x, g1, upx, downx, padx0, padx1, pady0, pady1, flip, gain)
        elif f.ndim == 2:
            y = _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
        elif _separable_fused_ok(x, f, upx, upy, downx, downy):
            y = _launch_separable(x, f, upx, downx, padx0, padx1, pady0, pady1, flip, gain)
        else:   # separable: a row pass then a column pass, sqrt(gain) each
            g = float(np.sqrt(gain))
            y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip, g)
            y = _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip, g)
        ctx.save_for_backward(f)
        ctx.cfg, ctx.in_hw = cfg, (x.shape[2], x.shape[3])
        return y

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain = ctx.cfg
        dx = None
        if ctx.needs_input_grad[0]:
            ih, iw = ctx.in_hw
            oh, ow = dy.shape[2], dy.shape[3]
            fw, fh = _get_filter_size(f)
            gcfg = (downx, downy, upx, upy,
                    fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1,
                    fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1,
                    not flip, gain)
            dx = _Upfirdn2d.apply(dy, f, gcfg)
        return dx, None, None


class TailHandle:
    """What the sole consumer of a fir_bias_act output needs in order to run THAT layer's backward head itself, fused with its own input
    gradients (ops/modconv.py: `dx = dxs * s`, `sum dxs * x` and the head below are three passes over the same two tensors), and the slot through
    which it hands the result back to _FirBiasAct.backward."""
    __slots__ = ("dcoefs", "noise", "b", "act", "alpha", "act_gain", "clamp", "result")

    def __init__(self, dcoefs, noise, b, act, alpha, act_gain, clamp):
        self.dcoefs, self.noise, self.b, self.act, self.alpha, self.act_gain, self.clamp = dcoefs, noise, b, act, alpha, act_gain, clamp
        self.result = None


def backward_head(dy, y, dcoefs, noise, b, act, alpha, act_gain, clamp, want_dn, prescale=None):
    """one pass over (dy, y) for y = clamp(act(c * dcoefs + noise + b) * act_gain): (d2 = gradient w.r.t. c, sums [2, N, C] = bias / demodulation
    partial sums, dn = per-pixel noise gradient or None, dc32) -- sbg_modconv_bwd.  With `prescale` [N, C]: dy is the gradient w.r.t. y * prescale; also
    returns sum_hw dy * y (the gradient of prescale) as a fifth value -- sbg_modconv_bwd_prescaled."""
    lib = _lib.load()
    n, c, oh, ow = y.shape
    dy = dy.to(y.dtype).contiguous(memory_format=torch.channels_last)
    dc32 = (dcoefs.detach().to(torch.float32).reshape(n, c) if dcoefs is not None else torch.ones([n, c], dtype=torch.float32, device=y.device)).contiguous()
    nz = nsn = None
    if noise is not None:
        nz = noise.detach().to(torch.float32)
        per_sample = nz.numel() != oh * ow
        nz = nz.reshape(n if per_sample else 1, oh * ow).contiguous()
        nsn = oh * ow if per_sample else 0
    b32 = b.detach().to(torch.float32).contiguous() if b is not None else None
    ns = lib.sbg_dot_hw_splits(1, n, c, oh * ow)
    part = torch.empty([2, ns, n, c], dtype=torch.float32, device=y.device)
    d2 = torch.empty_like(y)
    dn = torch.empty([n, 1, oh, ow], dtype=torch.float32, device=y.device) if want_dn else None
    code = {"linear": 1, "relu": 2, "lrelu": 3}[act]
    if prescale is None:
        _lib.check(lib.sbg_modconv_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dc32), _lib.ptr(nz), _lib.ptr(b32), _lib.ptr(d2), _lib.ptr(part),
                                       _lib.ptr(dn), _lib.dtype_code(y.dtype), n, c, oh * ow, nsn or 0, code,
                                       float(alpha), float(act_gain), float(clamp), _lib.stream_ptr(y.device)), "sbg_modconv_bwd")
        return d2, part.sum(1), dn, dc32
    ps32 = prescale.detach().to(torch.float32).reshape(n, c).contiguous()
    part3 = torch.empty([ns, n, c], dtype=torch.float32, device=y.device)
    _lib.check(lib.sbg_modconv_bwd_prescaled(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(ps32), _lib.ptr(dc32), _lib.ptr(nz), _lib.ptr(b32), _lib.ptr(d2),
                                             _lib.ptr(part), _lib.ptr(part3), _lib.ptr(dn), _lib.dtype_code(y.dtype), n, c, oh * ow, nsn or 0, code,
                                             float(alpha), float(act_gain), float(clamp), _lib.stream_ptr(y.device)), "sbg_modconv_bwd_prescaled")
    return d2, part.sum(1), dn, dc32, part3.sum(0)


class _FirBiasAct(torch.autograd.Function):
    """y = clamp(act(upfirdn2d(t, f, padding, gain) * dcoefs[n, c] + noise + b) * act_gain) in one kernel (up = down = 1; the low-pass after
    the transposed convolution of an up-sampling synthesis layer with the layer's whole tail, generators.py:84-88,328).  Backward: one pass
    over (dy, y) (sbg_modconv_bwd) gives the gradient w.r.t. the filtered tensor and the bias / demodulation / noise gradients, then the
    transposed FIR.  First order only (see ops/modconv.py).   cfg = (padx0, padx1, pady0, pady1, flip, gain, act, alpha, act_gain, clamp)"""

    @staticmethod
    def forward(ctx, t, f, dcoefs, noise, b, cfg, handle):
        padx0, padx1, pady0, pady1, flip, gain, act, alpha, act_gain, clamp = cfg
        y = _launch(t, f, 1, 1, 1, 1, padx0, padx1, pady0, pady1, flip, gain,
                    tail=dict(oscale=dcoefs, noise=noise, bias=b, act=act, alpha=alpha, gain=act_gain, clamp=clamp))
        ctx.save_for_backward(f, dcoefs, noise, b, y)
        ctx.cfg, ctx.in_hw, ctx.handle = cfg, (t.shape[2], t.shape[3]), handle
        return y

    @staticmethod
    def backward(ctx, dy):
        f, dcoefs, noise, b, y = ctx.saved_tensors
        padx0, padx1, pady0, pady1, flip, gain, act, alpha, act_gain, clamp = ctx.cfg
        if torch.is_grad_enabled():
            raise RuntimeError("fir_bias_act: first-order only; set torch_utils.ops.modconv.enabled = False for double backward")
        n, c, oh, ow = y.shape
        want_dn = noise is not None and ctx.needs_input_grad[3]
        handed = ctx.handle.result if ctx.handle is not None else None
        if handed is not None:      # the consumer of y ran this head together with its own input gradients (ops/modconv.py): dy IS d2
            ctx.handle.result = None
            if handed["ptr"] != dy.data_ptr() or dy.shape != y.shape:
                raise RuntimeError("fir_bias_act: the consumer handed over a backward head, but the incoming gradient is another tensor -- "
                                   "the output must have exactly that one consumer (TailHandle)")
            d2, sums, dn, dc32 = dy, handed["sums"], handed["dn"], handed["dc32"]
            want_dn = want_dn and dn is not None
        else:
            d2, sums, dn, dc32 = backward_head(dy, y, dcoefs, noise, b, act, alpha, act_gain, clamp, want_dn)
        dt = ddc = dnoise = db = None
        if dcoefs is not None and ctx.needs_input_grad[2]:
            ddc = (sums[1] / dc32).to(dcoefs.dtype).reshape(dcoefs.shape)
        if b is not None and ctx.needs_input_grad[4]:
            db = sums[0].sum(0).to(b.dtype)
        if want_dn:
            dnoise = (dn if noise.numel() != oh * ow else dn.sum(0, keepdim=True)).reshape(noise.shape).to(noise.dtype)
        if ctx.needs_input_grad[0]:
            ih, iw = ctx.in_hw
            fw, fh = _get_filter_size(f)
            gcfg = (1, 1, 1, 1, fw - padx0 - 1, iw - ow + padx0, fh - pady0 - 1, ih - oh + pady0, not flip, gain)
            dt = _Upfirdn2d.apply(d2, f, gcfg)
        return dt, None, ddc, dnoise, db, None, None


def fir_transposed_dact(dy, f, cfg, in_hw, y_saved, act, alpha, gain, clamp):
    """Backward of `bias_act -> upfirdn2d(f, cfg)` with respect to the bias_act's pre-activation, in ONE launch: the transposed low-pass of dy
    (what _Upfirdn2d.backward computes) times the activation's slope at the saved output, plus the bias gradient.  up = down = 1 only.
    Returns (gradient [like y_saved], db [C] fp32) or None when the launch does not fit the sliding-window matrix-core FIR."""
    upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, fgain = cfg
    if (upx, upy, downx, downy) != (1, 1, 1, 1) or f is None or f.ndim != 2 or dy.device.type != "cuda":
        return None
    ih, iw = in_hw
    fw, fh = _get_filter_size(f)
    oh, ow = dy.shape[2], dy.shape[3]
    dy = dy.to(y_saved.dtype).contiguous(memory_format=torch.channels_last)
    return _launch(dy, f, 1, 1, 1, 1, fw - padx0 - 1, iw - ow + padx0, fh - pady0 - 1, ih - oh + pady0, not flip, fgain,
                   dact=dict(y=y_saved, act=act, alpha=alpha, gain=gain, clamp=clamp))


def fir_tail_supported(x, f, padding, flip_filter=False):
    """can upfirdn2d(x, f, padding) with a fused tail run on the matrix-core FIR path?"""
    if f is None or f.ndim != 2 or x.device.type != "cuda" or x.ndim != 4:
        return False
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    return _launch(x, f, 1, 1, 1, 1, padx0, padx1, pady0, pady1, flip_filter, 1.0, probe=True)


def fir_bias_act(x, f, padding, gain, dcoefs, noise, b, act="lrelu", alpha=0.2, act_gain=1.0, clamp=-1.0, flip_filter=False, post_scale=None):
    """`post_scale` [N, C] (inference only: no graph is recorded): the result times post_scale[n, c], i.e. already modulated for the layer that reads it"""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    if post_scale is not None:
        assert not (torch.is_grad_enabled() and (x.requires_grad or (dcoefs is not None and dcoefs.requires_grad))), "fir_bias_act: post_scale is an inference-only extension"
        return _launch(x, f, 1, 1, 1, 1, padx0, padx1, pady0, pady1, bool(flip_filter), float(gain),
                       tail=dict(oscale=dcoefs, noise=noise, bias=b, act=act, alpha=float(alpha), gain=float(act_gain), clamp=float(clamp), post=post_scale))
    cfg = (padx0, padx1, pady0, pady1, bool(flip_filter), float(gain), act, float(alpha), float(act_gain), float(clamp))
    handle = TailHandle(dcoefs, noise, b, act, float(alpha), float(act_gain), float(clamp))
    y = _FirBiasAct.apply(x, f, dcoefs, noise, b, cfg, handle)
    y._sbg_tail = handle        # a consumer that KNOWS it is the only one may take over this layer's backward head (ops/modconv.py, x_tail)
    return y


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Pad, upsample, filter and downsample a batch of 2-D images [N, C, H, W] (reference: upfirdn2d.py:120).
    padding is w.r.t. the upsampled image, negative = crop; flip_filter False = convolution, True = correlation."""
    assert isinstance(x, torch.Tensor)
    assert impl in ["ref", "cuda"]
    if impl == "ref":
        raise RuntimeError("upfirdn2d: impl='ref' is not part of the MI355X build (the CPU restatement is oracle/, test-only)")
    _lib.require_cuda(x, "upfirdn2d")
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    assert f is None or (isinstance(f, torch.Tensor) and f.dtype == torch.float32 and not f.requires_grad)
    cfg = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, bool(flip_filter), float(gain))
    return _Upfirdn2d.apply(x, f, cfg)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """FIR-filter keeping the input size (extra `padding` on top).  Reference: upfirdn2d.py:272-304."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Upsample by `up`; output size = up * input size (+ padding).  Reference: upfirdn2d.py:308-343."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="cuda"):
    """Downsample by `down`; output size = input size / down (+ padding).  Reference: upfirdn2d.py:347-382."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
