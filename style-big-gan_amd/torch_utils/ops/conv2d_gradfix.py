"""conv2d / conv_transpose2d with arbitrarily differentiable gradients, on the MI355X matrix cores.

Host-side mirror of ``stylegan2ada/torch_utils/ops/conv2d_gradfix.py``: same public names (``enabled``,
``weight_gradients_disabled``, ``no_weight_gradients()``, ``conv2d``, ``conv_transpose2d`` -- :22-45) and the same
autograd structure (:107-165): the data gradient of a convolution is the opposite (transposed / plain)
convolution, the weight gradient is its own Function whose backward is again made of convolutions, so R1 and
path-length double-backward work.  Where the reference calls cuDNN, this module launches the hand-written
implicit-GEMM kernels ``sbg_conv2d_igemm`` / ``sbg_conv2d_wgrad`` (csrc/conv_igemm.hip, csrc/conv_k64.hip, csrc/conv_wgrad.hip).

Layout / precision: activations are processed channel-minor (``torch.channels_last``; other layouts are converted),
bf16 / f16 tensors take one MFMA pass with fp32 accumulation; fp32 tensors are split into three bf16 parts (hi + mid + lo =
24 mantissa bits) and take six MFMA passes accumulated in fp32 (every product term above 2^-24; still ~2.7x the throughput of
the fp32-input MFMA, which runs at 1/16 of the bf16 rate on gfx950).  ``fp32_mfma_passes = 3`` selects a cheaper hi/lo split.
"""
import contextlib

import weakref

import numpy as np

import torch

from ... import _lib

enabled = False                     # kept for API parity; the HIP path is always used on a ROCm device
weight_gradients_disabled = False   # forcefully disable computation of gradients with respect to the weights
import os as _os0
fp32_mfma_passes = int(_os0.environ.get('SBG_FP32_PASSES', '6'))     # fp32 tensors: 6 = bf16 hi/mid/lo split (~fp32 accuracy), 3 = hi/lo split (rel. error ~1e-5 per product); experiment switch


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def _pair(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert len(v) == 2 and all(isinstance(i, int) for i in v)
    return v


# ----------------------------------------------------------------------------------------------------------------
# kernel launch helpers (no autograd)

def _cl(t):
    """channel-minor (NHWC in memory) dense copy/view of a [N, C, H, W] tensor"""
    return t.contiguous(memory_format=torch.channels_last)


def _pad_channels(t, mult=8):
    """zero-pad dim 1 of a [N, C, H, W] tensor up to a multiple of `mult` (channel-minor result)"""
    c = t.shape[1]
    cp = (c + mult - 1) // mult * mult
    if cp == c:
        return _cl(t)
    out = torch.empty([t.shape[0], cp, t.shape[2], t.shape[3]], dtype=t.dtype, device=t.device, memory_format=torch.channels_last).zero_()
    out[:, :c] = t
    return out


def _split_bf16(t, parts):
    """t (fp32) ~= sum of `parts` bf16 tensors (8 mantissa bits each)"""
    out, rest = [], t
    for _ in range(parts):
        h = rest.to(torch.bfloat16)
        out.append(h)
        rest = rest - h.to(torch.float32)
    return out


def _operand_passes(a, b):
    """[(a_part, b_part), ...] MFMA passes for a product of two tensors of equal dtype (smallest terms first)."""
    if a.dtype in (torch.bfloat16, torch.float16):
        return [(a, b)]
    if a.dtype == torch.float32:
        if fp32_mfma_passes >= 6:
            a0, a1, a2 = _split_bf16(a, 3)
            b0, b1, b2 = _split_bf16(b, 3)
            return [(a2, b0), (a0, b2), (a1, b1), (a1, b0), (a0, b1), (a0, b0)]
        a0, a1 = _split_bf16(a, 2)
        b0, b1 = _split_bf16(b, 2)
        return [(a1, b0), (a0, b1), (a0, b0)]
    raise RuntimeError(f"conv2d: unsupported dtype {a.dtype}")


class Epilogue:
    """fused tail of a convolution launch: y = clamp(act(acc * oscale[n, co] + noise[n, pixel] + bias[co]) * gain)"""
    __slots__ = ("oscale", "noise", "bias", "act", "alpha", "gain", "clamp", "_keep")

    def __init__(self, oscale=None, noise=None, bias=None, act="linear", alpha=0.0, gain=1.0, clamp=-1.0):
        assert act in ("linear", "relu", "lrelu")
        self.oscale, self.noise, self.bias = oscale, noise, bias
        self.act, self.alpha, self.gain, self.clamp = {"linear": 1, "relu": 2, "lrelu": 3}[act], float(alpha), float(gain), float(clamp)
        self._keep = []

    def fill(self, p, n, cout, oh, ow):
        def f32(t, shape):
            t = t.detach().to(torch.float32).reshape(shape).contiguous()
            self._keep.append(t)
            return t
        if self.oscale is not None:
            p.oscale = f32(self.oscale, [n, cout]).data_ptr()
        if self.bias is not None:
            p.bias = f32(self.bias, [cout]).data_ptr()
        if self.noise is not None:
            per_sample = self.noise.numel() != oh * ow
            nz = f32(self.noise, [n if per_sample else 1, oh * ow])
            p.noise, p.noise_stride_n = nz.data_ptr(), (oh * ow if per_sample else 0)
        p.act, p.alpha, p.gain, p.clamp = self.act, self.alpha, self.gain, self.clamp


import os as _os
_KSPLIT_MAX_TILES = int(_os.environ.get('SBG_KSPLIT_MAX_TILES', '128'))     # experiment switches for the K split of few-tile launches
_KSPLIT_TARGET = int(_os.environ.get('SBG_KSPLIT_TARGET', '256'))

# fp32 operands up to this many elements run as ONE launch over the concatenated hi / mid / lo parts (each operand built by one pass of
# sbg_split_bf16_cat: 4 B read + 12 B written per element).  The alternative -- six launches that read-modify-write the fp32 output five times,
# fed by parts split with framework ops -- moves ~4x the bytes on large activations (BigGAN at 128x128: [48, 64, 128, 128] tensors; half of
# that workload's step was split / accumulate traffic) and is launch-bound on small ones; it remains only as the path for operands whose
# concatenated copy would exceed a few GB.
CONCAT_NUMEL = 1 << 28


_ORDER6 = ((2, 0, 1, 1, 0, 0), (0, 2, 1, 0, 1, 0))      # part indices of (a, b) in the six products of _operand_passes, smallest terms first
_ORDER3 = ((1, 0, 0), (0, 1, 0))


def _split_cat(t, dim, order, dense=False):
    """fp32 `t` -> bf16 tensor with `len(order)` times the size along `dim`: the hi / mid / lo parts `order[s]` of t side by side, in ONE
    kernel.  By default the result keeps t's memory order (channel-minor stays channel-minor: sbg_split_bf16_cat); `dense` asks for a
    contiguous result whatever t's strides are (sbg_split_bf16_cat_nd: a permuted weight view becomes the packed operand without the
    transposing copy of six times the data that `.contiguous()` would add)."""
    import ctypes
    arr = (ctypes.c_int * len(order))(*order)
    if dense and t.ndim <= 4 and not t.is_contiguous():
        lead = 4 - t.ndim
        shape = [1] * lead + list(t.shape)
        out = torch.empty([len(order) * n if d == dim else n for d, n in enumerate(t.shape)], dtype=torch.bfloat16, device=t.device)
        _lib.check(_lib.load().sbg_split_bf16_cat_nd(_lib.ptr(t), (ctypes.c_int64 * 4)(*shape), (ctypes.c_int64 * 4)(*([0] * lead + list(t.stride()))),
                                                     lead + dim, _lib.ptr(out), len(order), arr, _lib.stream_ptr(t.device)), "sbg_split_bf16_cat_nd")
        return out
    perm = sorted(range(t.ndim), key=lambda d: (-t.stride(d), d))
    tt = t.permute(perm)
    if not tt.is_contiguous():
        tt = tt.contiguous()
    k = perm.index(dim)
    outer = int(np.prod(tt.shape[:k], dtype=np.int64)) if k > 0 else 1
    C, inner = tt.shape[k], (int(np.prod(tt.shape[k + 1:], dtype=np.int64)) if k + 1 < tt.ndim else 1)
    out = torch.empty(list(tt.shape[:k]) + [len(order) * C] + list(tt.shape[k + 1:]), dtype=torch.bfloat16, device=t.device)
    _lib.check(_lib.load().sbg_split_bf16_cat(_lib.ptr(tt), _lib.ptr(out), outer, C, inner, len(order), arr, _lib.stream_ptr(t.device)), "sbg_split_bf16_cat")
    inv = [perm.index(d) for d in range(t.ndim)]
    return out.permute(inv)


def _mfma_operands(a, b, a_cat_dim, b_cat_dim, b_dense=False):
    """[(a_k, b_k)]: the matrix-core launches whose sum is the product of a and b (see _operand_passes); small fp32 operands become ONE launch
    over concatenated hi / mid / lo parts, each operand built by one kernel (`b_dense`: b is a weight view wanted contiguous)"""
    if (a.dtype == torch.float32 and b.dtype == torch.float32 and a.device.type == "cuda" and a.numel() <= CONCAT_NUMEL and a.numel() > 0 and b.numel() > 0):
        oa, ob = _ORDER6 if fp32_mfma_passes >= 6 else _ORDER3
        return [(_split_cat(a, a_cat_dim, oa), _split_cat(b, b_cat_dim, ob, dense=b_dense))]
    return _fold_passes(_operand_passes(a, b), a_cat_dim, b_cat_dim)


def _fold_passes(passes, x_cat_dim, w_cat_dim):
    """[(a_k, b_k)] -> [(cat a_k, cat b_k)]: sum_k conv(a_k, b_k) == conv over the concatenated reduction axis.  For small
    tensors (the 4x4 / 8x8 fp32 blocks) this turns six launch-bound passes with fp32 read-modify-write outputs into one."""
    if len(passes) == 1 or passes[0][0].numel() > CONCAT_NUMEL:
        return passes
    a = torch.cat([a_ for a_, _ in passes], dim=x_cat_dim)
    b = torch.cat([b_ for _, b_ in passes], dim=w_cat_dim)
    if x_cat_dim == 1:
        a = a.contiguous(memory_format=torch.channels_last)
    return [(a, b)]


# ----------------------------------------------------------------------------------------------------------------
# fp32 master weights with 16-bit activations ("mixed" launches): the weight operand is cast, scaled and packed by ONE kernel
# (sbg_pack_weight) instead of `w * gain` -> `.to(dtype)` -> permute/contiguous, and cached while the parameter is unchanged
# (a parameter is used by several forward / data-gradient launches between two optimizer steps).

_pack_cache = {}
_ptr_epoch = {}         # parameter storage -> number of optimizer steps that touched it


def _on_optimizer_step(optimizer, args, kwargs):
    """Fused optimizers (torch.optim.Adam(fused=True), ...) update parameters WITHOUT bumping their version counters, so the version alone
    cannot validate a cached operand: every optimizer step also advances a per-parameter epoch (global post-step hook, any optimizer)."""
    for group in optimizer.param_groups:
        for q in group["params"]:
            if isinstance(q, torch.Tensor) and q.device.type == "cuda":
                k = q.data_ptr()
                _ptr_epoch[k] = _ptr_epoch.get(k, 0) + 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook      # noqa: E402
_register_step_hook(_on_optimizer_step)


def invalidate_packed_weights():
    """Drop every cached operand.  The cache follows parameter updates through the tensor version counter (any in-place op, `copy_`,
    `load_state_dict`, foreach optimizers) and the optimizer-step hook above (fused optimizers); writes that bypass both -- in-place ops on
    `param.data`, raw pointer writes from other libraries -- must be followed by this call."""
    _pack_cache.clear()


def is_mixed(x, w):
    return w.dtype == torch.float32 and x.dtype in (torch.bfloat16, torch.float16)


def _packed_weight(w, rows_dim, dtype, bp, gain):
    """w: fp32 [d0, d1, kh, kw] (any strides) -> [kh*kw, A, bp] in `dtype`, rows A = dim `rows_dim`, columns = the other of the first
    two dims, zero-padded to bp, values cast(w * gain); also w2 [A, B] = sum over taps of (w * gain)^2 for rows_dim == 0 (the
    demodulation's reduction, free in the same pass).  Cached for (views of) parameters, keyed on the version counter."""
    assert w.dtype == torch.float32 and w.ndim == 4 and rows_dim in (0, 1)
    a_dim, b_dim = rows_dim, 1 - rows_dim
    A, B, kh, kw = w.shape[a_dim], w.shape[b_dim], w.shape[2], w.shape[3]
    base = w._base if w._base is not None else w
    cacheable = isinstance(base, torch.nn.Parameter)
    key = None
    if cacheable:
        key = (base.data_ptr(), w.storage_offset(), tuple(w.shape), tuple(w.stride()), rows_dim, dtype, bp, float(gain))
        hit = _pack_cache.get(key)
        stamp = (base._version, _ptr_epoch.get(base.data_ptr(), 0))
        if hit is not None and hit[0]() is base and hit[1] == stamp:
            return hit[2], hit[3]
    out = torch.empty([kh * kw, A, bp], dtype=dtype, device=w.device)
    w2 = torch.empty([A, B], dtype=torch.float32, device=w.device) if rows_dim == 0 else None
    wd = w.detach()
    _lib.check(_lib.load().sbg_pack_weight(wd.data_ptr(), out.data_ptr(), _lib.dtype_code(dtype), A, B, kh, kw, wd.stride(a_dim), wd.stride(b_dim),
                                           wd.stride(2), wd.stride(3), bp, float(gain), _lib.ptr(w2), _lib.stream_ptr(w.device)), "sbg_pack_weight")
    if cacheable:
        if len(_pack_cache) > 256:          # drop the operands of parameters that no longer exist
            for k in [k for k, v in _pack_cache.items() if v[0]() is None]:
                del _pack_cache[k]
            if len(_pack_cache) > 2048:
                _pack_cache.clear()
        _pack_cache[key] = (weakref.ref(base), (base._version, _ptr_epoch.get(base.data_ptr(), 0)), out, w2)
    return out, w2


def _unpack_wgrad(out, ca, cb, wshape, wstride, rows_dim, gain, w=None, dw2=None):
    """out: fp32 [taps, cap, cbp] from _wgrad (or None) -> fp32 gradient with the parameter's shape and strides:
    dw[.., t] = gain * out[t, a, b] (+ 2 gain^2 w * dw2[a, b])."""
    dev = out.device if out is not None else w.device
    dw = torch.empty_strided(wshape, wstride, dtype=torch.float32, device=dev)
    a_dim, b_dim = rows_dim, 1 - rows_dim
    assert wshape[a_dim] == ca and wshape[b_dim] == cb
    _lib.check(_lib.load().sbg_unpack_wgrad(_lib.ptr(out), out.stride(0) if out is not None else 0, out.stride(1) if out is not None else 0,
                                            dw.data_ptr(), _lib.ptr(w), _lib.ptr(dw2), ca, cb, wshape[2], wshape[3],
                                            wstride[a_dim], wstride[b_dim], wstride[2], wstride[3], float(gain), _lib.stream_ptr(dev)), "sbg_unpack_wgrad")
    return dw


def _igemm(x, wp, y, taps, stride, oh, ow, y_off=(0, 0), y_step=(1, 1), oscale=None, accumulate=False, epi=None):
    """x: [N, Cin, IH, IW] channel-minor 16-bit; wp: packed [slabs, Cout, Cin]; y: [N, Cout, YH, YW] channel-minor.
    Writes y[:, :, y_off[0] + y_step[0]*oy, y_off[1] + y_step[1]*ox] for oy < oh, ox < ow."""
    lib = _lib.load()
    p = _lib.ConvParams()
    n, cin, ih, iw = x.shape
    cout = wp.shape[1]
    assert wp.shape[2] == cin and x.stride(1) == 1 and y.stride(1) == 1 and wp.is_contiguous() and x.dtype == wp.dtype
    es = y.element_size()
    p.x, p.w = x.data_ptr(), wp.data_ptr()
    p.y = y.data_ptr() + (y_off[0] * y.stride(2) + y_off[1] * y.stride(3)) * es
    p.oscale = oscale.data_ptr() if oscale is not None else None
    p.xdtype, p.ydtype = _lib.dtype_code(x.dtype), _lib.dtype_code(y.dtype)
    p.N, p.IH, p.IW, p.Cin, p.Cout, p.OH, p.OW = n, ih, iw, cin, cout, oh, ow
    p.xs_n, p.xs_h, p.xs_w = x.stride(0), x.stride(2), x.stride(3)
    p.ys_n, p.ys_h, p.ys_w = y.stride(0), y.stride(2) * y_step[0], y.stride(3) * y_step[1]
    p.ws_slab, p.ws_co = wp.stride(0), wp.stride(1)
    p.stride, p.ntaps = stride, len(taps)
    assert 1 <= len(taps) <= _lib.SBG_MAX_TAPS
    for i, (dy, dx, slab) in enumerate(taps):
        p.tap_dy[i], p.tap_dx[i], p.tap_slab[i] = dy, dx, slab
    p.accumulate = int(accumulate)
    p.act, p.alpha, p.gain, p.clamp = 1, 0.0, 1.0, -1.0
    if epi is not None:
        assert not accumulate and y_step == (1, 1)
        epi.fill(p, n, cout, oh, ow)
    # few output tiles but a long reduction (the 4x4 fp32 block: 16 tiles x 432 K-steps; the 16-bit 4x4 .. 8x8 layers: 16-64 tiles x 72): split K over
    # workgroups; the fixed-order slab reduction applies the fused epilogue and the output cast
    ws = None
    tiles = -(-(n * oh * ow) // 128) * -(-cout // 128)
    ksteps = len(taps) * -(-cin // 64)
    # (an odd channel count -- the 513-channel data gradient of the discriminator's epilogue convolution -- splits too when the result is a plain
    # fp32 sum: only the fused-epilogue reduction stores 8-channel vectors)
    plain_f32 = epi is None and y.dtype == torch.float32
    if ((epi is None or not accumulate) and (y.dtype == torch.float32 or not accumulate) and y_step == (1, 1) and y_off == (0, 0) and cout > 64
            and (cout % 8 == 0 or plain_f32) and tiles < _KSPLIT_MAX_TILES and ksteps >= 16 and y.is_contiguous(memory_format=torch.channels_last)):
        p.ksplit = max(1, min(ksteps // 4, -(-_KSPLIT_TARGET // tiles)))
        if p.ksplit > 1:
            ws = torch.empty([lib.sbg_conv2d_igemm_workspace(p) // 4], dtype=torch.float32, device=x.device)
            p.workspace = ws.data_ptr()
    _lib.check(lib.sbg_conv2d_igemm(p, _lib.stream_ptr(x.device)), "sbg_conv2d_igemm")


def _igemm_phases(x, wp, y, phases, s):
    """ONE launch for all phases of a stride-s transposed convolution (sbg_conv_params.nphase): phase (a, b) writes
    y[:, :, a::s, b::s] from its own run of taps; the phases' tiles share the input through L2."""
    lib = _lib.load()
    p = _lib.ConvParams()
    n, cin, ih, iw = x.shape
    cout = wp.shape[1]
    p.x, p.w, p.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    p.xdtype, p.ydtype = _lib.dtype_code(x.dtype), _lib.dtype_code(y.dtype)
    p.N, p.IH, p.IW, p.Cin, p.Cout = n, ih, iw, cin, cout
    p.OH, p.OW = max(ph[2] for ph in phases), max(ph[3] for ph in phases)
    p.xs_n, p.xs_h, p.xs_w = x.stride(0), x.stride(2), x.stride(3)
    p.ys_n, p.ys_h, p.ys_w = y.stride(0), y.stride(2) * s, y.stride(3) * s
    p.ws_slab, p.ws_co = wp.stride(0), wp.stride(1)
    p.stride = 1
    p.act, p.alpha, p.gain, p.clamp = 1, 0.0, 1.0, -1.0
    t = 0
    for i, (a, b, goh, gow, taps) in enumerate(phases):
        p.ph_ntaps[i], p.ph_oh[i], p.ph_ow[i] = len(taps), goh, gow
        p.ph_yoff[i] = a * y.stride(2) + b * y.stride(3)
        for dy, dx, slab in taps:
            p.tap_dy[t], p.tap_dx[t], p.tap_slab[t] = dy, dx, slab
            t += 1
    p.ntaps, p.nphase = t, len(phases)
    _lib.check(lib.sbg_conv2d_igemm(p, _lib.stream_ptr(x.device)), "sbg_conv2d_igemm")


def _launch_groups(taps):
    """split a tap list into launches of at most SBG_MAX_TAPS taps"""
    m = _lib.SBG_MAX_TAPS
    return [taps[i:i + m] for i in range(0, len(taps), m)]


def epilogue_fusable(x):
    """the fused epilogue rides on single-pass (16-bit) launches"""
    return x.dtype in (torch.bfloat16, torch.float16)


def _conv_forward(x, w, stride, padding, epi=None, wgain=1.0):
    """y[n,co,oy,ox] = sum x[n,ci,oy*s+kh-p,ox*s+kw-p] w[co,ci,kh,kw] (correlation, like F.conv2d); `epi`: fused Epilogue.
    w may be the fp32 master weight with 16-bit x: the operand is then cast(w * wgain), packed by one kernel."""
    n, cin, ih, iw = x.shape
    cout, cin_w, kh, kw = w.shape
    mixed = is_mixed(x, w)
    assert cin == cin_w and (x.dtype == w.dtype or mixed) and (mixed or wgain == 1.0)
    (sh, sw), (ph, pw) = stride, padding
    assert sh == sw, "conv2d: only square strides are implemented"
    oh = (ih + 2 * ph - kh) // sh + 1
    ow = (iw + 2 * pw - kw) // sw + 1
    assert oh >= 1 and ow >= 1
    xp = _pad_channels(x)
    taps = [(i - ph, j - pw, i * kw + j) for i in range(kh) for j in range(kw)]
    if mixed:
        passes = [(xp, _packed_weight(w, 0, x.dtype, xp.shape[1], wgain)[0])]
    else:
        wpk = w.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
        if xp.shape[1] != cin:
            wpk = torch.nn.functional.pad(wpk, (0, xp.shape[1] - cin))
        passes = _mfma_operands(xp, wpk, 1, 2, b_dense=True)
    multi = len(passes) > 1 or len(taps) > _lib.SBG_MAX_TAPS
    y = torch.empty([n, cout, oh, ow], dtype=torch.float32 if multi else x.dtype, device=x.device, memory_format=torch.channels_last)
    assert epi is None or (not multi)
    first = True
    for xa, wa in passes:
        wa = wa.contiguous()
        for grp in _launch_groups(taps):
            _igemm(xa, wa, y, grp, sh, oh, ow, accumulate=not first, epi=epi)
            first = False
    return y.to(x.dtype) if y.dtype != x.dtype else y


def _conv_transpose_forward(x, w, stride, padding, output_padding, wgain=1.0):
    """y[n,co,iy*s-p+kh, ix*s-p+kw] += x[n,ci,iy,ix] w[ci,co,kh,kw]  (F.conv_transpose2d), computed per output phase."""
    n, cin, ih, iw = x.shape
    cin_w, cout, kh, kw = w.shape
    mixed = is_mixed(x, w)
    assert cin == cin_w and (x.dtype == w.dtype or mixed) and (mixed or wgain == 1.0)
    (sh, sw), (ph, pw), (oph, opw) = stride, padding, output_padding
    assert sh == sw, "conv_transpose2d: only square strides are implemented"
    s = sh
    oh = (ih - 1) * s - 2 * ph + kh + oph
    ow = (iw - 1) * s - 2 * pw + kw + opw
    assert oh >= 1 and ow >= 1
    xp = _pad_channels(x)
    if mixed:
        passes = [(xp, _packed_weight(w, 1, x.dtype, xp.shape[1], wgain)[0])]
    else:
        wpk = w.permute(2, 3, 1, 0).reshape(kh * kw, cout, cin)
        if xp.shape[1] != cin:
            wpk = torch.nn.functional.pad(wpk, (0, xp.shape[1] - cin))
        passes = _mfma_operands(xp, wpk, 1, 2, b_dense=True)
    # phases: output rows oy = s*o + a use taps kh == (a + p) mod s with input row o + (a + p - kh) / s
    phases = []
    need_zero = False
    for a in range(s):
        for b in range(s):
            goh, gow = (oh - a + s - 1) // s, (ow - b + s - 1) // s
            if goh <= 0 or gow <= 0:
                continue
            taps = [((a + ph - i) // s, (b + pw - j) // s, i * kw + j)
                    for i in range(kh) if (a + ph - i) % s == 0
                    for j in range(kw) if (b + pw - j) % s == 0]
            if not taps:
                need_zero = True
                continue
            phases.append((a, b, goh, gow, taps))
    multi = len(passes) > 1 or any(len(t) > _lib.SBG_MAX_TAPS for *_, t in phases)
    y = torch.empty([n, cout, oh, ow], dtype=torch.float32 if multi else x.dtype, device=x.device, memory_format=torch.channels_last)
    if need_zero:
        y.zero_()
    first = True
    if (not multi and 2 <= len(phases) <= 4 and sum(len(t) for *_, t in phases) <= _lib.SBG_MAX_TAPS):
        xa, wa = passes[0]
        _igemm_phases(xa, wa.contiguous(), y, phases, s)
        return y
    for xa, wa in passes:
        wa = wa.contiguous()
        for a, b, goh, gow, taps in phases:
            for gi, grp in enumerate(_launch_groups(taps)):
                _igemm(xa, wa, y, grp, 1, goh, gow, y_off=(a, b), y_step=(s, s), accumulate=(not first) or gi > 0)
        first = False
    return y.to(x.dtype) if y.dtype != x.dtype else y


def _wgrad(a, b, stride, taps):
    """out[t, ca, cb] = sum_{n,py,px} a[n,ca,py,px] * b[n,cb,py*s+dy_t,px*s+dx_t]   (fp32)."""
    lib = _lib.load()
    assert a.dtype == b.dtype and a.shape[0] == b.shape[0]
    ca, cb = a.shape[1], b.shape[1]
    # The fast weight-gradient kernel walks 32-pixel row chunks of `a`.  8- and 16-pixel-wide grids (the 8x8 / 16x16 blocks: few pixels, full
    # 512 x 512 x 9 weights) would fall to the generic kernel at ~180 TFLOP/s; zero columns appended to `a` contribute nothing and put them on
    # the fast kernel (half / three quarters of its MFMAs wasted, still ~2x faster; at 4 pixels the waste wins: measured slower).  `b` needs no padding: its columns are range-checked.
    pw = a.shape[3]
    if (len(taps) == 9 and stride in (1, 2) and pw in (8, 16)
            and all(taps[t] == (taps[0][0] + t // 3, taps[0][1] + t % 3) for t in range(9))):
        a = torch.nn.functional.pad(a, (0, 32 - pw))
    ap, bp = _pad_channels(a), _pad_channels(b)
    cap, cbp = ap.shape[1], bp.shape[1]
    out = torch.empty([len(taps), cap, cbp], dtype=torch.float32, device=a.device)
    first = True
    for aa, bb in _mfma_operands(ap, bp, 0, 0):
        aa, bb = _cl(aa), _cl(bb)
        for g0 in range(0, len(taps), _lib.SBG_MAX_TAPS):
            grp = taps[g0:g0 + _lib.SBG_MAX_TAPS]
            p = _lib.WgradParams()
            p.a, p.b, p.out = aa.data_ptr(), bb.data_ptr(), out[g0:].data_ptr()
            p.dtype = _lib.dtype_code(aa.dtype)
            p.N, p.PH, p.PW, p.Ca = aa.shape[0], aa.shape[2], aa.shape[3], cap
            p.BH, p.BW, p.Cb = bb.shape[2], bb.shape[3], cbp
            p.as_n, p.as_h, p.as_w = aa.stride(0), aa.stride(2), aa.stride(3)
            p.bs_n, p.bs_h, p.bs_w = bb.stride(0), bb.stride(2), bb.stride(3)
            p.stride, p.ntaps = stride, len(grp)
            for i, (dy, dx) in enumerate(grp):
                p.tap_dy[i], p.tap_dx[i] = dy, dx
            p.accumulate = int(not first)
            nbytes = lib.sbg_conv2d_wgrad_workspace(p)
            if nbytes < 0:
                _lib.check(1, "sbg_conv2d_wgrad_workspace")
            ws = torch.empty([max(nbytes, 4) // 4], dtype=torch.float32, device=a.device) if nbytes > 0 else None
            p.workspace = ws.data_ptr() if ws is not None else None
            _lib.check(lib.sbg_conv2d_wgrad(p, _lib.stream_ptr(a.device)), "sbg_conv2d_wgrad")
        first = False
    return out[:, :ca, :cb]


# ----------------------------------------------------------------------------------------------------------------
# autograd

def _output_padding_for(transpose, stride, padding, in_hw, out_hw, k_hw):
    """output_padding of the data-gradient op (reference: conv2d_gradfix.py:95-104)."""
    if transpose:
        return (0, 0)
    return tuple(in_hw[i] - (out_hw[i] - 1) * stride[i] - (1 - 2 * padding[i]) - (k_hw[i] - 1) for i in range(2))


class _Conv(torch.autograd.Function):
    """cfg = (transpose, stride, padding, output_padding[, wgain]); weight is [Cout, Cin, kh, kw] (plain) or [Cin, Cout, kh, kw]
    (transpose), in x's dtype -- or the fp32 master weight with 16-bit x ("mixed": the operand is cast(w * wgain), the weight
    gradient comes back in fp32 in the parameter's layout, already multiplied by wgain)."""

    @staticmethod
    def forward(ctx, x, w, cfg):
        transpose, stride, padding, output_padding = cfg[:4]
        wgain = cfg[4] if len(cfg) > 4 else 1.0
        _lib.require_cuda(x, "conv2d")
        if x.dtype != w.dtype and not is_mixed(x, w):
            raise RuntimeError(f"conv2d: input ({x.dtype}) and weight ({w.dtype}) must have the same dtype (or an fp32 weight with 16-bit input)")
        if not transpose:
            y = _conv_forward(x, w, stride, padding, wgain=wgain)
        else:
            y = _conv_transpose_forward(x, w, stride, padding, output_padding, wgain=wgain)
        ctx.save_for_backward(x, w)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        transpose, stride, padding, output_padding = ctx.cfg[:4]
        wgain = ctx.cfg[4] if len(ctx.cfg) > 4 else 1.0
        dx = dw = None
        if ctx.needs_input_grad[0]:
            op = _output_padding_for(transpose, stride, padding, x.shape[2:], dy.shape[2:], w.shape[2:])
            if dy.dtype != x.dtype:
                dy = dy.to(x.dtype)
            dx = _Conv.apply(dy, w, (not transpose, stride, padding, op, wgain))
            assert dx.shape == x.shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _ConvWgrad.apply(dy, x, ctx.cfg, tuple(w.shape), wmeta_of(x, w))
        return dx, dw, None


def wmeta_of(x, w):
    """strides of the master weight when the launch is mixed (the weight gradient is then produced in that layout), else None"""
    return tuple(w.stride()) if is_mixed(x, w) else None


class _ConvWgrad(torch.autograd.Function):
    """(dy, x, cfg, wshape, wmeta) -> dw.  wmeta None: dw in x's dtype (the reference's cuDNN weight-gradient op, conv2d_gradfix.py:140-147);
    wmeta = strides of the fp32 master weight: dw fp32 in that layout, multiplied by cfg's wgain."""

    @staticmethod
    def forward(ctx, dy, x, cfg, wshape, wmeta=None):
        transpose, stride, padding, _ = cfg[:4]
        wgain = cfg[4] if len(cfg) > 4 else 1.0
        kh, kw = wshape[2], wshape[3]
        taps = [(i - padding[0], j - padding[1]) for i in range(kh) for j in range(kw)]
        assert stride[0] == stride[1]
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if not transpose:
            out = _wgrad(dy, x, stride[0], taps)     # [t, Cout, Cin]
        else:
            out = _wgrad(x, dy, stride[0], taps)     # [t, Cin, Cout]
        if wmeta is not None:
            dw = _unpack_wgrad(out, wshape[0], wshape[1], wshape, wmeta, 0, wgain)
        else:
            dw = out.reshape(kh, kw, wshape[0], wshape[1]).permute(2, 3, 0, 1).to(x.dtype)
        ctx.save_for_backward(dy, x)
        ctx.cfg, ctx.wshape = cfg, wshape
        return dw

    @staticmethod
    def backward(ctx, ddw):
        dy, x = ctx.saved_tensors
        transpose, stride, padding, output_padding = ctx.cfg[:4]
        d_dy = d_x = None
        if ctx.needs_input_grad[0]:
            d_dy = _Conv.apply(x, ddw, ctx.cfg)
            assert d_dy.shape == dy.shape
        if ctx.needs_input_grad[1]:
            op = _output_padding_for(transpose, stride, padding, x.shape[2:], dy.shape[2:], ctx.wshape[2:])
            d_x = _Conv.apply(dy, ddw, (not transpose, stride, padding, op) + tuple(ctx.cfg[4:]))
            assert d_x.shape == x.shape
        return d_dy, d_x, None, None, None


def _grouped(fn, input, weight, groups, transpose):
    """groups > 1: one launch set per group (only the eval-time fused modulated conv and depthwise filters use it)."""
    cin_g = input.shape[1] // groups
    outs = []
    wg = weight.shape[0] // groups
    for g in range(groups):
        outs.append(fn(input[:, g * cin_g:(g + 1) * cin_g], weight[g * wg:(g + 1) * wg]))
    return torch.cat(outs, dim=1)


def _add_bias(y, bias):
    if bias is None:
        return y
    from . import bias_act
    return bias_act.bias_act(y, bias.to(y.dtype))


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, wgain=1.0):
    """Same contract as torch.nn.functional.conv2d / the reference's conv2d (conv2d_gradfix.py:35), HIP kernels inside.
    Extension: `weight` may be the fp32 master parameter with a 16-bit input; the operand is then cast(weight * wgain)."""
    _lib.require_cuda(input, "conv2d")
    assert _pair(dilation) == (1, 1), "conv2d: dilation is not implemented"
    cfg = (False, _pair(stride), _pair(padding), (0, 0), float(wgain))
    assert groups == 1 or not is_mixed(input, weight)
    assert all(p >= 0 for p in cfg[2]) and all(s >= 1 for s in cfg[1])
    if groups == 1:
        y = _Conv.apply(input, weight, cfg)
    else:
        y = _grouped(lambda a, b: _Conv.apply(a, b, cfg), input, weight, groups, False)
    return _add_bias(y, bias)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, wgain=1.0):
    """Same contract as torch.nn.functional.conv_transpose2d / the reference's conv_transpose2d (conv2d_gradfix.py:40)."""
    _lib.require_cuda(input, "conv_transpose2d")
    assert _pair(dilation) == (1, 1), "conv_transpose2d: dilation is not implemented"
    cfg = (True, _pair(stride), _pair(padding), _pair(output_padding), float(wgain))
    assert groups == 1 or not is_mixed(input, weight)
    assert all(0 <= cfg[3][i] < max(cfg[1][i], 1) or cfg[3][i] == 0 for i in range(2))
    if groups == 1:
        y = _Conv.apply(input, weight, cfg)
    else:
        y = _grouped(lambda a, b: _Conv.apply(a, b, cfg), input, weight, groups, True)
    return _add_bias(y, bias)
