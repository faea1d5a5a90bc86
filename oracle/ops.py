"""CPU restatement of the reference's custom ops (test infrastructure, see oracle/__init__.py).

Each function cites the reference lines it follows (paths relative to the reference checkout,
stylegan2ada/torch_utils/ops/).  Everything is plain differentiable PyTorch, so gradients of any order come from autograd.
"""
import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------------------------
# bias_act  (bias_act.py:23-33 activation table, :94-123 _bias_act_ref)

_SQRT2 = float(np.sqrt(2))
ACTIVATIONS = {
    #            fn                                   def_alpha def_gain
    "linear":   (lambda x, alpha: x,                          0.0, 1.0),
    "relu":     (lambda x, alpha: F.relu(x),                  0.0, _SQRT2),
    "lrelu":    (lambda x, alpha: F.leaky_relu(x, alpha),     0.2, _SQRT2),
    "tanh":     (lambda x, alpha: torch.tanh(x),              0.0, 1.0),
    "sigmoid":  (lambda x, alpha: torch.sigmoid(x),           0.0, 1.0),
    "elu":      (lambda x, alpha: F.elu(x),                   0.0, 1.0),
    "selu":     (lambda x, alpha: F.selu(x),                  0.0, 1.0),
    "softplus": (lambda x, alpha: F.softplus(x),              0.0, 1.0),
    "swish":    (lambda x, alpha: torch.sigmoid(x) * x,       0.0, _SQRT2),
}


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None):
    fn, def_alpha, def_gain = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


# ----------------------------------------------------------------------------------------------------------------
# upfirdn2d  (upfirdn2d.py:72-116 setup_filter, :169-208 _upfirdn2d_ref, :272-382 wrappers)

def setup_filter(f, normalize=True, flip_filter=False, gain=1, separable=None):
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = f[:, None] * f[None, :]
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding] * 2
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return padding


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _fsize(f):
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    n, c, h, w = x.shape
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    # zero-insertion: x sits on the up-grid, zeros after each sample (:184-186)
    u = x.new_zeros([n, c, h * upy, w * upx])
    u[:, :, ::upy, ::upx] = x
    # pad (positive) / crop (negative) (:189-190)
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0): u.shape[2] - max(-py1, 0), max(-px0, 0): u.shape[3] - max(-px1, 0)]
    # filter: gain folded in, true convolution unless flip_filter (:193-196)
    f = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    if f.ndim == 2:
        u = F.conv2d(u, f[None, None].repeat(c, 1, 1, 1), groups=c)
    else:   # separable: rows then columns (:203-204)
        u = F.conv2d(u, f[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        u = F.conv2d(u, f[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return u[:, :, ::downy, ::downx]    # decimate (:207)


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    return upfirdn2d(x, f, padding=[px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2],
                     flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    upx, upy = _pair(up)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    dx, dy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = _fsize(f)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


# ----------------------------------------------------------------------------------------------------------------
# conv2d_resample  (conv2d_resample.py:29-54 _conv2d_wrapper, :59-154 conv2d_resample)

def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    if not flip_weight:
        w = w.flip([2, 3])
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    cout, cin_g, kh, kw = w.shape
    fw, fh = _fsize(f)
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:      # (:95-99)
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2; py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:    # (:100-104)
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2; py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2
    if kh == 1 and kw == 1 and down > 1 and up == 1:    # (:107-110)
        x = upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kh == 1 and kw == 1 and up > 1 and down == 1:    # (:113-116)
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                            # (:119-122)
        x = upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                          # (:125-142)
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2).reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:     # (:145-147)
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d(x, (f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)   # (:150-154)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


# ----------------------------------------------------------------------------------------------------------------
# fma (fma.py:15) and modulated_conv2d (train_parts/generators.py:43-100)

def fma(a, b, c):
    return a * b + c


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True):
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    if x.dtype == torch.float16 and demodulate:     # (:63-65)
        weight = weight * (1 / np.sqrt(cin * kh * kw) / weight.norm(float("inf"), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float("inf"), dim=1, keepdim=True)
    w = dcoefs = None
    if demodulate or fused_modconv:                 # (:68-76)
        w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if demodulate and fused_modconv:
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    if not fused_modconv:                           # (:79-88)
        x = x * styles.to(x.dtype).reshape(n, -1, 1, 1)
        x = conv2d_resample(x, weight.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma(x, dcoefs.to(x.dtype).reshape(n, -1, 1, 1), noise.to(x.dtype))
        elif demodulate:
            x = x * dcoefs.to(x.dtype).reshape(n, -1, 1, 1)
        elif noise is not None:
            x = x + noise.to(x.dtype)
        return x
    x = x.reshape(1, -1, *x.shape[2:])              # (:90-100)
    w = w.reshape(-1, cin, kh, kw)
    x = conv2d_resample(x, w.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x
