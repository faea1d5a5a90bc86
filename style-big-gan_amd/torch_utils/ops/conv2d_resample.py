"""2-D convolution with optional FIR up / down-sampling.

Host-side mirror of ``stylegan2ada/torch_utils/ops/conv2d_resample.py`` (``conv2d_resample`` :59-154): the same
decision tree over (kernel size, up, down) and the same padding arithmetic, expressed over this package's HIP-backed
``conv2d_gradfix`` and ``upfirdn2d``.  `flip_weight=True` means correlation (what conv2d computes), False means true
convolution (the weight is flipped first), exactly as in the reference (:29-54).
"""
import torch

from . import bias_act
from . import conv2d_gradfix
from . import conv_bias_act
from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True, tail=None, wgain=1.0):
    """plain / transposed convolution; flips the taps when a true convolution is requested.
    `tail` = dict(b, act, alpha, gain, clamp): the bias_act that follows, fused into the kernel epilogue when possible."""
    if not flip_weight:
        w = w.flip([2, 3])
    if tail is not None and not transpose and conv_bias_act.fusable(x, w, tail["act"], groups):
        return conv_bias_act.conv2d_bias_act(x, w, tail["b"], stride=stride, padding=padding, act=tail["act"], alpha=tail["alpha"],
                                             gain=tail["gain"], clamp=tail["clamp"], wgain=wgain)
    fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    y = fn(x, w, stride=stride, padding=padding, groups=groups, wgain=wgain)
    return _tail(y, tail)


def _tail(y, tail):
    if tail is None:
        return y
    b = tail["b"]
    if b is not None and b.dtype != y.dtype:        # layers hand over their fp32 bias parameter (the fused conv epilogue reads fp32)
        b = b.to(y.dtype)
    return bias_act.bias_act(y, b, act=tail["act"], alpha=tail["alpha"], gain=tail["gain"], clamp=tail["clamp"])


def lowpass_padding(f, down, padding):
    """[px0, px1, py0, py1] of the low-pass that conv2d_resample(..., up=1, down=down, padding) applies in front of its strided convolution"""
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    return [px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2]


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False, bias_act_tail=None, fir_tail=None,
                    wgain=1.0, prefiltered=False):
    """x: [N, Cin, H, W]; w: [Cout, Cin // groups, kh, kw] (same dtype); f: filter from upfirdn2d.setup_filter().
    `padding` is relative to the upsampled image.  Returns [N, Cout, H * up // down (+ padding), ...].
    `bias_act_tail` (extension): dict(b, act, alpha, gain, clamp) -- apply that bias_act to the result, fused into the
    convolution kernel when the convolution is the last stage.
    `fir_tail` (extension, up-sampling branch): dict(dcoefs, noise, b, act, alpha, gain, clamp[, post]) -- the demodulation + noise + bias_act
    of a synthesis layer, fused into the low-pass kernel that ends the branch (first order only; the caller checks
    upfirdn2d.fir_tail_supported on the result of fir_tail_probe).
    `wgain` (extension): with 16-bit x, `w` may be the layer's fp32 parameter; the convolution operand is then cast(w * wgain), prepared
    (and cached) by one kernel, and the weight gradient arrives in fp32 in the parameter's layout."""
    tail = bias_act_tail
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and (w.dtype == x.dtype or conv2d_gradfix.is_mixed(x, w))
    assert wgain == 1.0 or conv2d_gradfix.is_mixed(x, w)
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    assert not prefiltered or (up == 1 and down > 1 and w.shape[2] > 1)
    cout, cin_g, kh, kw = [int(s) for s in w.shape]
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # The FIR kernels eat part of the requested padding.
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2
    pad4 = [px0, px1, py0, py1]
    pointwise = (kh == 1 and kw == 1)

    if pointwise and down > 1 and up == 1:          # resample on the cheap side of a 1x1 conv: first shrink ...
        x = upfirdn2d.upfirdn2d(x, f, down=down, padding=pad4, flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight, tail=tail, wgain=wgain)

    if pointwise and up > 1 and down == 1:          # ... or convolve first, then grow
        x = _conv(x, w, groups=groups, flip_weight=flip_weight, wgain=wgain)
        return _tail(upfirdn2d.upfirdn2d(x, f, up=up, padding=pad4, gain=up ** 2, flip_filter=flip_filter), tail)

    if down > 1 and up == 1:                        # low-pass at full resolution, then a strided convolution
        if not prefiltered:                         # (`prefiltered`, extension: x already is upfirdn2d(., f, lowpass_padding(f, down, padding)) -- the
            x = upfirdn2d.upfirdn2d(x, f, padding=pad4, flip_filter=flip_filter)      # producer applied it, conv_bias_act.conv2d_bias_act_fir)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight, tail=tail, wgain=wgain)

    if up > 1:                                      # transposed strided convolution, then low-pass (and optional decimation)
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2).reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up
        py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight), wgain=wgain)
        fpad = [px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt]
        if fir_tail is not None and down == 1 and upfirdn2d.fir_tail_supported(x, f, fpad, flip_filter):
            return upfirdn2d.fir_bias_act(x, f, fpad, up ** 2, fir_tail["dcoefs"], fir_tail["noise"], fir_tail["b"], act=fir_tail["act"],
                                          alpha=fir_tail["alpha"], act_gain=fir_tail["gain"], clamp=fir_tail["clamp"], flip_filter=flip_filter,
                                          post_scale=fir_tail.get("post"))
        x = upfirdn2d.upfirdn2d(x, f, padding=fpad, gain=up ** 2, flip_filter=flip_filter)
        if fir_tail is not None:       # unfused composition of the same tail
            from . import modulate
            x = modulate.scale_nc(x, fir_tail["dcoefs"], fir_tail["noise"]) if fir_tail["dcoefs"] is not None else (x if fir_tail["noise"] is None else x + fir_tail["noise"].to(x.dtype))
            x = bias_act.bias_act(x, fir_tail["b"].to(x.dtype) if fir_tail["b"] is not None else None, act=fir_tail["act"], alpha=fir_tail["alpha"],
                                  gain=fir_tail["gain"], clamp=(fir_tail["clamp"] if fir_tail["clamp"] >= 0 else None))
            return x if fir_tail.get("post") is None else modulate.scale_nc(x, fir_tail["post"])
        if down > 1:
            x = upfirdn2d.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return _tail(x, tail)

    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:     # plain convolution with symmetric padding
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight, tail=tail, wgain=wgain)

    # generic composition: pad/crop with an identity FIR, convolve, decimate
    x = upfirdn2d.upfirdn2d(x, (f if up > 1 else None), up=up, padding=pad4, gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight, wgain=wgain)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return _tail(x, tail)
