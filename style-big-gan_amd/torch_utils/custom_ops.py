"""Call-signature shim of the reference's plugin loader (``stylegan2ada/torch_utils/custom_ops.py:46``).

The reference JIT-compiles two pybind11 extensions and hands back modules with ONE function each: ``get_plugin('upfirdn2d_plugin',
sources=[...]).upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)`` (``upfirdn2d.cpp:16,98-101``) and
``get_plugin('bias_act_plugin', sources=[...]).bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)``
(``bias_act.cpp:32,94-97``).  Here nothing is compiled at run time: ``libsbg_hip.so`` is built ahead of time (``make -C csrc``) and
``get_plugin`` returns an object with the same function over the C ABI (``sbg_upfirdn2d`` / ``sbg_bias_act``); ``sources`` and the build
keywords are accepted and ignored, unknown plugin names raise like a failed build.  Conventions kept from the pybind entry points: an empty
tensor (``numel() == 0``) means "absent" (``bias_act.py:39,150-157``), the output is freshly allocated in the input's memory format, the
launch goes to the current stream, errors are ``RuntimeError``.  The op modules of this package do not go through this shim (they call
the library directly, with fused tails the pybind signature cannot express); it exists for code written against the reference's loader.
"""
import types

import torch

from .. import _lib

verbosity = 'brief'         # 'none' | 'brief' | 'full' (reference :22): kept so that `custom_ops.verbosity = 'none'` on rank != 0 works
_cached_plugins = dict()


def _absent(t):
    return t is None or t.numel() == 0


def _upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    from .ops import upfirdn2d as op
    _lib.require_cuda(x, "upfirdn2d_plugin.upfirdn2d")
    if f.dtype != torch.float32 or f.ndim != 2:
        raise RuntimeError("upfirdn2d: f must be a float32 rank-2 tensor")       # upfirdn2d.cpp:19-21
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError("upfirdn2d: x is too large")                           # upfirdn2d.cpp:22-23
    return op._launch(x, f.to(x.device), int(upx), int(upy), int(downx), int(downy), int(padx0), int(padx1), int(pady0), int(pady1),
                      bool(flip), float(gain))


def _bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    _lib.require_cuda(x, "bias_act_plugin.bias_act")
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError("bias_act: x is too large")                            # bias_act.cpp:40
    lib = _lib.load()
    fmt = torch.channels_last if (x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1) else torch.contiguous_format
    x = x.contiguous(memory_format=fmt)
    aux = []
    for t, name in ((xref, "xref"), (yref, "yref"), (dy, "dy")):
        if _absent(t):
            aux.append(None)
            continue
        if t.shape != x.shape or t.dtype != x.dtype or t.device != x.device:
            raise RuntimeError(f"bias_act: {name} must have the same shape, dtype and device as x")      # bias_act.cpp:42-51
        aux.append(t.contiguous(memory_format=fmt))
    size_b, step_b = 0, 1
    if not _absent(b):
        if b.ndim != 1 or b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError("bias_act: b must be a rank-1 tensor with the same dtype and device as x")
        if not (0 <= dim < x.ndim) or b.shape[0] != x.shape[dim]:
            raise RuntimeError("bias_act: b has wrong number of elements")
        b = b.contiguous()
        size_b, step_b = b.shape[0], max(x.stride(dim), 1)
    else:
        b = None
    y = torch.empty_like(x, memory_format=torch.preserve_format)
    if x.numel() == 0:
        return y
    status = lib.sbg_bias_act(_lib.ptr(x), _lib.ptr(b), _lib.ptr(aux[0]), _lib.ptr(aux[1]), _lib.ptr(aux[2]), _lib.ptr(y),
                              _lib.dtype_code(x.dtype), int(grad), int(act), float(alpha), float(gain), float(clamp),
                              x.numel(), size_b, step_b, _lib.stream_ptr(x.device))
    _lib.check(status, "sbg_bias_act")
    return y


_PLUGINS = {'upfirdn2d_plugin': dict(upfirdn2d=_upfirdn2d), 'bias_act_plugin': dict(bias_act=_bias_act)}


def get_plugin(module_name, sources=None, **build_kwargs):
    """-> module-like object exporting the plugin's function.  Raises RuntimeError when libsbg_hip.so is missing (no silent fallback)."""
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if module_name not in _PLUGINS:
        raise RuntimeError(f'custom_ops.get_plugin: no ahead-of-time kernel library serves "{module_name}" (known: {sorted(_PLUGINS)})')
    _lib.load()
    mod = types.SimpleNamespace(__name__=module_name, **_PLUGINS[module_name])
    _cached_plugins[module_name] = mod
    return mod
