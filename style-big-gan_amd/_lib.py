"""ctypes binding of libsbg_hip.so (the C ABI declared in include/sbg_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBG_HIP_LIBRARY") or os.path.join(_HERE, "libsbg_hip.so")      # SBG_HIP_LIBRARY: another build of the same C ABI (diagnosis builds)

SBG_F32, SBG_F16, SBG_BF16 = 0, 1, 2
SBG_MAX_TAPS = 16

_DTYPES = {torch.float32: SBG_F32, torch.float16: SBG_F16, torch.bfloat16: SBG_BF16}


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise RuntimeError(f"style_big_gan_amd: unsupported dtype {dtype} (float32 / float16 / bfloat16 only)")


class UpfirdnParams(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("f", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("dtype", ctypes.c_int),
        ("upx", ctypes.c_int), ("upy", ctypes.c_int), ("downx", ctypes.c_int), ("downy", ctypes.c_int),
        ("padx0", ctypes.c_int), ("pady0", ctypes.c_int),
        ("flip", ctypes.c_int), ("gain", ctypes.c_float),
        ("inSize", ctypes.c_int * 4), ("inStride", ctypes.c_int64 * 4),
        ("filterSize", ctypes.c_int * 2), ("filterStride", ctypes.c_int * 2),
        ("outSize", ctypes.c_int * 4), ("outStride", ctypes.c_int64 * 4),
        ("oscale", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("noise_stride_n", ctypes.c_int64), ("bias", ctypes.c_void_p),
        ("act", ctypes.c_int), ("alpha", ctypes.c_float), ("act_gain", ctypes.c_float), ("clamp", ctypes.c_float),
        ("filter_exact16", ctypes.c_int),
        ("dact_y", ctypes.c_void_p), ("dact_partial", ctypes.c_void_p), ("dact_act", ctypes.c_int),
        ("dact_alpha", ctypes.c_float), ("dact_gain", ctypes.c_float), ("dact_clamp", ctypes.c_float),
        ("post_scale", ctypes.c_void_p),
    ]


class ConvParams(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p), ("oscale", ctypes.c_void_p),
        ("xdtype", ctypes.c_int), ("ydtype", ctypes.c_int),
        ("N", ctypes.c_int), ("IH", ctypes.c_int), ("IW", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int),
        ("OH", ctypes.c_int), ("OW", ctypes.c_int),
        ("xs_n", ctypes.c_int64), ("xs_h", ctypes.c_int64), ("xs_w", ctypes.c_int64),
        ("ys_n", ctypes.c_int64), ("ys_h", ctypes.c_int64), ("ys_w", ctypes.c_int64),
        ("ws_slab", ctypes.c_int64), ("ws_co", ctypes.c_int64),
        ("stride", ctypes.c_int), ("ntaps", ctypes.c_int),
        ("tap_dy", ctypes.c_int * SBG_MAX_TAPS), ("tap_dx", ctypes.c_int * SBG_MAX_TAPS), ("tap_slab", ctypes.c_int * SBG_MAX_TAPS),
        ("accumulate", ctypes.c_int),
        ("bias", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("noise_stride_n", ctypes.c_int64),
        ("act", ctypes.c_int), ("alpha", ctypes.c_float), ("gain", ctypes.c_float), ("clamp", ctypes.c_float),
        ("workspace", ctypes.c_void_p), ("ksplit", ctypes.c_int),
        ("nphase", ctypes.c_int), ("ph_ntaps", ctypes.c_int * 4), ("ph_oh", ctypes.c_int * 4), ("ph_ow", ctypes.c_int * 4), ("ph_yoff", ctypes.c_int64 * 4),
    ]


class WgradParams(ctypes.Structure):
    _fields_ = [
        ("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("out", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("dtype", ctypes.c_int),
        ("N", ctypes.c_int), ("PH", ctypes.c_int), ("PW", ctypes.c_int), ("Ca", ctypes.c_int),
        ("BH", ctypes.c_int), ("BW", ctypes.c_int), ("Cb", ctypes.c_int),
        ("as_n", ctypes.c_int64), ("as_h", ctypes.c_int64), ("as_w", ctypes.c_int64),
        ("bs_n", ctypes.c_int64), ("bs_h", ctypes.c_int64), ("bs_w", ctypes.c_int64),
        ("stride", ctypes.c_int), ("ntaps", ctypes.c_int),
        ("tap_dy", ctypes.c_int * SBG_MAX_TAPS), ("tap_dx", ctypes.c_int * SBG_MAX_TAPS),
        ("accumulate", ctypes.c_int),
    ]


class GridSampleParams(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("grid", ctypes.c_void_p), ("theta", ctypes.c_void_p), ("dy", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("dgrid", ctypes.c_void_p),
        ("N", ctypes.c_int), ("C", ctypes.c_int), ("IH", ctypes.c_int), ("IW", ctypes.c_int), ("OH", ctypes.c_int), ("OW", ctypes.c_int),
        ("xs_n", ctypes.c_int64), ("xs_c", ctypes.c_int64), ("xs_h", ctypes.c_int64), ("xs_w", ctypes.c_int64),
        ("ys_n", ctypes.c_int64), ("ys_c", ctypes.c_int64), ("ys_h", ctypes.c_int64), ("ys_w", ctypes.c_int64),
        ("theta_host", ctypes.c_void_p),
    ]


class GgProblem(ctypes.Structure):
    """sbg_gg_problem (include/sbg_hip.h): C = sum_t alpha_t A_t B_t (+ bias * bias_scale), strides in elements"""
    _fields_ = [("a0", ctypes.c_void_p), ("b0", ctypes.c_void_p), ("a1", ctypes.c_void_p), ("b1", ctypes.c_void_p),
                ("c", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("rowsum", ctypes.c_void_p),
                ("a0_rs", ctypes.c_int64), ("a0_cs", ctypes.c_int64), ("b0_rs", ctypes.c_int64), ("b0_cs", ctypes.c_int64),
                ("a1_rs", ctypes.c_int64), ("a1_cs", ctypes.c_int64), ("b1_rs", ctypes.c_int64), ("b1_cs", ctypes.c_int64),
                ("c_rs", ctypes.c_int64), ("c_cs", ctypes.c_int64),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K0", ctypes.c_int), ("K1", ctypes.c_int), ("nterms", ctypes.c_int),
                ("alpha0", ctypes.c_float), ("alpha1", ctypes.c_float), ("bias_scale", ctypes.c_float), ("rowsum_scale", ctypes.c_float)]


class ProfRecord(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("dims", ctypes.c_int * 7), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("ms", ctypes.c_float), ("pad", ctypes.c_int)]


KERNEL_KINDS = {1: "bias_act", 2: "upfirdn2d", 3: "conv_igemm", 4: "conv_wgrad", 5: "wgrad_reduce", 6: "scale_nc", 7: "dot_hw", 9: "sn_power", 10: "attention",
                11: "grid_sample", 12: "filter1d", 13: "color", 14: "weight_prep", 15: "torgb", 16: "fromrgb", 17: "grouped_gemm"}

_lib = None
_lock = threading.Lock()

# every symbol include/sbg_hip.h declares: (name, restype, argtypes)
_c = ctypes
SYMBOLS = [
    ("sbg_version", _c.c_int, []),
    ("sbg_last_error", _c.c_char_p, []),
    ("sbg_bias_act", _c.c_int, [_c.c_void_p] * 6 + [_c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float,
                                                   _c.c_int64, _c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_upfirdn2d_tail_supported", _c.c_int, [_c.POINTER(UpfirdnParams)]),
    ("sbg_upfirdn2d", _c.c_int, [_c.POINTER(UpfirdnParams), _c.c_void_p]),
    ("sbg_upfirdn2d_dact_rows", _c.c_int64, [_c.POINTER(UpfirdnParams)]),
    ("sbg_upfirdn2d_separable_supported", _c.c_int, [_c.c_int] * 3),
    ("sbg_upfirdn2d_separable", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int] * 11 + [_c.c_float, _c.c_void_p]),
    ("sbg_conv2d_igemm_workspace", _c.c_int64, [_c.POINTER(ConvParams)]),
    ("sbg_conv2d_igemm", _c.c_int, [_c.POINTER(ConvParams), _c.c_void_p]),
    ("sbg_conv2d_wgrad_workspace", _c.c_int64, [_c.POINTER(WgradParams)]),
    ("sbg_conv2d_wgrad", _c.c_int, [_c.POINTER(WgradParams), _c.c_void_p]),
    ("sbg_scale_nc", _c.c_int, [_c.c_void_p] * 4 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_void_p]),
    ("sbg_scale_shift_nc", _c.c_int, [_c.c_void_p] * 4 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_sn_workspace", _c.c_int64, [_c.c_int, _c.c_int]),
    ("sbg_sn_power_iteration", _c.c_int, [_c.c_void_p] * 6 + [_c.c_int, _c.c_int, _c.c_float, _c.c_void_p]),
    ("sbg_mbstd_workspace", _c.c_int64, [_c.c_int] * 5),
    ("sbg_mbstd_fwd", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int] * 5 + [_c.c_void_p]),
    ("sbg_mbstd_bwd", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int] * 5 + [_c.c_void_p]),
    ("sbg_split_bf16_cat", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.POINTER(_c.c_int), _c.c_void_p]),
    ("sbg_split_bf16_cat_nd", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _c.c_int, _c.c_void_p, _c.c_int, _c.POINTER(_c.c_int), _c.c_void_p]),
    ("sbg_attention_supported", _c.c_int, [_c.c_int] * 4),
    ("sbg_attention_fwd", _c.c_int, [_c.c_void_p] * 4 + [_c.c_int] * 5 + [_c.c_void_p]),
    ("sbg_attention_bwd_supported", _c.c_int, [_c.c_int] * 4),
    ("sbg_attention_bwd_workspace", _c.c_int64, [_c.c_int, _c.c_int]),
    ("sbg_attention_bwd", _c.c_int, [_c.c_void_p] * 8 + [_c.c_int] * 5 + [_c.c_void_p]),
    ("sbg_dot_hw_splits", _c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int64]),
    ("sbg_dot_hw", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_moments_hw", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_grouped_gemm", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p]),
    ("sbg_dot_hw_scale_supported", _c.c_int, [_c.c_int]),
    ("sbg_dot_hw_scale", _c.c_int, [_c.c_void_p] * 5 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_modconv_bwd_supported", _c.c_int, [_c.c_int]),
    ("sbg_modconv_bwd", _c.c_int, [_c.c_void_p] * 8 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p]),
    ("sbg_modconv_bwd_prescaled", _c.c_int, [_c.c_void_p] * 10 + [_c.c_int, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p]),
    ("sbg_pack_weight", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int] + [_c.c_int64] * 4
     + [_c.c_int, _c.c_float, _c.c_void_p, _c.c_void_p]),
    ("sbg_unpack_wgrad", _c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 4
     + [_c.c_int64] * 4 + [_c.c_float, _c.c_void_p]),
    ("sbg_demod_coefs", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int] * 3 + [_c.c_float, _c.c_void_p]),
    ("sbg_demod_coefs_bwd", _c.c_int, [_c.c_void_p] * 6 + [_c.c_int] * 3 + [_c.c_void_p]),
    ("sbg_torgb_supported", _c.c_int, [_c.c_int, _c.c_int]),
    ("sbg_torgb_bwd_blocks", _c.c_int, [_c.c_int, _c.c_int, _c.c_int64]),
    ("sbg_torgb_fwd", _c.c_int, [_c.c_void_p] * 4 + [_c.c_int] * 4 + [_c.c_int64, _c.c_float, _c.c_void_p]),
    ("sbg_torgb_bwd", _c.c_int, [_c.c_void_p] * 6 + [_c.c_int] * 4 + [_c.c_int64, _c.c_float, _c.c_void_p]),
    ("sbg_fromrgb_supported", _c.c_int, [_c.c_int] * 3),
    ("sbg_fromrgb_bwd_blocks", _c.c_int, [_c.c_int, _c.c_int64]),
    ("sbg_fromrgb_fwd", _c.c_int, [_c.c_void_p] * 4 + [_c.c_int] * 4 + [_c.c_int64, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p]),
    ("sbg_fromrgb_bwd", _c.c_int, [_c.c_void_p] * 6 + [_c.c_int] * 4 + [_c.c_int64, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_void_p]),
    ("sbg_grid_sample2d", _c.c_int, [_c.POINTER(GridSampleParams), _c.c_void_p]),
    ("sbg_grid_sample2d_bwd", _c.c_int, [_c.POINTER(GridSampleParams), _c.c_void_p]),
    ("sbg_grid_sample2d_bwd_overwrites", _c.c_int, [_c.POINTER(GridSampleParams)]),
    ("sbg_color_transform", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int, _c.c_int64, _c.c_void_p]),
    ("sbg_filter1d_batch", _c.c_int, [_c.c_void_p] * 3 + [_c.c_int] * 8 + [_c.c_void_p]),
    ("sbg_experiment_set", _c.c_int, [_c.c_int]),
    ("sbg_prof_enable", _c.c_int, [_c.c_int]),
    ("sbg_prof_fetch", _c.c_int, [_c.POINTER(ProfRecord), _c.c_int]),
]


def load():
    """Load libsbg_hip.so once; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"style_big_gan_amd: HIP extension {LIB_PATH} not found. Build it with "
                f"`make -C {os.path.join(_HERE, 'csrc')}` or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().sbg_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_cuda(t, what):
    if t.device.type != "cuda":
        raise RuntimeError(
            f"{what}: tensor is on '{t.device}'. style_big_gan_amd ops run only as HIP kernels on a ROCm device "
            "(device type 'cuda'); there is no CPU path in the product (the CPU restatement lives in oracle/ and is test-only).")


def prof_enable(on):
    """switch the library's per-launch hipEvent timing on / off (measurement only)"""
    return load().sbg_prof_enable(int(bool(on)))


def prof_fetch():
    """-> list of dicts (kind, dims, flops, bytes, ms) for every launch logged since the last fetch"""
    lib = load()
    n = lib.sbg_prof_fetch(None, 0)
    if n <= 0:
        return []
    buf = (ProfRecord * n)()
    n = lib.sbg_prof_fetch(buf, n)
    return [dict(kind=KERNEL_KINDS.get(r.kind, str(r.kind)), dims=tuple(r.dims), flops=r.flops, bytes=r.bytes, ms=r.ms) for r in buf[:n]]
