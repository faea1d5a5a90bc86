// weight_prep.hip -- fp32 master weights <-> the packed operands of the matrix-core convolutions, one launch each way.
//
// The reference's layers prepare a convolution weight with a chain of framework ops per call -- `w = self.weight * weight_gain`,
// `w.to(x.dtype)` (train_parts/generators.py:176-179, discriminators.py:115-118), the layout change cuDNN does internally -- and
// autograd replays the chain backwards for the gradient.  At this model's size those are ~10 launch-bound micro-kernels per
// convolution and pass.  Here:
//   sbg_pack_weight   : out[t][a][b] = cast(w[a, b, t] * gain), b zero-padded to Bp   (the [tap][rows][cols] operand of conv_k64.hip;
//                       rows/cols = (Cout, Cin) for a forward convolution, swapped for its data gradient -- the caller passes strides)
//                       optionally w2[a][b] = sum_t (w * gain)^2, the demodulation's sum over taps (generators.py:71-76).
//   sbg_unpack_wgrad  : dw[a, b, t] = gain * dwp[t][a][b]  (+ 2 gain^2 w[a, b, t] * dw2[a][b]),  the weight-gradient kernel's
//                       fp32 [tap][rows][cols] result back in the parameter's own layout (any strides), no 16-bit round trip.
// Both are tiny streaming kernels (<= 2.4 M elements); one lane per (a, b) pair walks the taps.
#include "sbg_common.h"

namespace {

struct PackArgs {
    const float* w; void* out; float* w2;
    int A, B, KH, KW, Bp, dtype;
    int64_t sA, sB, sKH, sKW;
    float gain;
};

template <class T>
__global__ void __launch_bounds__(256) pack_weight_kernel(PackArgs p)
{
    const int64_t total = (int64_t)p.A * p.Bp;
    const int taps = p.KH * p.KW;
    T* out = (T*)p.out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % p.Bp), a = (int)(i / p.Bp);
        const bool real = b < p.B;
        const float* src = p.w + (int64_t)a * p.sA + (int64_t)b * p.sB;
        float sq = 0.0f;
        for (int kh = 0; kh < p.KH; kh++)
            for (int kw = 0; kw < p.KW; kw++) {
                const float v = real ? src[kh * p.sKH + kw * p.sKW] * p.gain : 0.0f;
                sq += v * v;
                Elem<T>::st(out + ((int64_t)(kh * p.KW + kw) * p.A + a) * p.Bp + b, v);
            }
        if (p.w2 && real) p.w2[(int64_t)a * p.B + b] = sq;
    }
}

struct UnpackArgs {
    const float* dwp; float* dw; const float* w; const float* dw2;
    int A, B, KH, KW;
    int64_t pT, pA;                 // strides of dwp (tap, row); columns are dense
    int64_t sA, sB, sKH, sKW;       // strides of dw (and w)
    float gain;
};

__global__ void __launch_bounds__(256) unpack_wgrad_kernel(UnpackArgs p)
{
    const int64_t total = (int64_t)p.A * p.B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % p.B), a = (int)(i / p.B);
        const int64_t off = (int64_t)a * p.sA + (int64_t)b * p.sB;
        const float extra = p.dw2 ? 2.0f * p.gain * p.gain * p.dw2[(int64_t)a * p.B + b] : 0.0f;
        for (int kh = 0; kh < p.KH; kh++)
            for (int kw = 0; kw < p.KW; kw++) {
                const int64_t o = off + kh * p.sKH + kw * p.sKW;
                float v = p.dwp ? p.gain * p.dwp[(int64_t)(kh * p.KW + kw) * p.pT + (int64_t)a * p.pA + b] : 0.0f;
                if (p.dw2) v += extra * p.w[o];
                p.dw[o] = v;
            }
    }
}

}  // namespace

extern "C" int sbg_pack_weight(const float* w, void* out, int out_dtype, int A, int B, int KH, int KW,
                               int64_t sA, int64_t sB, int64_t sKH, int64_t sKW, int Bp, float gain, float* w2, sbg_stream_t stream_)
{
    SBG_CHECK(w && out, "pack_weight: null pointer");
    SBG_CHECK(A >= 1 && B >= 1 && KH >= 1 && KW >= 1 && Bp >= B, "pack_weight: bad sizes");
    SBG_CHECK((int64_t)A * Bp * KH * KW <= INT32_MAX, "pack_weight: tensor too large");
    PackArgs a;
    a.w = w; a.out = out; a.w2 = w2; a.A = A; a.B = B; a.KH = KH; a.KW = KW; a.Bp = Bp; a.dtype = out_dtype;
    a.sA = sA; a.sB = sB; a.sKH = sKH; a.sKW = sKW; a.gain = gain;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t total = (int64_t)A * Bp;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 0.0, (double)A * B * KH * KW * (4.0 + sbg_dtype_size(out_dtype)), {A, B, KH * KW, 0, 0, 0, 0});
    dim3 grid(sbg_stream_grid(total, 256)), block(256);
    if (out_dtype == SBG_BF16)      hipLaunchKernelGGL(pack_weight_kernel<bf16_s>, grid, block, 0, stream, a);
    else if (out_dtype == SBG_F16)  hipLaunchKernelGGL(pack_weight_kernel<f16_s>, grid, block, 0, stream, a);
    else if (out_dtype == SBG_F32)  hipLaunchKernelGGL(pack_weight_kernel<float>, grid, block, 0, stream, a);
    else return sbg_fail(SBG_ERR_INVALID, "pack_weight: bad dtype %d", out_dtype);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_unpack_wgrad(const float* dwp, int64_t dwp_tap_stride, int64_t dwp_row_stride, float* dw, const float* w, const float* dw2,
                                int A, int B, int KH, int KW, int64_t sA, int64_t sB, int64_t sKH, int64_t sKW, float gain, sbg_stream_t stream_)
{
    SBG_CHECK(dw && (dwp || dw2), "unpack_wgrad: null pointer");
    SBG_CHECK(!dw2 || w, "unpack_wgrad: dw2 needs w");
    SBG_CHECK(A >= 1 && B >= 1 && KH >= 1 && KW >= 1, "unpack_wgrad: bad sizes");
    UnpackArgs a;
    a.dwp = dwp; a.dw = dw; a.w = w; a.dw2 = dw2; a.A = A; a.B = B; a.KH = KH; a.KW = KW;
    a.pT = dwp_tap_stride; a.pA = dwp_row_stride; a.sA = sA; a.sB = sB; a.sKH = sKH; a.sKW = sKW; a.gain = gain;
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 0.0, (double)A * B * KH * KW * 8.0, {A, B, KH * KW, 1, 0, 0, 0});
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(sbg_stream_grid((int64_t)A * B, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
