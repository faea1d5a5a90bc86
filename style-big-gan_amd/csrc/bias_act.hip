// bias_act.hip -- fused bias + activation + gain + clamp and its first/second derivative forms.
//
// Semantics follow the reference op (stylegan2ada/torch_utils/ops/bias_act.py:55-210 and the element
// formulae of bias_act.cu:23-146); the kernel itself is a gfx950 streaming design: 8 elements per lane
// per step (16 B for bf16/f16, 2 x 16 B for fp32), grid-stride over a grid sized to the chip
// (256 CUs x 8 workgroups), bias gathered once per 8-vector whenever the layout allows it.
// HBM-bound: algorithmic bytes = (1 + #aux inputs + 1) * numel * sizeof(T).
#include "sbg_common.h"

namespace {

struct BiasActArgs {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int grad; float alpha, gain, clamp;
    int64_t sizeX; int sizeB; int64_t stepB;
    int bmode;   // 0 none, 1 channel-minor (stepB == 1, sizeB % 8 == 0), 2 one bias per 8-vector, 3 generic
};

template <int A>
static __device__ __forceinline__ float act_elem(int G, float x, float xref, float& yref, float yy, float alpha, float gain)
{
    const float expRange = 80.f, halfExpRange = 40.f;
    const float seluScale = 1.0507009873554804934193349852946f;
    const float seluAlpha = 1.6732632423543772848170429916717f;
    float y = 0.f;
    if (A == SBG_ACT_LINEAR) { if (G <= 1) y = x; }
    if (A == SBG_ACT_RELU)   { if (G == 0) y = (x > 0.f) ? x : 0.f; if (G == 1) y = (yy > 0.f) ? x : 0.f; }
    if (A == SBG_ACT_LRELU)  { if (G == 0) y = (x > 0.f) ? x : x * alpha; if (G == 1) y = (yy > 0.f) ? x : x * alpha; }
    if (A == SBG_ACT_TANH) {
        if (G == 0) { float c = expf(x), d = 1.f / c; y = (x < -expRange) ? -1.f : (x > expRange) ? 1.f : (c - d) / (c + d); }
        if (G == 1) y = x * (1.f - yy * yy);
        if (G == 2) y = x * (1.f - yy * yy) * (-2.f * yy);
    }
    if (A == SBG_ACT_SIGMOID) {
        if (G == 0) y = (x < -expRange) ? 0.f : 1.f / (expf(-x) + 1.f);
        if (G == 1) y = x * yy * (1.f - yy);
        if (G == 2) y = x * yy * (1.f - yy) * (1.f - 2.f * yy);
    }
    if (A == SBG_ACT_ELU) {
        if (G == 0) y = (x >= 0.f) ? x : expf(x) - 1.f;
        if (G == 1) y = (yy >= 0.f) ? x : x * (yy + 1.f);
        if (G == 2) y = (yy >= 0.f) ? 0.f : x * (yy + 1.f);
    }
    if (A == SBG_ACT_SELU) {
        if (G == 0) y = (x >= 0.f) ? seluScale * x : (seluScale * seluAlpha) * (expf(x) - 1.f);
        if (G == 1) y = (yy >= 0.f) ? x * seluScale : x * (yy + seluScale * seluAlpha);
        if (G == 2) y = (yy >= 0.f) ? 0.f : x * (yy + seluScale * seluAlpha);
    }
    if (A == SBG_ACT_SOFTPLUS) {
        if (G == 0) y = (x > expRange) ? x : logf(expf(x) + 1.f);
        if (G == 1) y = x * (1.f - expf(-yy));
        if (G == 2) { float c = expf(-yy); y = x * c * (1.f - c); }
    }
    if (A == SBG_ACT_SWISH) {
        if (G == 0) y = (x < -expRange) ? 0.f : x / (expf(-x) + 1.f);
        else {
            float c = expf(xref), d = c + 1.f;
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);
            else        y = (xref > halfExpRange) ? 0.f : x * c * (xref * (2.f - d) + 2.f * d) / (d * d * d);
            yref = (xref < -expRange) ? 0.f : xref / (expf(-xref) + 1.f) * gain;
        }
    }
    return y;
}

template <int A>
static __device__ __forceinline__ float bias_act_elem(int G, float x, float b, float xref, float yref, float dy,
                                                      float alpha, float gain, float clamp)
{
    if (G == 0) x += b; else xref += b;
    float yy = (gain != 0.f) ? yref / gain : 0.f;
    float y = act_elem<A>(G, x, xref, yref, yy, alpha, gain);
    y *= gain * dy;
    if (clamp >= 0.f) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0.f) ? clamp : -clamp;
        else        y = (yref > -clamp && yref < clamp) ? y : 0.f;
    }
    return y;
}

// Vector kernel: every pointer 16-B aligned, 8 elements per lane per step.
template <class T, int A>
__global__ __launch_bounds__(256) void bias_act_vec8(BiasActArgs p)
{
    const T* px = (const T*)p.x; const T* pb = (const T*)p.b; const T* pxr = (const T*)p.xref;
    const T* pyr = (const T*)p.yref; const T* pdy = (const T*)p.dy; T* py = (T*)p.y;
    const int G = p.grad;
    const int64_t nvec = p.sizeX >> 3;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += step) {
        const int64_t i0 = v << 3;
        float x[8], b[8], xr[8], yr[8], dy[8], y[8];
        Vec8<T>::ld(px + i0, x);
#pragma unroll
        for (int j = 0; j < 8; j++) { b[j] = 0.f; xr[j] = 0.f; yr[j] = 0.f; dy[j] = 1.f; }
        if (pxr) Vec8<T>::ld(pxr + i0, xr);
        if (pyr) Vec8<T>::ld(pyr + i0, yr);
        if (pdy) Vec8<T>::ld(pdy + i0, dy);
        if (p.bmode == 1) {
            Vec8<T>::ld(pb + (i0 % p.sizeB), b);
        } else if (p.bmode == 2) {
            float bv = Elem<T>::ld(pb + ((i0 / p.stepB) % p.sizeB));
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = bv;
        } else if (p.bmode == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = Elem<T>::ld(pb + (((i0 + j) / p.stepB) % p.sizeB));
        }
#pragma unroll
        for (int j = 0; j < 8; j++) y[j] = bias_act_elem<A>(G, x[j], b[j], xr[j], yr[j], dy[j], p.alpha, p.gain, p.clamp);
        Vec8<T>::st(py + i0, y);
    }
    // Tail (< 8 elements): first lanes of workgroup 0.
    const int64_t tail0 = nvec << 3;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < p.sizeX) {
        const int64_t i = tail0 + threadIdx.x;
        float b = pb ? Elem<T>::ld(pb + ((i / p.stepB) % p.sizeB)) : 0.f;
        float y = bias_act_elem<A>(G, Elem<T>::ld(px + i), b, pxr ? Elem<T>::ld(pxr + i) : 0.f, pyr ? Elem<T>::ld(pyr + i) : 0.f,
                                   pdy ? Elem<T>::ld(pdy + i) : 1.f, p.alpha, p.gain, p.clamp);
        Elem<T>::st(py + i, y);
    }
}

// Scalar kernel for unaligned views.
template <class T, int A>
__global__ __launch_bounds__(256) void bias_act_scalar(BiasActArgs p)
{
    const T* px = (const T*)p.x; const T* pb = (const T*)p.b; const T* pxr = (const T*)p.xref;
    const T* pyr = (const T*)p.yref; const T* pdy = (const T*)p.dy; T* py = (T*)p.y;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.sizeX; i += step) {
        float b = pb ? Elem<T>::ld(pb + ((i / p.stepB) % p.sizeB)) : 0.f;
        float y = bias_act_elem<A>(p.grad, Elem<T>::ld(px + i), b, pxr ? Elem<T>::ld(pxr + i) : 0.f, pyr ? Elem<T>::ld(pyr + i) : 0.f,
                                   pdy ? Elem<T>::ld(pdy + i) : 1.f, p.alpha, p.gain, p.clamp);
        Elem<T>::st(py + i, y);
    }
}

template <class T, int A>
static int launch_bias_act(const BiasActArgs& p, bool vec, hipStream_t stream)
{
    const int es = sizeof(T) == 4 ? 4 : 2;
    const int streams = 2 + (p.xref ? 1 : 0) + (p.yref ? 1 : 0) + (p.dy ? 1 : 0);
    SbgProfScope prof(stream, SBG_K_BIAS_ACT, 0.0, (double)p.sizeX * es * streams, {(int)p.sizeX, p.grad, A, es});
    if (vec) {
        unsigned grid = sbg_stream_grid((p.sizeX >> 3) + 1, 256);
        SBG_LAUNCH((bias_act_vec8<T, A>), dim3(grid), dim3(256), 0, stream, p);
    } else {
        unsigned grid = sbg_stream_grid(p.sizeX, 256);
        SBG_LAUNCH((bias_act_scalar<T, A>), dim3(grid), dim3(256), 0, stream, p);
    }
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

template <class T>
static int dispatch_act(const BiasActArgs& p, int act, bool vec, hipStream_t stream)
{
    switch (act) {
        case SBG_ACT_LINEAR:   return launch_bias_act<T, SBG_ACT_LINEAR>(p, vec, stream);
        case SBG_ACT_RELU:     return launch_bias_act<T, SBG_ACT_RELU>(p, vec, stream);
        case SBG_ACT_LRELU:    return launch_bias_act<T, SBG_ACT_LRELU>(p, vec, stream);
        case SBG_ACT_TANH:     return launch_bias_act<T, SBG_ACT_TANH>(p, vec, stream);
        case SBG_ACT_SIGMOID:  return launch_bias_act<T, SBG_ACT_SIGMOID>(p, vec, stream);
        case SBG_ACT_ELU:      return launch_bias_act<T, SBG_ACT_ELU>(p, vec, stream);
        case SBG_ACT_SELU:     return launch_bias_act<T, SBG_ACT_SELU>(p, vec, stream);
        case SBG_ACT_SOFTPLUS: return launch_bias_act<T, SBG_ACT_SOFTPLUS>(p, vec, stream);
        case SBG_ACT_SWISH:    return launch_bias_act<T, SBG_ACT_SWISH>(p, vec, stream);
    }
    return sbg_fail(SBG_ERR_INVALID, "bias_act: no kernel for activation id %d", act);
}

} // namespace

extern "C" int sbg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                            void* y, int dtype, int grad, int act, float alpha, float gain, float clamp,
                            int64_t sizeX, int sizeB, int64_t stepB, sbg_stream_t stream)
{
    SBG_CHECK(x != nullptr && y != nullptr, "bias_act: x and y must be device pointers");
    SBG_CHECK(sizeX >= 0 && sizeX <= INT32_MAX, "bias_act: x is too large");
    SBG_CHECK(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2");
    SBG_CHECK(dtype == SBG_F32 || dtype == SBG_F16 || dtype == SBG_BF16, "bias_act: unsupported dtype %d", dtype);
    SBG_CHECK(b == nullptr || (sizeB >= 1 && stepB >= 1), "bias_act: b has wrong number of elements");
    if (sizeX == 0) return SBG_OK;

    BiasActArgs p;
    p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y;
    p.grad = grad; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.sizeX = sizeX; p.sizeB = b ? sizeB : 1; p.stepB = b ? stepB : 1;
    p.bmode = 0;
    if (b) {
        if (stepB == 1 && (sizeB % 8) == 0) p.bmode = 1;
        else if ((stepB % 8) == 0)          p.bmode = 2;
        else                                p.bmode = 3;
    }
    bool vec = sbg_aligned16(x) && sbg_aligned16(y) && (!xref || sbg_aligned16(xref)) && (!yref || sbg_aligned16(yref)) &&
               (!dy || sbg_aligned16(dy)) && (p.bmode != 1 || sbg_aligned16(b));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SBG_F32)  return dispatch_act<float>(p, act, vec, s);
    if (dtype == SBG_F16)  return dispatch_act<f16_s>(p, act, vec, s);
    return dispatch_act<bf16_s>(p, act, vec, s);
}
