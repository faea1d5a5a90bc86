// modulate.hip -- per-sample channel scaling (+ per-pixel addend) and its reduction gradient.
//
//   scale_nc : y[n,c,p] = x[n,c,p] * a[n,c] (+ z[n,p])      -- `x * styles` and `fma(x, dcoefs, noise)` of the modulated
//                                                              convolution (train_parts/generators.py:79-88, ops/fma.py:15)
//   dot_hw   : r[n,c]   = sum_p u[n,c,p] * v[n,c,p]         -- gradient w.r.t. a; with v == NULL the per-sample bias
//                                                              gradient of bias_act (ops/bias_act.py:172-173)
// Both are HBM-bound streaming ops: 16-B-per-lane accesses, grid sized to the chip, fp32 math.
// dot_hw reduces in two deterministic stages (per-workgroup partial rows, then the caller's fixed-order sum).
#include "sbg_common.h"

namespace {

struct ScaleArgs { const void* x; const float* a; const float* z; const float* bnc; void* y; int N, C; int64_t HW, zsn, total; };

// channel-minor: one lane = 8 channels of one pixel.
template <class T>
__global__ __launch_bounds__(256) void scale_nc_cminor8(ScaleArgs p)
{
    const T* px = (const T*)p.x; T* py = (T*)p.y;
    const int cv = p.C >> 3;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.total; i += step) {
        const int c = (int)(i % cv) << 3; const int64_t r = i / cv; const int64_t pix = r % p.HW; const int n = (int)(r / p.HW);
        float v[8], a[8];
        Vec8<T>::ld(px + (i << 3), v);
        Vec8<float>::ld(p.a + (int64_t)n * p.C + c, a);
        const float z = p.z ? p.z[n * p.zsn + pix] : 0.f;
        if (p.bnc) {
            float b[8];
            Vec8<float>::ld(p.bnc + (int64_t)n * p.C + c, b);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = v[j] * a[j] + (z + b[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = v[j] * a[j] + z;
        }
        Vec8<T>::st(py + (i << 3), v);
    }
}

// generic scalar form for either layout.
template <class T>
__global__ __launch_bounds__(256) void scale_nc_scalar(ScaleArgs p, int layout)
{
    const T* px = (const T*)p.x; T* py = (T*)p.y;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.total; i += step) {
        int c, n; int64_t pix;
        if (layout == 1) { c = (int)(i % p.C); const int64_t r = i / p.C; pix = r % p.HW; n = (int)(r / p.HW); }
        else             { pix = i % p.HW; const int64_t r = i / p.HW; c = (int)(r % p.C); n = (int)(r / p.C); }
        const float z = (p.z ? p.z[n * p.zsn + pix] : 0.f) + (p.bnc ? p.bnc[(int64_t)n * p.C + c] : 0.f);
        Elem<T>::st(py + i, Elem<T>::ld(px + i) * p.a[(int64_t)n * p.C + c] + z);
    }
}

// ---- dot_hw ---------------------------------------------------------------------------------------------------------
struct DotArgs { const void* u; const void* v; float* partial; int N, C; int64_t HW; int nsplit; int64_t pix_per_split;
                 const float* scale; void* y; };      // optional second result of the same pass: y = u * scale[n, c]  (sbg_dot_hw_scale)

#define DOT_PIX_PER_SPLIT 2048

// channel-minor, C % 8 == 0: workgroup = (sample n, pixel split s); lane = (channel vector, pixel lane).
template <class T>
__global__ __launch_bounds__(256) void dot_hw_cminor8(DotArgs p)
{
    __shared__ float red[256 * 8];
    const T* pu = (const T*)p.u; const T* pv = (const T*)p.v;
    const int n = blockIdx.x, s = blockIdx.y;
    const int cv = p.C >> 3;
    const int64_t p0 = (int64_t)s * p.pix_per_split;
    int64_t p1 = p0 + p.pix_per_split; if (p1 > p.HW) p1 = p.HW;
    const int64_t base = (int64_t)n * p.HW * p.C;
    // lanes own a fixed channel vector when 256 % cv == 0 (cv a power of two <= 256): 16-B coalesced loads, pixel lanes
    // stride the split; other channel counts walk whole channel vectors serially.
    if (cv <= 256 && (256 % cv) == 0) {
        const int myc = threadIdx.x % cv, plane = threadIdx.x / cv, planes = 256 / cv;
        float acc[8], sc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { acc[j] = 0.f; sc[j] = 0.f; }
        T* py = (T*)p.y;
        if (py) Vec8<float>::ld(p.scale + (int64_t)n * p.C + (myc << 3), sc);
        for (int64_t pix = p0 + plane; pix < p1; pix += planes) {
            float a[8], b[8];
            const int64_t off = base + pix * p.C + (myc << 3);
            Vec8<T>::ld(pu + off, a);
            if (pv) { Vec8<T>::ld(pv + off, b);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += a[j] * b[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] += a[j];
            }
            if (py) {
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] *= sc[j];
                Vec8<T>::st(py + off, a);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) red[threadIdx.x * 8 + j] = acc[j];
        __syncthreads();
        // fixed-order tree over the pixel lanes
        for (int stride = planes >> 1; stride >= 1; stride >>= 1) {
            if (plane < stride) {
#pragma unroll
                for (int j = 0; j < 8; j++) red[threadIdx.x * 8 + j] += red[(threadIdx.x + stride * cv) * 8 + j];
            }
            __syncthreads();
        }
        if (plane == 0) {
            float* dst = p.partial + ((int64_t)s * p.N + n) * p.C + (myc << 3);
#pragma unroll
            for (int j = 0; j < 8; j++) dst[j] = red[threadIdx.x * 8 + j];
        }
    } else {
        // many / odd channel vectors: each lane walks whole channel vectors serially over the split's pixels
        for (int myc = threadIdx.x; myc < cv; myc += 256) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.f;
            for (int64_t pix = p0; pix < p1; pix++) {
                float a[8], b[8];
                const int64_t off = base + pix * p.C + (myc << 3);
                Vec8<T>::ld(pu + off, a);
                if (pv) { Vec8<T>::ld(pv + off, b);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += a[j] * b[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += a[j];
                }
            }
            float* dst = p.partial + ((int64_t)s * p.N + n) * p.C + (myc << 3);
#pragma unroll
            for (int j = 0; j < 8; j++) dst[j] = acc[j];
        }
    }
}

// generic: workgroup = (n, c) pair-range; one wave-reduction per (n, c); handles both layouts element-wise.
template <class T>
__global__ __launch_bounds__(256) void dot_hw_generic(DotArgs p, int layout)
{
    __shared__ float red[256];
    const T* pu = (const T*)p.u; const T* pv = (const T*)p.v;
    const int64_t nc = blockIdx.x;                    // n*C + c
    const int n = (int)(nc / p.C), c = (int)(nc % p.C);
    float acc = 0.f;
    for (int64_t pix = threadIdx.x; pix < p.HW; pix += 256) {
        const int64_t off = (layout == 1) ? ((int64_t)n * p.HW + pix) * p.C + c : ((int64_t)n * p.C + c) * p.HW + pix;
        const float a = Elem<T>::ld(pu + off);
        acc += pv ? a * Elem<T>::ld(pv + off) : a;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride) red[threadIdx.x] += red[threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.partial[nc] = red[0];
}

// ---- modconv_bwd -----------------------------------------------------------------------------------------------------
// Backward head of the fused modulated-convolution layer  y = clamp(lrelu(c * dcoef[n,o] + noise[n,p] + b[o]) * gain):
// ONE pass over (dy, y) yields everything that does not need a convolution,
//     d1 = dy * gain * (y > 0 ? 1 : alpha) * [|y| < clamp]                (= dL/dpre, bias_act.py:159-210 on the saved output)
//     d2[n,o,p]        = d1 * dcoef[n,o]                                  (dL/dc, the input of the data / weight gradient convs)
//     part[0][s][n][o] = sum_{p in split s} d1                            (bias gradient, summed over n and s by the caller)
//     part[1][s][n][o] = sum_{p in split s} d1 * (pre - noise - b)        (= dcoef * sum d1 * c: demodulation gradient)
//     dnoise[n,p]      = sum_o d1                                         (optional)
// with pre recovered from the saved output (the activation is piecewise linear and invertible; clamped elements carry no
// gradient), so the pre-activation tensor c is never written or read.  Same workgroup shape and fixed-order reduction as
// dot_hw_cminor8: channel-minor, C / 8 a power of two <= 64.
struct ModBwdArgs {
    const void* dy; const void* y; const float* dcoef; const float* noise; const float* bias;
    void* d2; float* part; float* dnoise;
    int N, C; int64_t HW, nsn; int nsplit; int64_t pix_per_split;
    float alpha, gain, clamp;
    // PRE: `dy` is the gradient w.r.t. y * prescale[n, c] (y feeds a style-modulated convolution and nothing else): the kernel takes
    // sum_p dy * y (-> the gradient of prescale, part3[s][n][c]) and continues with dy * prescale -- the passes `dx = dy * s`, `sum dy * x` and this
    // kernel's own over (dx, y) become one (sbg_modconv_bwd_prescaled)
    const float* prescale; float* part3;
};

template <class T, bool PRE>
__global__ __launch_bounds__(256) void modconv_bwd_kernel(ModBwdArgs p)
{
    constexpr int RW = PRE ? 24 : 16;                   // floats per thread in the reduction buffer
    __shared__ float red[256 * RW];
    const T* pdy = (const T*)p.dy; const T* py = (const T*)p.y; T* pd2 = (T*)p.d2;
    const int n = blockIdx.x, s = blockIdx.y;
    const int cv = p.C >> 3;
    const int64_t p0 = (int64_t)s * p.pix_per_split;
    int64_t p1 = p0 + p.pix_per_split; if (p1 > p.HW) p1 = p.HW;
    const int64_t base = (int64_t)n * p.HW * p.C;
    const int myc = threadIdx.x % cv, plane = threadIdx.x / cv, planes = 256 / cv;
    float dc[8], bb[8], s1[8], s2[8], s3[8], ps[8];
    Vec8<float>::ld(p.dcoef + (int64_t)n * p.C + (myc << 3), dc);
#pragma unroll
    for (int j = 0; j < 8; j++) { bb[j] = p.bias ? p.bias[(myc << 3) + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; s3[j] = 0.f; ps[j] = 1.f; }
    if (PRE) Vec8<float>::ld(p.prescale + (int64_t)n * p.C + (myc << 3), ps);
    const float inv_pos = 1.f / p.gain, inv_neg = p.alpha > 0.f ? 1.f / (p.gain * p.alpha) : 0.f;
    const float gpos = p.gain, gneg = p.gain * p.alpha;
    const float cl = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    // every lane of a wave runs the same number of iterations (pixel lanes of one wave differ by < planes), so the
    // cross-lane channel sum below sees all lanes of a pixel
    const int64_t iters = (p1 - p0 + planes - 1) / planes;
    for (int64_t it = 0; it < iters; it++) {
        const int64_t pix = p0 + it * planes + plane;
        const bool ok = pix < p1;
        float g[8], yv[8], o[8];
        float dn = 0.f;
        if (ok) {
            const int64_t off = base + pix * p.C + (myc << 3);
            Vec8<T>::ld(pdy + off, g);
            Vec8<T>::ld(py + off, yv);
            const float nz = p.noise ? p.noise[n * p.nsn + pix] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (PRE) { s3[j] += g[j] * yv[j]; g[j] *= ps[j]; }
                const bool pos = yv[j] > 0.f;
                const bool live = (yv[j] > -cl) && (yv[j] < cl);
                const float d1 = live ? g[j] * (pos ? gpos : gneg) : 0.f;
                const float pre = yv[j] * (pos ? inv_pos : inv_neg);
                s1[j] += d1;
                s2[j] += d1 * (pre - nz - bb[j]);
                dn += d1;
                o[j] = d1 * dc[j];
            }
            Vec8<T>::st(pd2 + off, o);
        }
        if (p.dnoise) {                                 // sum over the cv lanes that hold this pixel's channels (cv <= 64, power of two)
            for (int m = 1; m < cv; m <<= 1) dn += __shfl_xor(dn, m, 64);
            if (ok && myc == 0) p.dnoise[(int64_t)n * p.HW + pix] = dn;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { red[threadIdx.x * RW + j] = s1[j]; red[threadIdx.x * RW + 8 + j] = s2[j]; if (PRE) red[threadIdx.x * RW + 16 + j] = s3[j]; }
    __syncthreads();
    for (int stride = planes >> 1; stride >= 1; stride >>= 1) {      // fixed-order tree over the pixel lanes
        if (plane < stride) {
#pragma unroll
            for (int j = 0; j < RW; j++) red[threadIdx.x * RW + j] += red[(threadIdx.x + stride * cv) * RW + j];
        }
        __syncthreads();
    }
    if (plane == 0) {
        float* d0 = p.part + ((int64_t)s * p.N + n) * p.C + (myc << 3);
        float* d1p = d0 + (int64_t)p.nsplit * p.N * p.C;
#pragma unroll
        for (int j = 0; j < 8; j++) { d0[j] = red[threadIdx.x * RW + j]; d1p[j] = red[threadIdx.x * RW + 8 + j]; }
        if (PRE) {
            float* d3 = p.part3 + ((int64_t)s * p.N + n) * p.C + (myc << 3);
#pragma unroll
            for (int j = 0; j < 8; j++) d3[j] = red[threadIdx.x * RW + 16 + j];
        }
    }
}

static bool dot_fast(int layout, int C) { return layout == 1 && (C % 8) == 0; }

template <class T>
static int run_scale(const ScaleArgs& a0, int layout, bool vec, hipStream_t s)
{
    ScaleArgs a = a0;
    const double es = sizeof(T) == 4 ? 4 : 2;
    SbgProfScope prof(s, SBG_K_SCALE_NC, 0.0, 2.0 * es * a.N * (double)a.C * a.HW, {a.N, a.C, (int)a.HW, layout});
    if (vec) {
        a.total = (int64_t)a.N * a.HW * (a.C >> 3);
        SBG_LAUNCH((scale_nc_cminor8<T>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, s, a);
    } else {
        a.total = (int64_t)a.N * a.HW * a.C;
        SBG_LAUNCH((scale_nc_scalar<T>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, s, a, layout);
    }
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

template <class T>
static int run_dot(const DotArgs& a, int layout, bool fast, hipStream_t s)
{
    const double es = sizeof(T) == 4 ? 4 : 2;
    SbgProfScope prof(s, SBG_K_DOT_HW, 0.0, (a.v ? 2.0 : 1.0) * es * a.N * (double)a.C * a.HW, {a.N, a.C, (int)a.HW, layout});
    if (fast) SBG_LAUNCH((dot_hw_cminor8<T>), dim3(a.N, a.nsplit), dim3(256), 0, s, a);
    else      SBG_LAUNCH((dot_hw_generic<T>), dim3((unsigned)((int64_t)a.N * a.C)), dim3(256), 0, s, a, layout);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

static int scale_impl(const void* x, const float* a, const float* z, const float* bnc, void* y, int dtype, int layout,
                      int N, int C, int64_t HW, int64_t z_stride_n, sbg_stream_t stream)
{
    SBG_CHECK(x && a && y, "scale_nc: null pointer");
    SBG_CHECK(dtype == SBG_F32 || dtype == SBG_F16 || dtype == SBG_BF16, "scale_nc: unsupported dtype %d", dtype);
    SBG_CHECK(layout == 0 || layout == 1, "scale_nc: layout must be 0 (planar) or 1 (channel-minor)");
    SBG_CHECK(N >= 0 && C >= 1 && HW >= 0, "scale_nc: bad sizes");
    if ((int64_t)N * C * HW == 0) return SBG_OK;
    ScaleArgs p; p.x = x; p.a = a; p.z = z; p.bnc = bnc; p.y = y; p.N = N; p.C = C; p.HW = HW; p.zsn = z_stride_n; p.total = 0;
    const bool vec = layout == 1 && (C % 8) == 0 && sbg_aligned16(x) && sbg_aligned16(y) && sbg_aligned16(a) && (!bnc || sbg_aligned16(bnc));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SBG_F32) return run_scale<float>(p, layout, vec, s);
    if (dtype == SBG_F16) return run_scale<f16_s>(p, layout, vec, s);
    return run_scale<bf16_s>(p, layout, vec, s);
}

extern "C" int sbg_scale_nc(const void* x, const float* a, const float* z, void* y, int dtype, int layout,
                            int N, int C, int64_t HW, int64_t z_stride_n, sbg_stream_t stream)
{
    return scale_impl(x, a, z, nullptr, y, dtype, layout, N, C, HW, z_stride_n, stream);
}

extern "C" int sbg_scale_shift_nc(const void* x, const float* a, const float* b, void* y, int dtype, int layout,
                                  int N, int C, int64_t HW, sbg_stream_t stream)
{
    SBG_CHECK(b != nullptr, "scale_shift_nc: null shift");
    return scale_impl(x, a, nullptr, b, y, dtype, layout, N, C, HW, 0, stream);
}

extern "C" int sbg_dot_hw_splits(int layout, int N, int C, int64_t HW)
{
    if (!dot_fast(layout, C)) return 1;
    // enough (sample, split) workgroups to fill the chip (~4 per CU), at least 64 pixels of work each
    int64_t want = (1024 + N - 1) / (N > 0 ? N : 1);
    const int64_t most = (HW + 63) / 64;
    if (want > most) want = most;
    if (want < 1) want = 1;
    const int64_t pps = (HW + want - 1) / want;
    return (int)((HW + pps - 1) / pps);
}

extern "C" int sbg_dot_hw(const void* u, const void* v, float* partial, int dtype, int layout,
                          int N, int C, int64_t HW, sbg_stream_t stream)
{
    SBG_CHECK(u && partial, "dot_hw: null pointer");
    SBG_CHECK(dtype == SBG_F32 || dtype == SBG_F16 || dtype == SBG_BF16, "dot_hw: unsupported dtype %d", dtype);
    SBG_CHECK(layout == 0 || layout == 1, "dot_hw: layout must be 0 (planar) or 1 (channel-minor)");
    SBG_CHECK(N >= 1 && C >= 1 && HW >= 1, "dot_hw: bad sizes");
    SBG_CHECK((int64_t)N * C <= INT32_MAX, "dot_hw: too many (sample, channel) pairs");
    DotArgs p; p.u = u; p.v = v; p.partial = partial; p.N = N; p.C = C; p.HW = HW; p.scale = nullptr; p.y = nullptr;
    const bool fast = dot_fast(layout, C) && sbg_aligned16(u) && (!v || sbg_aligned16(v));
    p.nsplit = dot_fast(layout, C) ? sbg_dot_hw_splits(layout, N, C, HW) : 1;
    p.pix_per_split = (HW + p.nsplit - 1) / p.nsplit;
    hipStream_t s = (hipStream_t)stream;
    if (!fast && p.nsplit > 1) {
        // unaligned view of a channel-minor tensor: the generic kernel writes split 0 only; zero the others
        if (hipMemsetAsync(partial, 0, sizeof(float) * (size_t)p.nsplit * N * C, s) != hipSuccess) return sbg_fail(SBG_ERR_LAUNCH, "dot_hw: memset failed");
    }
    if (dtype == SBG_F32) return run_dot<float>(p, layout, fast, s);
    if (dtype == SBG_F16) return run_dot<f16_s>(p, layout, fast, s);
    return run_dot<bf16_s>(p, layout, fast, s);
}


// ---- moments_hw: r[0][n,c] = sum_p x, r[1][n,c] = sum_p x^2 in ONE pass over a planar tensor (the batch statistics of BigGAN's normalisation
// layers: reference biggan/layers.py:188-205 `manual_bn`, sync_batchnorm/batchnorm.py:71-79 -- two reductions over x there, two dot_hw launches
// here until round 3).  One workgroup per (n, c) plane, 16-B loads when the plane allows, fp32 accumulation, fixed-order tree.
template <class T>
__global__ __launch_bounds__(256) void moments_hw_planar(const T* x, float* r, int64_t NC, int64_t HW, int vec)
{
    __shared__ float red[2][256];
    const int64_t nc = blockIdx.x;
    const T* px = x + nc * HW;
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        const int64_t nv = HW >> 3;
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
            float v[8];
            Vec8<T>::ld(px + i * 8, v);
#pragma unroll
            for (int e = 0; e < 8; e++) { s1 += v[e]; s2 += v[e] * v[e]; }
        }
    } else {
        for (int64_t i = threadIdx.x; i < HW; i += 256) { const float a = Elem<T>::ld(px + i); s1 += a; s2 += a * a; }
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride) { red[0][threadIdx.x] += red[0][threadIdx.x + stride]; red[1][threadIdx.x] += red[1][threadIdx.x + stride]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { r[nc] = red[0][0]; r[NC + nc] = red[1][0]; }
}

extern "C" int sbg_moments_hw(const void* x, float* r, int dtype, int N, int C, int64_t HW, sbg_stream_t stream)
{
    SBG_CHECK(x && r, "moments_hw: null pointer");
    SBG_CHECK(dtype == SBG_F32 || dtype == SBG_F16 || dtype == SBG_BF16, "moments_hw: unsupported dtype %d", dtype);
    SBG_CHECK(N >= 1 && C >= 1 && HW >= 1 && (int64_t)N * C <= INT32_MAX, "moments_hw: bad sizes");
    const int64_t NC = (int64_t)N * C;
    const int vec = sbg_aligned16(x) && (HW % 8) == 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SBG_F32)      SBG_LAUNCH((moments_hw_planar<float>), dim3((unsigned)NC), dim3(256), 0, s, (const float*)x, r, NC, HW, vec);
    else if (dtype == SBG_F16) SBG_LAUNCH((moments_hw_planar<f16_s>), dim3((unsigned)NC), dim3(256), 0, s, (const f16_s*)x, r, NC, HW, vec);
    else                       SBG_LAUNCH((moments_hw_planar<bf16_s>), dim3((unsigned)NC), dim3(256), 0, s, (const bf16_s*)x, r, NC, HW, vec);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

// One pass over (u, v) for both gradients of y = x * a[n, c] given u = dy, v = x:  partial sums of u * v (-> da) AND dx = u * scale.
extern "C" int sbg_dot_hw_scale_supported(int C)
{
    const int cv = C >> 3;
    return (C % 8) == 0 && cv >= 1 && cv <= 256 && (256 % cv) == 0;       // the lane-owns-a-channel-vector form of dot_hw_cminor8
}

extern "C" int sbg_dot_hw_scale(const void* u, const void* v, const float* scale, void* y, float* partial, int dtype,
                                int N, int C, int64_t HW, sbg_stream_t stream)
{
    SBG_CHECK(u && v && scale && y && partial, "dot_hw_scale: null pointer");
    SBG_CHECK(dtype == SBG_F32 || dtype == SBG_F16 || dtype == SBG_BF16, "dot_hw_scale: unsupported dtype %d", dtype);
    SBG_CHECK(N >= 1 && HW >= 1 && sbg_dot_hw_scale_supported(C) && dot_fast(1, C), "dot_hw_scale: channel-minor tensors with C / 8 dividing 256 (got C = %d)", C);
    SBG_CHECK((int64_t)N * C <= INT32_MAX, "dot_hw_scale: too many (sample, channel) pairs");
    SBG_CHECK(sbg_aligned16(u) && sbg_aligned16(v) && sbg_aligned16(y) && sbg_aligned16(scale), "dot_hw_scale: tensors must be 16-byte aligned");
    DotArgs p; p.u = u; p.v = v; p.partial = partial; p.N = N; p.C = C; p.HW = HW; p.scale = scale; p.y = y;
    p.nsplit = sbg_dot_hw_splits(1, N, C, HW);
    p.pix_per_split = (HW + p.nsplit - 1) / p.nsplit;
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == SBG_F32 ? 4 : 2;
    SbgProfScope prof(s, SBG_K_DOT_HW, 0.0, 3.0 * es * N * (double)C * HW, {N, C, (int)HW, 3});
    if (dtype == SBG_F32)      SBG_LAUNCH((dot_hw_cminor8<float>), dim3(N, p.nsplit), dim3(256), 0, s, p);
    else if (dtype == SBG_F16) SBG_LAUNCH((dot_hw_cminor8<f16_s>), dim3(N, p.nsplit), dim3(256), 0, s, p);
    else                       SBG_LAUNCH((dot_hw_cminor8<bf16_s>), dim3(N, p.nsplit), dim3(256), 0, s, p);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

extern "C" int sbg_modconv_bwd_supported(int C)
{
    const int cv = C >> 3;
    return (C % 8) == 0 && cv >= 1 && cv <= 64 && (cv & (cv - 1)) == 0;
}

static int modconv_bwd_impl(const void* dy, const void* y, const float* prescale, const float* dcoef, const float* noise, const float* bias,
                            void* d2, float* partial, float* partial3, float* dnoise, int dtype, int N, int C, int64_t HW, int64_t noise_stride_n,
                            int act, float alpha, float gain, float clamp, sbg_stream_t stream);

extern "C" int sbg_modconv_bwd(const void* dy, const void* y, const float* dcoef, const float* noise, const float* bias,
                               void* d2, float* partial, float* dnoise, int dtype, int N, int C, int64_t HW, int64_t noise_stride_n,
                               int act, float alpha, float gain, float clamp, sbg_stream_t stream)
{
    return modconv_bwd_impl(dy, y, nullptr, dcoef, noise, bias, d2, partial, nullptr, dnoise, dtype, N, C, HW, noise_stride_n, act, alpha, gain, clamp, stream);
}

extern "C" int sbg_modconv_bwd_prescaled(const void* dy, const void* y, const float* prescale, const float* dcoef, const float* noise, const float* bias,
                                         void* d2, float* partial, float* partial3, float* dnoise, int dtype, int N, int C, int64_t HW,
                                         int64_t noise_stride_n, int act, float alpha, float gain, float clamp, sbg_stream_t stream)
{
    SBG_CHECK(prescale && partial3 && sbg_aligned16(prescale), "modconv_bwd_prescaled: prescale (16-byte aligned) and partial3 are required");
    return modconv_bwd_impl(dy, y, prescale, dcoef, noise, bias, d2, partial, partial3, dnoise, dtype, N, C, HW, noise_stride_n, act, alpha, gain, clamp, stream);
}

static int modconv_bwd_impl(const void* dy, const void* y, const float* prescale, const float* dcoef, const float* noise, const float* bias,
                            void* d2, float* partial, float* partial3, float* dnoise, int dtype, int N, int C, int64_t HW, int64_t noise_stride_n,
                            int act, float alpha, float gain, float clamp, sbg_stream_t stream)
{
    SBG_CHECK(dy && y && dcoef && d2 && partial, "modconv_bwd: null pointer");
    SBG_CHECK(dtype == SBG_F16 || dtype == SBG_BF16 || dtype == SBG_F32, "modconv_bwd: unsupported dtype %d", dtype);
    SBG_CHECK(sbg_modconv_bwd_supported(C), "modconv_bwd: C / 8 must be a power of two <= 64 (got C = %d)", C);
    SBG_CHECK(act == SBG_ACT_LINEAR || act == SBG_ACT_RELU || act == SBG_ACT_LRELU, "modconv_bwd: activation must be linear, relu or lrelu");
    SBG_CHECK(N >= 1 && HW >= 1 && gain > 0.f, "modconv_bwd: bad sizes / gain");
    SBG_CHECK(sbg_aligned16(dy) && sbg_aligned16(y) && sbg_aligned16(d2) && sbg_aligned16(dcoef), "modconv_bwd: tensors must be 16-byte aligned");
    ModBwdArgs p;
    p.dy = dy; p.y = y; p.dcoef = dcoef; p.noise = noise; p.bias = bias; p.d2 = d2; p.part = partial; p.dnoise = dnoise;
    p.prescale = prescale; p.part3 = partial3;
    p.N = N; p.C = C; p.HW = HW; p.nsn = noise_stride_n;
    p.nsplit = sbg_dot_hw_splits(1, N, C, HW);
    p.pix_per_split = (HW + p.nsplit - 1) / p.nsplit;
    p.alpha = act == SBG_ACT_LRELU ? alpha : (act == SBG_ACT_RELU ? 0.f : 1.f); p.gain = gain; p.clamp = clamp;
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == SBG_F32 ? 4 : 2;
    SbgProfScope prof(s, SBG_K_DOT_HW, 0.0, 3.0 * es * N * (double)C * HW, {N, C, (int)HW, prescale ? 4 : 2});
    if (prescale) {
        if (dtype == SBG_F32)      SBG_LAUNCH((modconv_bwd_kernel<float, true>), dim3(N, p.nsplit), dim3(256), 0, s, p);
        else if (dtype == SBG_F16) SBG_LAUNCH((modconv_bwd_kernel<f16_s, true>), dim3(N, p.nsplit), dim3(256), 0, s, p);
        else                       SBG_LAUNCH((modconv_bwd_kernel<bf16_s, true>), dim3(N, p.nsplit), dim3(256), 0, s, p);
    } else {
        if (dtype == SBG_F32)      SBG_LAUNCH((modconv_bwd_kernel<float, false>), dim3(N, p.nsplit), dim3(256), 0, s, p);
        else if (dtype == SBG_F16) SBG_LAUNCH((modconv_bwd_kernel<f16_s, false>), dim3(N, p.nsplit), dim3(256), 0, s, p);
        else                       SBG_LAUNCH((modconv_bwd_kernel<bf16_s, false>), dim3(N, p.nsplit), dim3(256), 0, s, p);
    }
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}
