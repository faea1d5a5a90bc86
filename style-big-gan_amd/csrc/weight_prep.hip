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

// The same packing when the SOURCE is contiguous along the rows a (stride 1) and far-strided along the columns b -- the data-gradient operand
// [tap][Cin][Cout] of a channel-minor parameter [Cout][kh][kw][Cin]: with one lane per (a, b) pair the reads above are 4 bytes at an 18 KB stride.
// 32 x 32 tiles go through LDS instead: reads run along a, writes along b.  grid = (A / 32, Bp / 32, taps).
template <class T>
__global__ void __launch_bounds__(256) pack_weight_t_kernel(PackArgs p)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32, t = blockIdx.z;
    const int kh = t / p.KW, kw = t - kh * p.KW;
    const float* src = p.w + (int64_t)kh * p.sKH + (int64_t)kw * p.sKW;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = a0 + tx, b = b0 + ty + 8 * r;
        tile[ty + 8 * r][tx] = (a < p.A && b < p.B) ? src[(int64_t)a * p.sA + (int64_t)b * p.sB] * p.gain : 0.0f;
    }
    __syncthreads();
    T* out = (T*)p.out + (int64_t)t * p.A * p.Bp;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = a0 + ty + 8 * r, b = b0 + tx;
        if (a < p.A && b < p.Bp) Elem<T>::st(out + (int64_t)a * p.Bp + b, tile[tx][ty + 8 * r]);
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
    if (!w2 && sA == 1 && sB >= 64 && A >= 32 && B >= 32 && KH * KW <= 65535) {       // transposing form (see pack_weight_t_kernel)
        const dim3 tgrid((unsigned)((A + 31) / 32), (unsigned)((Bp + 31) / 32), (unsigned)(KH * KW));
        if (out_dtype == SBG_BF16)      SBG_LAUNCH(pack_weight_t_kernel<bf16_s>, tgrid, block, 0, stream, a);
        else if (out_dtype == SBG_F16)  SBG_LAUNCH(pack_weight_t_kernel<f16_s>, tgrid, block, 0, stream, a);
        else if (out_dtype == SBG_F32)  SBG_LAUNCH(pack_weight_t_kernel<float>, tgrid, block, 0, stream, a);
        else return sbg_fail(SBG_ERR_INVALID, "pack_weight: bad dtype %d", out_dtype);
        SBG_HIP_LAUNCH_CHECK();
        return 0;
    }
    if (out_dtype == SBG_BF16)      SBG_LAUNCH(pack_weight_kernel<bf16_s>, grid, block, 0, stream, a);
    else if (out_dtype == SBG_F16)  SBG_LAUNCH(pack_weight_kernel<f16_s>, grid, block, 0, stream, a);
    else if (out_dtype == SBG_F32)  SBG_LAUNCH(pack_weight_kernel<float>, grid, block, 0, stream, a);
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
    SBG_LAUNCH(unpack_wgrad_kernel, dim3(sbg_stream_grid((int64_t)A * B, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Demodulation coefficients from the tap-summed squared weights (generators.py:71-76 without the [N, O, I, k, k] tensor):
//   d[n, o] = rsqrt(sum_i s[n, i]^2 * w2[o, i] + eps)
// and their first-order gradient: with q[n, o] = -0.5 * g[n, o] * d[n, o]^3,
//   ds[n, i] = 2 s[n, i] * sum_o q[n, o] * w2[o, i],      dw2[o, i] = sum_n q[n, o] * s[n, i]^2.
namespace {

// one wavefront per (n, o): lanes stride over i (coalesced w2 row), shuffle reduction
__global__ void __launch_bounds__(256) demod_coefs_kernel(const float* __restrict__ s, const float* __restrict__ w2, float* __restrict__ d,
                                                          int N, int O, int I, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)N * O) return;
    const int n = (int)(pair / O), o = (int)(pair % O);
    const float* sr = s + (int64_t)n * I;
    const float* wr = w2 + (int64_t)o * I;
    float acc = 0.0f;
    for (int i = lane; i < I; i += 64) { const float sv = sr[i]; acc += sv * sv * wr[i]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) d[pair] = rsqrtf(acc + eps);
}

// ds: block = (n, 64 consecutive i); the four wavefronts split the sum over o (stride 4, eight loads in flight each), LDS combine
__global__ void __launch_bounds__(256) demod_bwd_styles_kernel(const float* __restrict__ g, const float* __restrict__ d, const float* __restrict__ s,
                                                               const float* __restrict__ w2, float* __restrict__ ds, int N, int O, int I)
{
    extern __shared__ float q[];            // [O] then 4 x 64 partial sums
    float* part = q + O;
    const int chunks = (I + 63) / 64;
    const int n = blockIdx.x / chunks, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = (blockIdx.x % chunks) * 64 + lane;
    for (int o = threadIdx.x; o < O; o += 256) { const float dv = d[(int64_t)n * O + o]; q[o] = -0.5f * g[(int64_t)n * O + o] * dv * dv * dv; }
    __syncthreads();
    float acc = 0.0f;
    if (i < I) {
        int o = wave;
        for (; o + 28 < O; o += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = w2[(int64_t)(o + 4 * u) * I + i];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += q[o + 4 * u] * v[u];
        }
        for (; o < O; o += 4) acc += q[o] * w2[(int64_t)o * I + i];
    }
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && i < I)
        ds[(int64_t)n * I + i] = 2.0f * s[(int64_t)n * I + i] * (part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane]);
}

// dw2: one lane per (o, i); the N-term sum reads s^2 coalesced and q broadcast
__global__ void __launch_bounds__(256) demod_bwd_w2_kernel(const float* __restrict__ g, const float* __restrict__ d, const float* __restrict__ s,
                                                           float* __restrict__ dw2, int N, int O, int I)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)O * I) return;
    const int o = (int)(idx / I), i = (int)(idx % I);
    float acc = 0.0f;
#pragma unroll 8
    for (int n = 0; n < N; n++) {
        const float dv = d[(int64_t)n * O + o], sv = s[(int64_t)n * I + i];
        acc += (-0.5f * g[(int64_t)n * O + o] * dv * dv * dv) * sv * sv;
    }
    dw2[idx] = acc;
}

}  // namespace

extern "C" int sbg_demod_coefs(const float* styles, const float* w2, float* dcoefs, int N, int O, int I, float eps, sbg_stream_t stream_)
{
    SBG_CHECK(styles && w2 && dcoefs && N >= 1 && O >= 1 && I >= 1, "demod_coefs: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 2.0 * N * O * (double)I, 4.0 * ((double)N * I + (double)O * I + (double)N * O), {N, O, I, 2, 0, 0, 0});
    SBG_LAUNCH(demod_coefs_kernel, dim3((unsigned)(((int64_t)N * O + 3) / 4)), dim3(256), 0, stream, styles, w2, dcoefs, N, O, I, eps);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_demod_coefs_bwd(const float* g, const float* dcoefs, const float* styles, const float* w2, float* dstyles, float* dw2,
                                   int N, int O, int I, sbg_stream_t stream_)
{
    SBG_CHECK(g && dcoefs && styles && w2 && (dstyles || dw2) && N >= 1 && O >= 1 && I >= 1, "demod_coefs_bwd: bad arguments");
    SBG_CHECK((O + 256) * sizeof(float) <= 64 * 1024, "demod_coefs_bwd: too many output channels");
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 4.0 * N * O * (double)I, 4.0 * (2.0 * N * I + 2.0 * O * I + 2.0 * N * O), {N, O, I, 3, 0, 0, 0});
    if (dstyles) {
        const int chunks = (I + 63) / 64;
        SBG_LAUNCH(demod_bwd_styles_kernel, dim3((unsigned)(N * chunks)), dim3(256), (O + 256) * sizeof(float), stream, g, dcoefs, styles, w2, dstyles, N, O, I);
        SBG_HIP_LAUNCH_CHECK();
    }
    if (dw2) {
        SBG_LAUNCH(demod_bwd_w2_kernel, dim3((unsigned)(((int64_t)O * I + 255) / 256)), dim3(256), 0, stream, g, dcoefs, styles, dw2, N, O, I);
        SBG_HIP_LAUNCH_CHECK();
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------
// fp32 -> bf16 hi / mid / lo split with the parts laid side by side along one axis, in one pass.
// fp32 tensors run through the bf16 matrix cores as sum_k conv(a_part_k, b_part_k) over the concatenated reduction axis (six products of
// three-part splits, conv2d_gradfix.py); building the concatenated operands with framework ops took ~10 launches per operand per launch.
//   x: [outer, C, inner] fp32 dense;  y: [outer, nseg * C, inner] bf16;  y[o, s*C + c, i] = part_{order[s]}(x[o, c, i]),
//   part_0 = bf16(x), part_1 = bf16(x - part_0), part_2 = bf16(x - part_0 - part_1).
namespace {
struct SplitArgs { const float* x; unsigned short* y; int64_t total; int64_t C, inner; int nseg; int order[8]; };

__global__ __launch_bounds__(256) void split_bf16_cat_kernel(SplitArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total; idx += stride) {
        const int64_t i = idx % a.inner, oc = idx / a.inner;
        const int64_t c = oc % a.C, o = oc / a.C;
        const float v = a.x[idx];
        unsigned short part[3];
        float r = v;
#pragma unroll
        for (int k = 0; k < 3; k++) { part[k] = f32_to_bf16_bits(r); r -= bf16_bits_to_f32(part[k]); }
        unsigned short* dst = a.y + (o * a.nseg * a.C + c) * a.inner + i;
        for (int s = 0; s < a.nseg; s++) dst[(int64_t)s * a.C * a.inner] = part[a.order[s]];
    }
}

// the same split of a STRIDED 4-D view, written densely in the view's own dimension order (a permuted weight view -> the packed
// [tap][cout][parts x cin] operand in one pass, instead of a split in memory order followed by a transposing copy of six times the data)
struct SplitNdArgs { const float* x; unsigned short* y; int64_t total; int64_t shape[4], xs[4]; int cat; int nseg; int order[8]; };

__global__ __launch_bounds__(256) void split_bf16_cat_nd_kernel(SplitNdArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total; idx += stride) {
        int64_t d[4], r = idx;
#pragma unroll
        for (int k = 3; k >= 0; k--) { d[k] = r % a.shape[k]; r /= a.shape[k]; }
        float v = a.x[d[0] * a.xs[0] + d[1] * a.xs[1] + d[2] * a.xs[2] + d[3] * a.xs[3]];
        unsigned short part[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { part[k] = f32_to_bf16_bits(v); v -= bf16_bits_to_f32(part[k]); }
        int64_t off = 0, seg = 1;                            // dense output offset of (d, part 0); distance between parts
#pragma unroll
        for (int k = 0; k < 4; k++) off = off * (k == a.cat ? a.shape[k] * a.nseg : a.shape[k]) + d[k];
#pragma unroll
        for (int k = 3; k >= 0; k--) { if (k == a.cat) { seg *= a.shape[k]; break; } seg *= a.shape[k]; }
        for (int s = 0; s < a.nseg; s++) a.y[off + (int64_t)s * seg] = part[a.order[s]];
    }
}
}

extern "C" int sbg_split_bf16_cat_nd(const float* x, const int64_t* shape, const int64_t* xstrides, int cat_dim, void* y, int nseg, const int* order,
                                     sbg_stream_t stream_)
{
    SBG_CHECK(x && y && order && shape && xstrides, "split_bf16_cat_nd: null pointer");
    SBG_CHECK(cat_dim >= 0 && cat_dim < 4 && nseg >= 1 && nseg <= 8, "split_bf16_cat_nd: bad sizes");
    SplitNdArgs a;
    a.x = x; a.y = (unsigned short*)y; a.cat = cat_dim; a.nseg = nseg; a.total = 1;
    for (int k = 0; k < 4; k++) {
        SBG_CHECK(shape[k] >= 0 && xstrides[k] >= 0, "split_bf16_cat_nd: negative extent or stride");
        a.shape[k] = shape[k]; a.xs[k] = xstrides[k]; a.total *= shape[k];
    }
    for (int s = 0; s < 8; s++) { a.order[s] = s < nseg ? order[s] : 0; SBG_CHECK(a.order[s] >= 0 && a.order[s] <= 2, "split_bf16_cat_nd: part index out of range"); }
    if (a.total == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 0.0, (double)a.total * (4.0 + 2.0 * nseg), {(int)shape[0], (int)shape[1], (int)shape[2], 5, nseg, (int)shape[3], 0});
    SBG_LAUNCH(split_bf16_cat_nd_kernel, dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_split_bf16_cat(const float* x, void* y, int64_t outer, int64_t C, int64_t inner, int nseg, const int* order, sbg_stream_t stream_)
{
    SBG_CHECK(x && y && order, "split_bf16_cat: null pointer");
    SBG_CHECK(outer >= 0 && C >= 1 && inner >= 1 && nseg >= 1 && nseg <= 8, "split_bf16_cat: bad sizes");
    SplitArgs a;
    a.x = x; a.y = (unsigned short*)y; a.total = outer * C * inner; a.C = C; a.inner = inner; a.nseg = nseg;
    for (int s = 0; s < 8; s++) { a.order[s] = s < nseg ? order[s] : 0; SBG_CHECK(a.order[s] >= 0 && a.order[s] <= 2, "split_bf16_cat: part index out of range"); }
    if (a.total == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_WEIGHT_PREP, 0.0, (double)a.total * (4.0 + 2.0 * nseg), {(int)(outer > INT32_MAX ? INT32_MAX : outer), (int)C, (int)inner, 4, nseg, 0, 0});
    SBG_LAUNCH(split_bf16_cat_kernel, dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
