// fromrgb.hip -- the discriminator's fromRGB layer as streaming kernels.
//
// DiscriminatorBlock.fromrgb (train_parts/discriminators.py:270-277 -> Conv2dLayer :115-124) = 1x1 convolution from the 3 image
// channels + bias + leaky ReLU * gain + clamp.  With K = 3 a GEMM tile is 95 % padding and the image has to be cast and re-laid
// out first; here the fp32 planar image is read as it is and the 16-bit channel-minor activation is written once:
//   forward : y[n, p, co] = clamp(act(sum_c img[n, c, p] * w[co, c] + b[co]) * gain)                 -- HBM-bound, numel(y) * 2 B
//   backward: d1 = dy * act'(y) masked by the clamp (from the saved y);  dw[co, c] = sum d1 * img,  db[co] = sum d1,
//             dimg[n, c, p] = sum_co d1 * w[co, c] (optional)                                        -- HBM-bound, numel(y) * 4 B
// Lane mapping as in torgb.hip: Co / 8 lanes share a pixel, 8 channels (16 B) each; a workgroup stages the image values of 256
// pixels in LDS with coalesced planar loads, then every lane needs one LDS read per pixel.  Co = 8 * 2^k <= 512, Cin <= 4.
#include "sbg_common.h"

namespace {

constexpr int MAX_CI = 4;

struct FromArgs {
    const float* img; const float* w; const float* bias; const void* dy; const void* ysaved;
    void* y; float* dimg; float* partial;
    int N, Ci, Co; int64_t HW;
    int act; float alpha, gain, clamp;
    int blocks_per_n;
};

static __device__ __forceinline__ float act_fwd(float v, int act, float alpha)
{
    if (act == SBG_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == SBG_ACT_LRELU) return v > 0.0f ? v : v * alpha;
    return v;
}
// derivative of clamp(act(x) * gain) with respect to x, from the OUTPUT y (piecewise linear activations; bias_act.cu:141 conventions)
static __device__ __forceinline__ float act_grad_from_y(float y, int act, float alpha, float gain, float clamp)
{
    if (clamp >= 0.0f && !(fabsf(y) < clamp)) return 0.0f;
    if (act == SBG_ACT_RELU) return y > 0.0f ? gain : 0.0f;
    if (act == SBG_ACT_LRELU) return y > 0.0f ? gain : gain * alpha;
    return gain;
}

template <class T>
__global__ void __launch_bounds__(256) fromrgb_fwd_kernel(FromArgs p)
{
    __shared__ float4_t px[256];
    const int lpp = p.Co >> 3, ppw = 64 / lpp;
    const int n = blockIdx.x / p.blocks_per_n, blk = blockIdx.x % p.blocks_per_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (lpp - 1), pl = lane / lpp;
    float w[MAX_CI][8], b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        b[j] = p.bias ? p.bias[cl * 8 + j] : 0.0f;
#pragma unroll
        for (int c = 0; c < MAX_CI; c++) w[c][j] = c < p.Ci ? p.w[(int64_t)(cl * 8 + j) * p.Ci + c] : 0.0f;
    }
    const float* ib = p.img + (int64_t)n * p.Ci * p.HW;
    T* yb = (T*)p.y + (int64_t)n * p.HW * p.Co;
    for (int64_t chunk = (int64_t)blk * 256; chunk < p.HW; chunk += (int64_t)p.blocks_per_n * 256) {
        {
            const int64_t pix = chunk + threadIdx.x;
            float4_t v = {0.f, 0.f, 0.f, 0.f};
            if (pix < p.HW) {
#pragma unroll
                for (int c = 0; c < MAX_CI; c++)
                    if (c < p.Ci) v[c] = ib[(int64_t)c * p.HW + pix];
            }
            __syncthreads();
            px[threadIdx.x] = v;
            __syncthreads();
        }
        for (int q = 0; q < 64; q += ppw) {
            const int loc = wave * 64 + q + pl;
            const int64_t pix = chunk + loc;
            if (pix >= p.HW) continue;
            const float4_t v = px[loc];
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float a = b[j];
#pragma unroll
                for (int c = 0; c < MAX_CI; c++) a += v[c] * w[c][j];
                a = act_fwd(a, p.act, p.alpha) * p.gain;
                if (p.clamp >= 0.0f) a = fminf(fmaxf(a, -p.clamp), p.clamp);
                o[j] = a;
            }
            Vec8<T>::st(yb + pix * p.Co + cl * 8, o);
        }
    }
}

// CI = image channels the lane arithmetic is unrolled for (3: RGB, the case that carries the bytes; MAX_CI: any);  DIMG = the image gradient
// is wanted (generator phases) -- without it the per-pixel products with w and the cross-lane sums disappear from the loop.
template <class T, int CI, bool DIMG>
__global__ void __launch_bounds__(256) fromrgb_bwd_kernel(FromArgs p)
{
    constexpr int RS = MAX_CI * 8 + 8 + 1;                 // floats per (wave, channel lane) row of the final reduction
    __shared__ float4_t px[256];
    extern __shared__ float red[];                        // [4 waves][lpp channel lanes][RS]: 10.5 KB at 128 channels, so LDS does not cap the
                                                          // occupancy of a streaming kernel (a [4][64][RS] array did: three workgroups per CU)
    const int lpp = p.Co >> 3, ppw = 64 / lpp;
    const int n = blockIdx.x / p.blocks_per_n, blk = blockIdx.x % p.blocks_per_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (lpp - 1), pl = lane / lpp;
    float w[MAX_CI][8];
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int c = 0; c < MAX_CI; c++) w[c][j] = c < p.Ci ? p.w[(int64_t)(cl * 8 + j) * p.Ci + c] : 0.0f;
    float dw[MAX_CI][8], db[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        db[j] = 0.0f;
#pragma unroll
        for (int c = 0; c < MAX_CI; c++) dw[c][j] = 0.0f;
    }
    // the saved output is 16-bit: a value clamped to +-clamp was stored as round(clamp), so the rail test uses the rounded bound
    float clamp_r = p.clamp;
    if (p.clamp >= 0.0f) { T tmp; Elem<T>::st(&tmp, p.clamp); clamp_r = Elem<T>::ld(&tmp); }
    const float* ib = p.img + (int64_t)n * p.Ci * p.HW;
    const T* dyb = (const T*)p.dy + (int64_t)n * p.HW * p.Co;
    const T* ysb = (const T*)p.ysaved + (int64_t)n * p.HW * p.Co;
    float* dib = DIMG ? p.dimg + (int64_t)n * p.Ci * p.HW : nullptr;
    // slope of clamp(act(x) * gain) on either side of zero (piecewise linear activations; bias_act.cu:141 conventions), zero on the rails
    const float gpos = p.gain, gneg = p.act == SBG_ACT_LRELU ? p.gain * p.alpha : (p.act == SBG_ACT_RELU ? 0.0f : p.gain);
    const float rail = p.clamp >= 0.0f ? clamp_r : __builtin_inff();
    for (int64_t chunk = (int64_t)blk * 256; chunk < p.HW; chunk += (int64_t)p.blocks_per_n * 256) {
        {
            const int64_t pix = chunk + threadIdx.x;
            float4_t v = {0.f, 0.f, 0.f, 0.f};
            if (pix < p.HW) {
#pragma unroll
                for (int c = 0; c < MAX_CI; c++)
                    if (c < p.Ci) v[c] = ib[(int64_t)c * p.HW + pix];
            }
            __syncthreads();
            px[threadIdx.x] = v;
            __syncthreads();
        }
        for (int q = 0; q < 64; q += ppw) {
            const int loc = wave * 64 + q + pl;
            const int64_t pix = chunk + loc;
            const bool ok = pix < p.HW;             // (uniform per pixel group; the DPP sums below need every lane of a row to take part)
            float g[8], yv[8];
            if (ok) {
                Vec8<T>::ld(dyb + pix * p.Co + cl * 8, g);
                Vec8<T>::ld(ysb + pix * p.Co + cl * 8, yv);
            }
            const float4_t v = px[loc & 255];
            float di[MAX_CI] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const bool live = ok && (fabsf(yv[j]) < rail);
                const float d1 = live ? g[j] * (yv[j] > 0.0f ? gpos : gneg) : 0.0f;
                db[j] += d1;
#pragma unroll
                for (int c = 0; c < CI; c++) { dw[c][j] += d1 * v[c]; if (DIMG) di[c] += d1 * w[c][j]; }
            }
            if (DIMG) {         // sum over the pixel's lanes (every lane ends with the total), then lane c mod lpp writes channel c
#pragma unroll
                for (int c = 0; c < CI; c++) {
                    float a = di[c];
                    for (int m = lpp >> 1; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
                    if (ok && cl == (c & (lpp - 1)) && c < p.Ci) dib[(int64_t)c * p.HW + pix] = a;
                }
            }
        }
    }
    // fixed-order reduction of dw / db: over the wave's pixel groups by a butterfly (lanes cl, cl + lpp, ...), then over the four waves
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int c = 0; c < MAX_CI; c++) {
            float a = c < CI ? dw[c][j] : 0.0f;
            for (int m = lpp; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
            if (pl == 0) red[(wave * lpp + cl) * RS + c * 8 + j] = a;
        }
        float a = db[j];
        for (int m = lpp; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
        if (pl == 0) red[(wave * lpp + cl) * RS + MAX_CI * 8 + j] = a;
    }
    __syncthreads();
    // partial layout: [N][blocks_per_n][Co * Ci + Co]
    float* out = p.partial + ((int64_t)n * p.blocks_per_n + blk) * ((int64_t)p.Co * p.Ci + p.Co);
    for (int idx = threadIdx.x; idx < p.Co * (p.Ci + 1); idx += 256) {
        const bool is_b = idx >= p.Co * p.Ci;
        const int co = is_b ? idx - p.Co * p.Ci : idx / p.Ci, c = is_b ? 0 : idx % p.Ci;
        const int gq = co >> 3, j = co & 7;
        const int slot = is_b ? MAX_CI * 8 + j : c * 8 + j;
        float s = 0.0f;
        for (int wv = 0; wv < 4; wv++) s += red[(wv * lpp + gq) * RS + slot];
        out[idx] = s;
    }
}

static int check_shape(int N, int Ci, int Co, int64_t HW)
{
    const int lpp = Co >> 3;
    SBG_CHECK(N >= 1 && HW >= 1 && Ci >= 1 && Ci <= MAX_CI, "fromrgb: bad sizes (Cin <= %d)", MAX_CI);
    SBG_CHECK(Co >= 8 && Co <= 512 && (Co & 7) == 0 && (lpp & (lpp - 1)) == 0, "fromrgb: Cout must be 8 * 2^k <= 512");
    SBG_CHECK((int64_t)N * Co * HW <= INT32_MAX, "fromrgb: tensors are limited to INT_MAX elements");
    return 0;
}

static int blocks_for(int N, int64_t HW)
{
    int64_t want = (HW + 255) / 256;                 // one 256-pixel chunk per workgroup at least
    int64_t cap = (256 * 8 + N - 1) / N;             // ~2048 workgroups in total
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

extern "C" int sbg_fromrgb_supported(int Ci, int Co, int act)
{
    const int lpp = Co >> 3;
    return (Ci >= 1 && Ci <= MAX_CI && Co >= 8 && Co <= 512 && (Co & 7) == 0 && (lpp & (lpp - 1)) == 0
            && (act == SBG_ACT_LINEAR || act == SBG_ACT_RELU || act == SBG_ACT_LRELU)) ? 1 : 0;
}

extern "C" int sbg_fromrgb_bwd_blocks(int N, int64_t HW) { return blocks_for(N, HW); }

extern "C" int sbg_fromrgb_fwd(const float* img, const float* w, const float* bias, void* y, int dtype, int N, int Ci, int Co, int64_t HW,
                               int act, float alpha, float gain, float clamp, sbg_stream_t stream_)
{
    SBG_CHECK(img && w && y, "fromrgb_fwd: null pointer");
    if (int rc = check_shape(N, Ci, Co, HW)) return rc;
    SBG_CHECK(sbg_fromrgb_supported(Ci, Co, act), "fromrgb_fwd: unsupported activation %d", act);
    SBG_CHECK(dtype == SBG_BF16 || dtype == SBG_F16, "fromrgb_fwd: 16-bit output only");
    FromArgs a = {};
    a.img = img; a.w = w; a.bias = bias; a.y = y; a.N = N; a.Ci = Ci; a.Co = Co; a.HW = HW; a.act = act; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    a.blocks_per_n = blocks_for(N, HW);
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_FROMRGB, 2.0 * N * Co * (double)Ci * HW, (double)N * HW * (2.0 * Co + 4.0 * Ci), {N, Ci, Co, (int)HW, 0, 0, 0});
    dim3 grid((unsigned)(N * a.blocks_per_n)), block(256);
    if (dtype == SBG_BF16) SBG_LAUNCH(fromrgb_fwd_kernel<bf16_s>, grid, block, 0, stream, a);
    else                   SBG_LAUNCH(fromrgb_fwd_kernel<f16_s>, grid, block, 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_fromrgb_bwd(const float* img, const float* w, const void* dy, const void* y, float* dimg, float* partial, int dtype,
                               int N, int Ci, int Co, int64_t HW, int act, float alpha, float gain, float clamp, sbg_stream_t stream_)
{
    SBG_CHECK(img && w && dy && y && partial, "fromrgb_bwd: null pointer");
    if (int rc = check_shape(N, Ci, Co, HW)) return rc;
    SBG_CHECK(sbg_fromrgb_supported(Ci, Co, act), "fromrgb_bwd: unsupported activation %d", act);
    SBG_CHECK(dtype == SBG_BF16 || dtype == SBG_F16, "fromrgb_bwd: 16-bit activations only");
    FromArgs a = {};
    a.img = img; a.w = w; a.dy = dy; a.ysaved = y; a.dimg = dimg; a.partial = partial;
    a.N = N; a.Ci = Ci; a.Co = Co; a.HW = HW; a.act = act; a.alpha = alpha; a.gain = gain; a.clamp = clamp;
    a.blocks_per_n = blocks_for(N, HW);
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_FROMRGB, 4.0 * N * Co * (double)Ci * HW, (double)N * HW * (4.0 * Co + 8.0 * Ci), {N, Ci, Co, (int)HW, 1, 0, 0});
    dim3 grid((unsigned)(N * a.blocks_per_n)), block(256);
    const int lds = 4 * (Co >> 3) * (MAX_CI * 8 + 8 + 1) * (int)sizeof(float);      // <= 42 KB (Co = 512)
#define SBG_FROMRGB_BWD(T_) do { \
        if (Ci == 3) { if (dimg) SBG_LAUNCH((fromrgb_bwd_kernel<T_, 3, true>), grid, block, lds, stream, a); else SBG_LAUNCH((fromrgb_bwd_kernel<T_, 3, false>), grid, block, lds, stream, a); } \
        else { if (dimg) SBG_LAUNCH((fromrgb_bwd_kernel<T_, MAX_CI, true>), grid, block, lds, stream, a); else SBG_LAUNCH((fromrgb_bwd_kernel<T_, MAX_CI, false>), grid, block, lds, stream, a); } } while (0)
    if (dtype == SBG_BF16) SBG_FROMRGB_BWD(bf16_s); else SBG_FROMRGB_BWD(f16_s);
#undef SBG_FROMRGB_BWD
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
