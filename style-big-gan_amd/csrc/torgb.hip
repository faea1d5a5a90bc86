// torgb.hip -- the ToRGB layer as streaming kernels.
//
// ToRGBLayer.forward (train_parts/generators.py:344-348) = modulated 1x1 convolution without demodulation to 3 channels + linear
// bias_act with clamp:  y[n, o, p] = clamp(sum_c x[n, c, p] * (w[o, c] * s[n, c]) + b[o]).  With O = 3 this is not matrix-core work:
// a GEMM tile is >= 97 % padding, and the implicit-GEMM path additionally needs the modulated copy x * s.  Here the per-sample
// weights wmod[n, o, c] = w[o, c] * s[n, c] (a [N, O, C] tensor, formed by the host so autograd splits its gradient into dw and ds)
// are applied while x streams through once:
//   forward : read x (channel-minor 16-bit), write y (planar fp32)                        -- HBM-bound, numel(x) * 2 B
//   backward: read x and dy, write dx = sum_o d1[o] * wmod[n, o, c] and the partial sums of dwmod[n, o, c] = sum_p d1[o, p] * x[c, p],
//             dbias[o] = sum d1[o, p]  (d1 = dy masked by the clamp, from the saved y)     -- HBM-bound, numel(x) * 4 B
// Lane mapping: a pixel's C channels are contiguous (C * 2 bytes); C / 8 lanes share a pixel with 8 channels (16 B) each, so a
// wavefront covers 64 / (C / 8) pixels per load.  C / 8 must be a power of two <= 64 (C = 8 ... 512), O <= 4.
#include "sbg_common.h"

namespace {

constexpr int MAX_O = 4;

struct RgbArgs {
    const void* x; const float* wmod; const float* bias; const float* dy; const float* ysaved;
    float* y; void* dx; float* partial;
    int N, C, O; int64_t HW;
    float clamp;
    int blocks_per_n;
};

// NO = outputs the instantiation carries (3: the RGB case without the padding row -- a quarter of the kernel's vector instructions; it is
// issue-bound: ~100 vector instructions per KiB read)
template <class T, int NO>
__global__ void __launch_bounds__(256) torgb_fwd_kernel(RgbArgs p)
{
    const int lpp = p.C >> 3;                       // lanes per pixel
    const int ppw = 64 / lpp;                       // pixels per wavefront load
    const int n = blockIdx.x / p.blocks_per_n, blk = blockIdx.x % p.blocks_per_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (lpp - 1), pl = lane / lpp;
    float w[NO][8];
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
        for (int j = 0; j < 8; j++) w[o][j] = o < p.O ? p.wmod[((int64_t)n * p.O + o) * p.C + cl * 8 + j] : 0.0f;
    const T* xb = (const T*)p.x + (int64_t)n * p.HW * p.C;
    float* yb = p.y + (int64_t)n * p.O * p.HW;
    // (issuing 2 or 4 pixel groups' loads per iteration measured slower: 268 / 288 us vs 199 us at [32,128,256,256])
    const int64_t stride = (int64_t)p.blocks_per_n * 4 * ppw;
    for (int64_t pix0 = ((int64_t)blk * 4 + wave) * ppw; pix0 < p.HW; pix0 += stride) {
        const int64_t pix = pix0 + pl;
        float acc[NO] = {};
        if (pix < p.HW) {
            float v[8];
            Vec8<T>::ld(xb + pix * p.C + cl * 8, v);
#pragma unroll
            for (int o = 0; o < NO; o++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[o] += v[j] * w[o][j];
        }
        if (lpp >= 16) {            // sum over the 16 lanes of a row with DPP rotates (no LDS crossbar), then across rows if a pixel spans several
#pragma unroll
            for (int o = 0; o < NO; o++) {
                float a = acc[o];
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xf, 0xf, false));    // row_ror:8
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x124, 0xf, 0xf, false));    // row_ror:4
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x122, 0xf, 0xf, false));    // row_ror:2
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x121, 0xf, 0xf, false));    // row_ror:1
                acc[o] = a;
            }
            for (int m = lpp >> 1; m >= 16; m >>= 1)
#pragma unroll
                for (int o = 0; o < NO; o++) acc[o] += __shfl_xor(acc[o], m, 64);
        } else {
            for (int m = lpp >> 1; m >= 1; m >>= 1)
#pragma unroll
                for (int o = 0; o < NO; o++) acc[o] += __shfl_xor(acc[o], m, 64);
        }
        if (cl == 0 && pix < p.HW) {
#pragma unroll
            for (int o = 0; o < NO; o++)
                if (o < p.O) {
                    float r = acc[o] + (p.bias ? p.bias[o] : 0.0f);
                    if (p.clamp >= 0.0f) r = fminf(fmaxf(r, -p.clamp), p.clamp);
                    yb[(int64_t)o * p.HW + pix] = r;
                }
        }
    }
}

// Forward on the matrix cores after all -- not for the arithmetic (3 useful columns of 16) but because the streaming form above is ISSUE-bound:
// ~100 vector instructions per KiB read (8 conversions, 24-32 FMAs, a 16-lane DPP reduction per output) hold it at 2.2-3.1 TB/s whatever is done to
// the loads (two in flight per wave: slower; a contiguous run per workgroup: the same; one output row less: +2 %).  Here a wave takes 16 pixels:
// lane (pixel fr, k-group fg) loads its 16 B straight into the A operand of v_mfma_f32_16x16x32 (no conversion, no reduction), the per-sample
// weights sit in registers as the B operand -- split into three 16-bit parts (head, remainder, remainder of that), three MFMAs per 32 channels, so
// the product keeps the fp32 weights' accuracy (two parts, ~2^-17, miss the 1e-5 the test holds the output to) -- and lanes fr < O hold the results
// of pixels 4 fg .. 4 fg + 3 as one 16-B store.  C = 32 NKB = 128 / 256 (at 512 the 192 weight registers leave no room: the streaming kernel stays,
// its launches are the small 4x4 .. 64x64 layers), HW a multiple of 16.
// Measured ([64, C, R, R], same box): 128 @256^2 369 -> 253 us (3.05 -> 4.44 TB/s), 256 @128^2 207 -> 143 us (2.66 -> 3.83).
template <class T> struct RgbMfma;
template <> struct RgbMfma<bf16_s> {
    static __device__ __forceinline__ unsigned short cvt(float v) { return f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float back(unsigned short b) { return bf16_bits_to_f32(b); }
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct RgbMfma<f16_s> {
    static __device__ __forceinline__ unsigned short cvt(float v) { return f32_to_f16_bits(v); }
    static __device__ __forceinline__ float back(unsigned short b) { return f16_bits_to_f32(b); }
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

template <class T, int NKB>
__global__ void __launch_bounds__(256) torgb_fwd_mfma_kernel(RgbArgs p)
{
    const int n = blockIdx.x / p.blocks_per_n, blk = blockIdx.x % p.blocks_per_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    short8_t bh[NKB], bm[NKB], bl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; kb++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float w = fr < p.O ? p.wmod[((int64_t)n * p.O + fr) * p.C + kb * 32 + fg * 8 + j] : 0.0f;
            const unsigned short h = RgbMfma<T>::cvt(w);
            const float r1 = w - RgbMfma<T>::back(h);
            const unsigned short m = RgbMfma<T>::cvt(r1);
            bh[kb][j] = (short)h; bm[kb][j] = (short)m;
            bl[kb][j] = (short)RgbMfma<T>::cvt(r1 - RgbMfma<T>::back(m));
        }
    const float bias = (p.bias && fr < p.O) ? p.bias[fr] : 0.0f;
    const T* xb = (const T*)p.x + (int64_t)n * p.HW * p.C;
    float* yb = p.y + (int64_t)n * p.O * p.HW;
    const int64_t groups = p.HW >> 4;
    const int64_t per = (groups + p.blocks_per_n - 1) / p.blocks_per_n;      // a workgroup streams one contiguous run of 16-pixel groups
    const int64_t gend = (blk + 1) * per < groups ? (blk + 1) * per : groups;
    for (int64_t g = blk * per + wave; g < gend; g += 4) {
        const T* px = xb + (g * 16 + fr) * p.C + fg * 8;
        short8_t a[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) a[kb] = *reinterpret_cast<const short8_t*>(px + kb * 32);
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) {
            acc = RgbMfma<T>::run(a[kb], bl[kb], acc);      // smallest parts first
            acc = RgbMfma<T>::run(a[kb], bm[kb], acc);
            acc = RgbMfma<T>::run(a[kb], bh[kb], acc);
        }
        if (fr < p.O) {
            float4_t r;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v = acc[e] + bias;
                if (p.clamp >= 0.0f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                r[e] = v;
            }
            *reinterpret_cast<float4_t*>(yb + (int64_t)fr * p.HW + g * 16 + 4 * fg) = r;
        }
    }
}

template <class T>
__global__ void __launch_bounds__(256) torgb_bwd_kernel(RgbArgs p)
{
    __shared__ float red[4][64][MAX_O * 8 + 1];     // per-wave copies of the lane accumulators for the cross-lane / cross-wave sum
    __shared__ float redb[4][64][MAX_O];
    const int lpp = p.C >> 3, ppw = 64 / lpp;
    const int n = blockIdx.x / p.blocks_per_n, blk = blockIdx.x % p.blocks_per_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & (lpp - 1), pl = lane / lpp;
    float w[MAX_O][8];
#pragma unroll
    for (int o = 0; o < MAX_O; o++)
#pragma unroll
        for (int j = 0; j < 8; j++) w[o][j] = o < p.O ? p.wmod[((int64_t)n * p.O + o) * p.C + cl * 8 + j] : 0.0f;
    const T* xb = (const T*)p.x + (int64_t)n * p.HW * p.C;
    T* dxb = (T*)p.dx + (int64_t)n * p.HW * p.C;
    const float* dyb = p.dy + (int64_t)n * p.O * p.HW;
    const float* ysb = p.ysaved + (int64_t)n * p.O * p.HW;
    float dw[MAX_O][8];
    float db[MAX_O] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < MAX_O; o++)
#pragma unroll
        for (int j = 0; j < 8; j++) dw[o][j] = 0.0f;
    // chunks of 256 pixels: every thread turns one pixel's (dy, y) into the masked gradient d1 (coalesced planar loads) and parks it in
    // LDS; then each wavefront walks its 64 pixels, ppw at a time, with one LDS read + one 16-B load + one 16-B store per lane
    __shared__ float4_t d1s[256];
    for (int64_t chunk = (int64_t)blk * 256; chunk < p.HW; chunk += (int64_t)p.blocks_per_n * 256) {
        {
            const int64_t pix = chunk + threadIdx.x;
            float4_t d = {0.f, 0.f, 0.f, 0.f};
            if (pix < p.HW) {
#pragma unroll
                for (int o = 0; o < MAX_O; o++)
                    if (o < p.O) {
                        const float g = dyb[(int64_t)o * p.HW + pix];
                        const float yv = ysb[(int64_t)o * p.HW + pix];
                        d[o] = (p.clamp >= 0.0f && !(fabsf(yv) < p.clamp)) ? 0.0f : g;      // clamp gradient: zero where the output sits on the rail (bias_act.cu:141)
                    }
            }
            __syncthreads();            // the previous chunk's readers are done
            d1s[threadIdx.x] = d;
            __syncthreads();
        }
        for (int q = 0; q < 64; q += 2 * ppw) {         // two pixel groups per iteration: both loads in flight before the FMAs
            float v[2][8];
            float4_t dd[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int loc = wave * 64 + q + u * ppw + pl;
                ok[u] = (q + u * ppw < 64) && (chunk + loc < p.HW);
                if (ok[u]) { Vec8<T>::ld(xb + (chunk + loc) * p.C + cl * 8, v[u]); dd[u] = d1s[loc]; }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!ok[u]) continue;
                const int64_t pix = chunk + wave * 64 + q + u * ppw + pl;
                const float d1[MAX_O] = {dd[u][0], dd[u][1], dd[u][2], dd[u][3]};
                float dxv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float a = 0.0f;
#pragma unroll
                    for (int o = 0; o < MAX_O; o++) { a += d1[o] * w[o][j]; dw[o][j] += d1[o] * v[u][j]; }
                    dxv[j] = a;
                }
                if (p.dx) Vec8<T>::st(dxb + pix * p.C + cl * 8, dxv);
                if (cl == 0) {
#pragma unroll
                    for (int o = 0; o < MAX_O; o++) db[o] += d1[o];
                }
            }
        }
    }
    // block reduction in a fixed order: lanes with the same channel group (cl) over pl, then the four waves
#pragma unroll
    for (int o = 0; o < MAX_O; o++) {
#pragma unroll
        for (int j = 0; j < 8; j++) red[wave][lane][o * 8 + j] = dw[o][j];
        redb[wave][lane][o] = db[o];
    }
    __syncthreads();
    // partial layout: [N][blocks_per_n][O * C + O]
    float* out = p.partial + ((int64_t)n * p.blocks_per_n + blk) * ((int64_t)p.O * p.C + p.O);
    for (int idx = threadIdx.x; idx < p.O * p.C; idx += 256) {
        const int o = idx / p.C, c = idx % p.C;
        const int g = c >> 3, j = c & 7;
        float s = 0.0f;
        for (int wv = 0; wv < 4; wv++)
            for (int q = 0; q < ppw; q++) s += red[wv][q * lpp + g][o * 8 + j];
        out[idx] = s;
    }
    if (threadIdx.x < p.O) {
        float s = 0.0f;
        for (int wv = 0; wv < 4; wv++)
            for (int q = 0; q < ppw; q++) s += redb[wv][q * lpp][threadIdx.x];
        out[(int64_t)p.O * p.C + threadIdx.x] = s;
    }
}

static int check_shape(int N, int C, int O, int64_t HW)
{
    const int lpp = C >> 3;
    SBG_CHECK(N >= 1 && HW >= 1 && O >= 1 && O <= MAX_O, "torgb: bad sizes (O <= %d)", MAX_O);
    SBG_CHECK(C >= 8 && C <= 512 && (C & 7) == 0 && (lpp & (lpp - 1)) == 0, "torgb: C must be 8 * 2^k <= 512");
    SBG_CHECK((int64_t)N * C * HW <= INT32_MAX, "torgb: tensors are limited to INT_MAX elements");
    return 0;
}

static int blocks_for(int N, int C, int64_t HW)
{
    const int ppw = 64 / (C >> 3);
    int64_t want = (HW + 4 * ppw * 8 - 1) / (4 * ppw * 8);      // >= 8 pixel groups per wavefront
    int64_t cap = (256 * 8 + N - 1) / N;                        // ~2048 workgroups in total
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace

extern "C" int sbg_torgb_supported(int C, int O)
{
    const int lpp = C >> 3;
    return (O >= 1 && O <= MAX_O && C >= 8 && C <= 512 && (C & 7) == 0 && (lpp & (lpp - 1)) == 0) ? 1 : 0;
}

extern "C" int sbg_torgb_bwd_blocks(int N, int C, int64_t HW) { return blocks_for(N, C, HW); }

extern "C" int sbg_torgb_fwd(const void* x, const float* wmod, const float* bias, float* y, int dtype, int N, int C, int O, int64_t HW,
                             float clamp, sbg_stream_t stream_)
{
    SBG_CHECK(x && wmod && y, "torgb_fwd: null pointer");
    if (int rc = check_shape(N, C, O, HW)) return rc;
    SBG_CHECK(dtype == SBG_BF16 || dtype == SBG_F16, "torgb_fwd: 16-bit activations only");
    RgbArgs a = {};
    a.x = x; a.wmod = wmod; a.bias = bias; a.y = y; a.N = N; a.C = C; a.O = O; a.HW = HW; a.clamp = clamp;
    a.blocks_per_n = blocks_for(N, C, HW);
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_TORGB, 2.0 * N * O * (double)C * HW, (double)N * HW * (2.0 * C + 4.0 * O), {N, C, O, (int)HW, 0, 0, 0});
    dim3 grid((unsigned)(N * a.blocks_per_n)), block(256);
    static const char* emf = sbg_env("SBG_TORGB_MFMA");          // experiment switch: 0 = the streaming kernel for every shape
    if ((HW & 15) == 0 && (C == 128 || C == 256) && sbg_aligned16(x) && sbg_aligned16(y) && !(emf && atoi(emf) == 0)) {
        if (dtype == SBG_BF16) {
            if (C == 128) SBG_LAUNCH((torgb_fwd_mfma_kernel<bf16_s, 4>), grid, block, 0, stream, a);
            else SBG_LAUNCH((torgb_fwd_mfma_kernel<bf16_s, 8>), grid, block, 0, stream, a);
        } else {
            if (C == 128) SBG_LAUNCH((torgb_fwd_mfma_kernel<f16_s, 4>), grid, block, 0, stream, a);
            else SBG_LAUNCH((torgb_fwd_mfma_kernel<f16_s, 8>), grid, block, 0, stream, a);
        }
        SBG_HIP_LAUNCH_CHECK();
        return 0;
    }
    if (O <= 3) {
        if (dtype == SBG_BF16) SBG_LAUNCH((torgb_fwd_kernel<bf16_s, 3>), grid, block, 0, stream, a);
        else                   SBG_LAUNCH((torgb_fwd_kernel<f16_s, 3>), grid, block, 0, stream, a);
    } else {
        if (dtype == SBG_BF16) SBG_LAUNCH((torgb_fwd_kernel<bf16_s, MAX_O>), grid, block, 0, stream, a);
        else                   SBG_LAUNCH((torgb_fwd_kernel<f16_s, MAX_O>), grid, block, 0, stream, a);
    }
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_torgb_bwd(const void* x, const float* wmod, const float* dy, const float* y, void* dx, float* partial, int dtype,
                             int N, int C, int O, int64_t HW, float clamp, sbg_stream_t stream_)
{
    SBG_CHECK(x && wmod && dy && y && partial, "torgb_bwd: null pointer");
    if (int rc = check_shape(N, C, O, HW)) return rc;
    SBG_CHECK(dtype == SBG_BF16 || dtype == SBG_F16, "torgb_bwd: 16-bit activations only");
    RgbArgs a = {};
    a.x = x; a.wmod = wmod; a.dy = dy; a.ysaved = y; a.dx = dx; a.partial = partial; a.N = N; a.C = C; a.O = O; a.HW = HW; a.clamp = clamp;
    a.blocks_per_n = blocks_for(N, C, HW);
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_TORGB, 4.0 * N * O * (double)C * HW, (double)N * HW * (4.0 * C + 8.0 * O), {N, C, O, (int)HW, 1, 0, 0});
    dim3 grid((unsigned)(N * a.blocks_per_n)), block(256);
    if (dtype == SBG_BF16) SBG_LAUNCH(torgb_bwd_kernel<bf16_s>, grid, block, 0, stream, a);
    else                   SBG_LAUNCH(torgb_bwd_kernel<f16_s>, grid, block, 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
