// conv3x3_halo.hip -- stride-1 KxK (K <= 3) convolution on the matrix cores with halo-staged activations.
//
// The dominant convolutions of the StyleGAN2 step (every 3x3 stride-1 layer of G and D, and their data gradients) spend
// their time, in the generic gather kernel (conv_igemm.hip), moving the same activation bytes L2 -> LDS nine times (once
// per tap).  Here a workgroup owns a TH x TW patch of output pixels of one image; per 32-channel slice it stages the
// (TH+2) x (TW+2) input halo ONCE and all nine taps read shifted windows of it straight out of LDS, so the per-tap global
// traffic is only the 8 KB weight tile:  L2 -> LDS bytes per MFMA clock drop from ~64 B/clk/CU (128x128x32 gather tile)
// to ~21 B/clk/CU, under the ~56 B/clk/CU the L2 can feed every CU at once.
//
// Same contract as sbg_conv2d_igemm restricted to stride 1, |dy|,|dx| <= 1, one image per tile (OH % TH == 0,
// OW % TW == 0); optional fused prologue / epilogue (see sbg_conv3x3_params in include/sbg_hip.h):
//   x' = x * iscale[n, ci]                                   (modulation, applied while staging the halo)
//   y  = act((acc * oscale[n, co] + noise[n, pixel] + bias[co])) * gain, clamped        (demod + noise + bias_act)
//
// Workgroup = 256 lanes = 2 (cout) x 2 (pixel) waves; tile 128 cout x 256 pixels; wave tile 64 x 128 (4 x 8 MFMA tiles,
// 128 accumulator VGPRs).  LDS: halo [2 buffers][4 k-groups][PLANE cells of 16 B] (PLANE = halo pixels rounded up to 16, so
// every 16-pixel fragment read hits 16 distinct bank slots), weights [2][4][128] cells.  K-step = (channel slice, tap);
// the next step's weight tile and one sixth of the next slice's halo are issued before the MFMAs of the current step and
// written to the other buffers after them (issue early / write late), one barrier per step.
#include "sbg_common.h"

namespace {

struct bf16_mfma {}; struct f16_mfma {};
template <class MF> struct Mfma;
template <> struct Mfma<bf16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned short v) { return bf16_bits_to_f32(v); }
    static __device__ __forceinline__ unsigned short down(float v) { return f32_to_bf16_bits(v); }
};
template <> struct Mfma<f16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned short v) { return f16_bits_to_f32(v); }
    static __device__ __forceinline__ unsigned short down(float v) { return f32_to_f16_bits(v); }
};

struct HaloArgs {
    const unsigned short* x; const unsigned short* w; void* y;
    const float* iscale; const float* oscale; const float* noise; const float* bias;
    int ydtype;
    int N, H, W, Cin, Cout;
    int64_t xs_n, xs_h, xs_w, ys_n, ys_h, ys_w, ws_slab, ws_co, noise_sn;
    int ntaps;
    int tap_dy[9], tap_dx[9], tap_slab[9];
    int act; float alpha, gain, clamp;
    int accumulate;
    int tiles_x, tiles_y, ctiles;
};

template <class MF, int TH, int TW>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(HaloArgs p)
{
    constexpr int BC = 128, BP = TH * TW;
    static_assert(BP == 256 && (TW % 16) == 0, "256-pixel tile with 16-pixel row segments");
    constexpr int PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
    constexpr int PLANE = (NPIX + 15) / 16 * 16;            // cells per k-group plane
    constexpr int HALO_BYTES = 4 * PLANE * 16, WT_BYTES = 4 * BC * 16;
    constexpr int HL = (NPIX * 4 + 255) / 256;              // halo staging loads per lane per slice
    constexpr int SEG = TW / 16;                            // 16-pixel segments per tile row

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;                               // 2 halo buffers
    unsigned char* sW = smem + 2 * HALO_BYTES;              // 2 weight buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    {   // XCD-aware order: workgroups sharing an XCD (b mod 8) walk neighbouring tiles -> halo rows re-used from that L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ct = bid % p.ctiles; int t = bid / p.ctiles;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int c0 = ct * BC, y0 = ty * TH, x0 = tx * TW;

    // ---- staging coordinates --------------------------------------------------------------------------------------
    const int sg = (tid >> 3) & 3;                                   // k-group of this lane (8 consecutive lanes share it)
    const int srow = (tid & 7) | ((tid >> 5) << 3);                  // 0..63: row within a 64-row staging pass
    int64_t h_off[HL]; bool h_ok[HL]; int h_cell[HL];
#pragma unroll
    for (int i = 0; i < HL; i++) {
        const int pp = srow + 64 * i;                                // halo pixel index
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        h_ok[i] = pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        h_off[i] = (int64_t)n * p.xs_n + (int64_t)iy * p.xs_h + (int64_t)ix * p.xs_w;
        h_cell[i] = (pp < NPIX) ? (sg * PLANE + pp) : -1;
    }
    int64_t a_off[2]; bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int co = c0 + srow + 64 * i;
        a_ok[i] = co < p.Cout;
        a_off[i] = (int64_t)(a_ok[i] ? co : 0) * p.ws_co;
    }
    const int kchunks = (p.Cin + 31) >> 5;
    const int nsteps = kchunks * p.ntaps;

    auto load_halo = [&](int chunk, int i) -> short8_t {
        const int ck = chunk * 32 + sg * 8;
        short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h_ok[i] && ck < p.Cin) {
            v = *reinterpret_cast<const short8_t*>(p.x + h_off[i] + ck);
            if (p.iscale) {
                const float* sc = p.iscale + (int64_t)n * p.Cin + ck;
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (short)Mfma<MF>::down(Mfma<MF>::up((unsigned short)v[e]) * sc[e]);
            }
        }
        return v;
    };
    auto load_w = [&](int step, short8_t (&r)[2]) {
        const int chunk = step / p.ntaps, tp = step - chunk * p.ntaps;
        const int ck = chunk * 32 + sg * 8;
        const unsigned short* ws = p.w + (int64_t)p.tap_slab[tp] * p.ws_slab + ck;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a_ok[i] && ck < p.Cin) v = *reinterpret_cast<const short8_t*>(ws + a_off[i]);
            r[i] = v;
        }
    };
    auto store_w = [&](int buf, const short8_t (&r)[2]) {
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<short8_t*>(sW + buf * WT_BYTES + (sg * BC + srow + 64 * i) * 16) = r[i];
    };

    // ---- MFMA coordinates -------------------------------------------------------------------------------------------
    const int wc = (wave >> 1) * 64;                 // cout offset of this wave
    const int wseg0 = (wave & 1) * 8;                // first of the wave's eight 16-pixel segments (tile has 16 segments)
    const int fr = lane & 15, fg = lane >> 4;
    float4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    int seg_cell[8];                                 // halo cell of (segment j, lane fr) for tap (0, 0)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int s = wseg0 + j, r = s / SEG, cseg = (s - r * SEG) * 16;
        seg_cell[j] = (r + 1) * PW + cseg + fr + 1;
    }

    // ---- prologue: slice 0 halo, step 0 weights --------------------------------------------------------------------
    {
#pragma unroll
        for (int i = 0; i < HL; i++) {
            short8_t v = load_halo(0, i);
            if (h_cell[i] >= 0) *reinterpret_cast<short8_t*>(sH + h_cell[i] * 16) = v;
        }
        short8_t rw[2];
        load_w(0, rw);
        store_w(0, rw);
    }
    __syncthreads();

    for (int s = 0; s < nsteps; s++) {
        const int chunk = s / p.ntaps, tp = s - chunk * p.ntaps;
        const int hbuf = chunk & 1, wbuf = s & 1;
        // issue early: next step's weights, and piece `tp` of the next slice's halo
        short8_t rw[2], rh = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool more = (s + 1 < nsteps);
        if (more) load_w(s + 1, rw);
        const bool halo_piece = (tp < HL) && (chunk + 1 < kchunks);
        if (halo_piece) {
#pragma unroll
            for (int i = 0; i < HL; i++) if (i == tp) rh = load_halo(chunk + 1, i);
        }
        // taps beyond HL-1 never carry a halo piece; with ntaps < HL (1x1 ..) the remaining pieces ride on the last tap
        // compute
        const unsigned char* hb = sH + hbuf * HALO_BYTES;
        const unsigned char* wb = sW + wbuf * WT_BYTES;
        const int shift = p.tap_dy[tp] * PW + p.tap_dx[tp];
        short8_t fa[4], fb[8];
#pragma unroll
        for (int i = 0; i < 4; i++) fa[i] = *reinterpret_cast<const short8_t*>(wb + (fg * BC + wc + 16 * i + fr) * 16);
#pragma unroll
        for (int j = 0; j < 8; j++) fb[j] = *reinterpret_cast<const short8_t*>(hb + (fg * PLANE + seg_cell[j] + shift) * 16);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = Mfma<MF>::run(fa[i], fb[j], acc[i][j]);
        // write late
        if (more) store_w(wbuf ^ 1, rw);
        if (halo_piece) {
#pragma unroll
            for (int i = 0; i < HL; i++) if (i == tp && h_cell[i] >= 0) *reinterpret_cast<short8_t*>(sH + (hbuf ^ 1) * HALO_BYTES + h_cell[i] * 16) = rh;
        }
        if (tp == p.ntaps - 1 && p.ntaps < HL && chunk + 1 < kchunks) {     // few taps: stage the rest of the next halo now
            for (int i = p.ntaps; i < HL; i++) {
                short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < HL; k++) if (k == i) v = load_halo(chunk + 1, k);
#pragma unroll
                for (int k = 0; k < HL; k++) if (k == i && h_cell[k] >= 0) *reinterpret_cast<short8_t*>(sH + (hbuf ^ 1) * HALO_BYTES + h_cell[k] * 16) = v;
            }
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int s = wseg0 + j, r = s / SEG, cseg = (s - r * SEG) * 16;
        const int oy = y0 + r, ox = x0 + cseg + fr;
        const int64_t yoff = (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        const float nz = p.noise ? p.noise[(int64_t)n * p.noise_sn + (int64_t)oy * p.W + ox] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int co = c0 + wc + 16 * i + 4 * fg;
            if (co >= p.Cout) continue;
            float4_t v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (co + e >= p.Cout) continue;
                float u = v[e];
                if (p.oscale) u *= p.oscale[(int64_t)n * p.Cout + co + e];
                u += nz;
                if (p.bias) u += p.bias[co + e];
                if (p.act == SBG_ACT_LRELU) u = (u > 0.f) ? u : u * p.alpha;
                else if (p.act == SBG_ACT_RELU) u = (u > 0.f) ? u : 0.f;
                u *= p.gain;
                if (p.clamp >= 0.f) u = (u > -p.clamp && u < p.clamp) ? u : (u >= 0.f ? p.clamp : -p.clamp);
                v[e] = u;
            }
            const bool full = (co + 4 <= p.Cout);
            if (p.ydtype == SBG_F32) {
                float* dst = (float*)p.y + yoff + co;
                if (full && ((((uintptr_t)dst) & 15) == 0)) {
                    float4_t o = v;
                    if (p.accumulate) o += *reinterpret_cast<float4_t*>(dst);
                    *reinterpret_cast<float4_t*>(dst) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (co + e < p.Cout) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
                }
            } else {
                unsigned short* dst = (unsigned short*)p.y + yoff + co;
                unsigned short h[4];
#pragma unroll
                for (int e = 0; e < 4; e++) h[e] = (p.ydtype == SBG_BF16) ? f32_to_bf16_bits(v[e]) : f32_to_f16_bits(v[e]);
                if (full && ((((uintptr_t)dst) & 7) == 0)) {
                    short4_t o = {(short)h[0], (short)h[1], (short)h[2], (short)h[3]};
                    *reinterpret_cast<short4_t*>(dst) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (co + e < p.Cout) dst[e] = h[e];
                }
            }
        }
    }
}

template <class MF, int TH, int TW>
static int launch_halo(HaloArgs& a, hipStream_t stream)
{
    constexpr int NPIX = (TH + 2) * (TW + 2), PLANE = (NPIX + 15) / 16 * 16;
    constexpr int lds = 2 * 4 * PLANE * 16 + 2 * 4 * 128 * 16;
    a.tiles_x = a.W / TW; a.tiles_y = a.H / TH; a.ctiles = (a.Cout + 127) / 128;
    const int64_t nblk = (int64_t)a.N * a.tiles_x * a.tiles_y * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv3x3: grid too large");
    auto kern = conv3x3_halo_kernel<MF, TH, TW>;
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    const double P = (double)a.N * a.H * a.W;
    SbgProfScope prof(stream, SBG_K_CONV3X3_HALO, 2.0 * P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * P * a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * P * a.Cout * (a.accumulate ? 2 : 1),
                      {(int)P, a.Cout, a.Cin, a.ntaps, 1, a.H, TH * 1000 + TW});
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

extern "C" int sbg_conv3x3_supported(const sbg_conv3x3_params* q)
{
    if (!q) return 0;
    if (q->H % 8 != 0 && q->H % 16 != 0) return 0;
    const bool t32 = (q->W % 32 == 0) && (q->H % 8 == 0);
    const bool t16 = (q->W % 16 == 0) && (q->H % 16 == 0);
    if (!t32 && !t16) return 0;
    if (q->ntaps < 1 || q->ntaps > 9 || (q->Cin % 8) != 0) return 0;
    for (int t = 0; t < q->ntaps; t++)
        if (q->tap_dy[t] < -1 || q->tap_dy[t] > 1 || q->tap_dx[t] < -1 || q->tap_dx[t] > 1) return 0;
    return 1;
}

extern "C" int sbg_conv3x3(const sbg_conv3x3_params* q, sbg_stream_t stream)
{
    SBG_CHECK(q && q->x && q->w && q->y, "conv3x3: null pointer");
    SBG_CHECK(sbg_conv3x3_supported(q), "conv3x3: shape not supported by the halo kernel (use sbg_conv2d_igemm)");
    SBG_CHECK(q->xdtype == SBG_BF16 || q->xdtype == SBG_F16, "conv3x3: x/w must be bf16 or f16");
    SBG_CHECK(q->ydtype == SBG_F32 || q->ydtype == SBG_BF16 || q->ydtype == SBG_F16, "conv3x3: bad output dtype");
    SBG_CHECK(!q->accumulate || q->ydtype == SBG_F32, "conv3x3: accumulate needs an fp32 output");
    SBG_CHECK(sbg_aligned16(q->x) && sbg_aligned16(q->w), "conv3x3: x and w must be 16-byte aligned");
    SBG_CHECK((q->xs_n % 8) == 0 && (q->xs_h % 8) == 0 && (q->xs_w % 8) == 0 && (q->ws_slab % 8) == 0 && (q->ws_co % 8) == 0,
              "conv3x3: pixel / row strides must be multiples of 8 elements");
    SBG_CHECK(q->act == SBG_ACT_LINEAR || q->act == SBG_ACT_LRELU || q->act == SBG_ACT_RELU, "conv3x3: fused activation must be linear, relu or lrelu");
    SBG_CHECK(!q->iscale || sbg_aligned16(q->iscale), "conv3x3: iscale must be 16-byte aligned");
    if (q->N == 0) return SBG_OK;
    HaloArgs a;
    a.x = (const unsigned short*)q->x; a.w = (const unsigned short*)q->w; a.y = q->y;
    a.iscale = q->iscale; a.oscale = q->oscale; a.noise = q->noise; a.bias = q->bias;
    a.ydtype = q->ydtype;
    a.N = q->N; a.H = q->H; a.W = q->W; a.Cin = q->Cin; a.Cout = q->Cout;
    a.xs_n = q->xs_n; a.xs_h = q->xs_h; a.xs_w = q->xs_w; a.ys_n = q->ys_n; a.ys_h = q->ys_h; a.ys_w = q->ys_w;
    a.ws_slab = q->ws_slab; a.ws_co = q->ws_co; a.noise_sn = q->noise_stride_n;
    a.ntaps = q->ntaps;
    for (int t = 0; t < 9; t++) { a.tap_dy[t] = q->tap_dy[t]; a.tap_dx[t] = q->tap_dx[t]; a.tap_slab[t] = q->tap_slab[t]; }
    a.act = q->act; a.alpha = q->alpha; a.gain = q->gain; a.clamp = q->clamp;
    a.accumulate = q->accumulate;
    hipStream_t s = (hipStream_t)stream;
    const bool t32 = (q->W % 32 == 0) && (q->H % 8 == 0);
    if (q->xdtype == SBG_BF16) return t32 ? launch_halo<bf16_mfma, 8, 32>(a, s) : launch_halo<bf16_mfma, 16, 16>(a, s);
    return t32 ? launch_halo<f16_mfma, 8, 32>(a, s) : launch_halo<f16_mfma, 16, 16>(a, s);
}
