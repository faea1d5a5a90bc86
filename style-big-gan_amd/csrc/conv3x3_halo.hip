// conv3x3_halo.hip -- 3x3 stride-1 (pad-same) convolution on the matrix cores with halo-staged activations.
//
// Measured on the gather kernel (conv_igemm.hip, 128 x 128 x 32 tile, LDS-DMA pipeline): with the MFMAs ablated the kernel
// still takes 87 % of its time, and the DMA stream alone 64 % -- it is bound by L2 -> LDS traffic (16 KB per K-step per
// workgroup, ~26 B/clk/CU), not by the matrix pipe.  This kernel moves a third of the bytes per flop: a workgroup owns a
// TH x TW patch of output pixels (256 pixels) of one image; per 32-channel slice the (TH+2) x (TW+2) input halo is staged
// ONCE and all nine taps read shifted windows of it from LDS, so a K-step (slice, tap) fetches only its 8 KB weight tile
// plus a ninth of the 22 KB halo.
//
// Same contract as sbg_conv2d_igemm restricted to: stride 1, the nine taps of a 3x3 window (any order, |offset| <= 1),
// output grid == input grid with H % TH == 0 and W % TW == 0; optional fused epilogue (see sbg_conv3x3_params):
//   y = clamp(act(acc * oscale[n, co] + noise[n, pixel] + bias[co]) * gain)          (demodulation + noise + bias_act)
// (`iscale` is not supported by this kernel: modulation is applied to the activations or the weights beforehand.)
//
// Workgroup = 256 lanes = 2 (cout) x 2 (pixel) waves, tile 128 cout x 256 pixels, wave tile 64 x 128 (4 x 8 MFMA tiles).
// Everything is staged by `buffer_load_dwordx4 ... lds`: weights 4 stages deep (3 K-steps in flight; deeper needs a third halo buffer), the next slice's
// halo in 16-pixel pieces spread over the current slice's last six taps, behind one counted s_waitcnt vmcnt + raw s_barrier per
// K-step.  Image borders / ragged channels are fetched at an out-of-range buffer offset (hardware returns zeros).
// LDS images are [row][4 slots of 16 B] written lane-linearly; the k-group -> slot XOR swizzles (weights: (-(row>>2))&3,
// halo: ((pixel>>2)&1)<<1 -- conflict-free for a 16-pixel fragment at ANY pixel alignment, found by exhaustive search)
// are applied on the source address and on the fragment reads.
#include "sbg_common.h"
#include <cstdlib>

namespace {

struct bf16_mfma {}; struct f16_mfma {};
template <class MF> struct Mfma;
template <> struct Mfma<bf16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<f16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

struct HaloArgs {
    const unsigned short* x; const unsigned short* w; void* y;
    const float* oscale; const float* noise; const float* bias;
    int ydtype;
    int N, H, W, Cin, Cout;
    int xs_n, xs_h, xs_w;                       // element strides (tensors < 2 GiB)
    int64_t ys_n, ys_h, ys_w;
    int ws_slab, ws_co; int64_t noise_sn;
    int tap_dy[9], tap_dx[9], tap_slab[9];
    int act; float alpha, gain, clamp;
    int accumulate;
    int tiles_x, tiles_y, ctiles;
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define SBG_OOB_OFFSET 0x80000000u

template <class MF, int TH, int TW>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(HaloArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int BC = 128, BP = TH * TW, NT = 9;
    static_assert(BP == 256 && (TW % 16) == 0, "256-pixel tile with 16-pixel row segments");
    constexpr int PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
    constexpr int HPIECES = (NPIX + 15) / 16;               // 16-pixel DMA pieces per halo
    static_assert(HPIECES <= 6 * 4, "halo pieces must fit taps 0..5 x 4 waves");
    constexpr int HALO_BYTES = HPIECES * 1024, WT_BYTES = BC * 64;
    constexpr int NSTAGE = 4, DEPTH = 3, PER_STEP = 3;     // weight stages / K-steps of loads in flight (bytes in flight, not
                                                            // bandwidth, bound the DMA stream: ~2 us round trip under load);
                                                            // DMA instructions per wave per K-step: 2 weight + 1 halo / filler
    constexpr int SEG = TW / 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;                               // 2 halo buffers
    unsigned char* sW = smem + 2 * HALO_BYTES;              // NSTAGE weight stages
    unsigned char* sDummy = sW + NSTAGE * WT_BYTES;         // 1 KiB sink for the filler DMA

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware order: workgroups sharing an XCD (b mod 8) walk neighbouring tiles -> halo rows re-used from that L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ct = bid % p.ctiles; int t = bid / p.ctiles;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int c0 = ct * BC, y0 = ty * TH, x0 = tx * TW;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);

    // ---- DMA lane coordinates: a piece = 16 rows x 4 slots of 16 B; lane -> (row = lane / 4, slot = lane & 3)
    const int lrow = lane >> 2, lslot = lane & 3;
    // weights: rows (wave*2 + i)*16 + lrow, source k-group = slot ^ ((-(lrow >> 2)) & 3)
    const int wg_src = lslot ^ ((-(lrow >> 2)) & 3);
    unsigned a_base[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int co = c0 + (wave * 2 + i) * 16 + lrow;
        a_base[i] = (co < p.Cout) ? (unsigned)(co * p.ws_co) * 2u : SBG_OOB_OFFSET;
    }
    // halo piece handled by this wave at tap slot `tp` (tp = 3..8): piece = (tp - 3) * 4 + wave, pixel pp = piece*16 + lrow
    // source k-group = slot ^ (((pp >> 2) & 1) << 1); byte offset of the pixel (channel 0) or OOB
    const int kchunks = (p.Cin + 31) >> 5;
    const int nsteps = kchunks * NT;

    auto issue = [&](int step) {
        const int chunk = step / NT, tp = step - chunk * NT;
        // -- weights of `step` into stage step % NSTAGE
        {
            const int ck = chunk * 32 + wg_src * 8;
            const unsigned wtap = (unsigned)(p.tap_slab[tp] * p.ws_slab + ck) * 2u;
            unsigned char* st = sW + (step % NSTAGE) * WT_BYTES;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const unsigned okm = 0u - (unsigned)((ck < p.Cin) & (a_base[i] != SBG_OOB_OFFSET));
                const unsigned off = ((a_base[i] + wtap) & okm) | (SBG_OOB_OFFSET & ~okm);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + (wave * 2 + i) * 1024), 16, off, 0, 0, 0);
            }
        }
        // -- one halo piece of slice chunk + 1 (taps 3..8: issued 3 iterations earlier, i.e. while slice `chunk` is already
        //    being consumed, so the buffer slice chunk - 1 used is free), otherwise a filler: PER_STEP instructions always
        {
            const int piece = (tp - 3) * 4 + wave;
            const bool real = (tp >= 3) && (piece < HPIECES) && (chunk + 1 < kchunks);
            const int pp = piece * 16 + lrow;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const int ck = (chunk + 1) * 32 + (lslot ^ (((pp >> 2) & 1) << 1)) * 8;
            const unsigned okm = 0u - (unsigned)(real & (pp < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W) & (ck < p.Cin));
            unsigned off = ((unsigned)(n * p.xs_n + iy * p.xs_h + ix * p.xs_w + ck) * 2u & okm) | (SBG_OOB_OFFSET & ~okm);
            unsigned char* dst = real ? sH + ((chunk + 1) & 1) * HALO_BYTES + piece * 1024 : sDummy;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, off, 0, 0, 0);
        }
    };
    // first slice's halo: every wave loads its share up front (pieces wave, wave + 4, ...)
    auto issue_first_halo = [&]() {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int piece = i * 4 + wave;
            const bool real = piece < HPIECES;
            const int pp = piece * 16 + lrow;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const int ck = (lslot ^ (((pp >> 2) & 1) << 1)) * 8;
            const unsigned okm = 0u - (unsigned)(real & (pp < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W) & (ck < p.Cin));
            const unsigned off = ((unsigned)(n * p.xs_n + iy * p.xs_h + ix * p.xs_w + ck) * 2u & okm) | (SBG_OOB_OFFSET & ~okm);
            unsigned char* dst = real ? sH + piece * 1024 : sDummy;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, off, 0, 0, 0);
        }
    };

    // ---- MFMA coordinates -------------------------------------------------------------------------------------------
    const int wc = (wave >> 1) * 64;                 // cout offset of this wave
    const int wseg0 = (wave & 1) * 8;                // first of the wave's eight 16-pixel segments
    const int fr = lane & 15, fg = lane >> 4;
    const int rdA = (fg ^ ((-(fr >> 2)) & 3)) * 16;
    float4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    int seg_pp[8];                                   // halo pixel of (segment j, lane fr) for tap offset (0, 0)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int s = wseg0 + j, r = s / SEG, cseg = (s - r * SEG) * 16;
        seg_pp[j] = (r + 1) * PW + cseg + fr + 1;
    }

    issue_first_halo();
#pragma unroll
    for (int s = 0; s < DEPTH; s++) if (s < nsteps) issue(s);

    for (int s = 0; s < nsteps; s++) {
        // this wave's loads of step s have landed once at most (steps issued after s) * PER_STEP remain outstanding
        const int ahead = (nsteps - 1 - s < DEPTH - 1) ? nsteps - 1 - s : DEPTH - 1;
        switch (ahead) {
#define SBG_WAIT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(%0)" :: "n"((k) * PER_STEP) : "memory"); break;
            SBG_WAIT_CASE(0) SBG_WAIT_CASE(1) SBG_WAIT_CASE(2) SBG_WAIT_CASE(3) SBG_WAIT_CASE(4) SBG_WAIT_CASE(5)
            SBG_WAIT_CASE(6) SBG_WAIT_CASE(7) SBG_WAIT_CASE(8) SBG_WAIT_CASE(9)
            default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 1) * PER_STEP) : "memory"); break;
#undef SBG_WAIT_CASE
        }
        __builtin_amdgcn_s_barrier();
        if (s + DEPTH < nsteps) issue(s + DEPTH);
        const int chunk = s / NT, tp = s - chunk * NT;
        const unsigned char* hb = sH + (chunk & 1) * HALO_BYTES;
        const unsigned char* wb = sW + (s % NSTAGE) * WT_BYTES;
        const int shift = p.tap_dy[tp] * PW + p.tap_dx[tp];
        short8_t fa[4], fb[8];
#pragma unroll
        for (int i = 0; i < 4; i++) fa[i] = *reinterpret_cast<const short8_t*>(wb + (wc + 16 * i + fr) * 64 + rdA);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pp = seg_pp[j] + shift;
            fb[j] = *reinterpret_cast<const short8_t*>(hb + (pp << 6) + ((fg << 4) ^ ((pp & 4) << 3)));
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = Mfma<MF>::run(fa[i], fb[j], acc[i][j]);
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
static int launch_halo(HaloArgs& a, unsigned x_bytes, unsigned w_bytes, hipStream_t stream)
{
    constexpr int NPIX = (TH + 2) * (TW + 2), HPIECES = (NPIX + 15) / 16;
    constexpr int lds = 2 * HPIECES * 1024 + 4 * 128 * 64 + 1024;
    a.tiles_x = a.W / TW; a.tiles_y = a.H / TH; a.ctiles = (a.Cout + 127) / 128;
    const int64_t nblk = (int64_t)a.N * a.tiles_x * a.tiles_y * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv3x3: grid too large");
    auto kern = conv3x3_halo_kernel<MF, TH, TW>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return sbg_fail(SBG_ERR_LAUNCH, "conv3x3: cannot raise the dynamic LDS limit to %d bytes", lds);
        attr_set = true;
    }
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    const double P = (double)a.N * a.H * a.W;
    SbgProfScope prof(stream, SBG_K_CONV3X3_HALO, 2.0 * P * a.Cout * (double)a.Cin * 9,
                      2.0 * P * a.Cin + 2.0 * 9 * a.Cout * (double)a.Cin + ys * P * a.Cout * (a.accumulate ? 2 : 1),
                      {(int)P, a.Cout, a.Cin, 9, 1, a.H, TH * 1000 + TW});
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, stream, a, x_bytes, w_bytes);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

static int64_t x_extent_bytes(const sbg_conv3x3_params* q)
{
    return 2 * ((int64_t)(q->N - 1) * q->xs_n + (int64_t)(q->H - 1) * q->xs_h + (int64_t)(q->W - 1) * q->xs_w + q->Cin);
}
static int64_t w_extent_bytes(const sbg_conv3x3_params* q)
{
    int maxslab = 0;
    for (int t = 0; t < q->ntaps; t++) if (q->tap_slab[t] > maxslab) maxslab = q->tap_slab[t];
    return 2 * ((int64_t)maxslab * q->ws_slab + (int64_t)(q->Cout - 1) * q->ws_co + q->Cin);
}

} // namespace

extern "C" int sbg_conv3x3_supported(const sbg_conv3x3_params* q)
{
    if (!q) return 0;
    const bool t32 = (q->W % 32 == 0) && (q->H % 8 == 0);
    const bool t16 = (q->W % 16 == 0) && (q->H % 16 == 0);
    if (!t32 && !t16) return 0;
    if (q->ntaps != 9 || (q->Cin % 8) != 0 || q->iscale != nullptr) return 0;
    for (int t = 0; t < 9; t++)
        if (q->tap_dy[t] < -1 || q->tap_dy[t] > 1 || q->tap_dx[t] < -1 || q->tap_dx[t] > 1 || q->tap_slab[t] < 0) return 0;
    if (q->xs_n < 0 || q->xs_h < 0 || q->xs_w < 0 || q->ws_slab < 0 || q->ws_co < 0) return 0;
    if (x_extent_bytes(q) >= (int64_t)SBG_OOB_OFFSET || w_extent_bytes(q) >= (int64_t)SBG_OOB_OFFSET) return 0;
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
    if (q->N == 0) return SBG_OK;
    HaloArgs a;
    a.x = (const unsigned short*)q->x; a.w = (const unsigned short*)q->w; a.y = q->y;
    a.oscale = q->oscale; a.noise = q->noise; a.bias = q->bias;
    a.ydtype = q->ydtype;
    a.N = q->N; a.H = q->H; a.W = q->W; a.Cin = q->Cin; a.Cout = q->Cout;
    a.xs_n = (int)q->xs_n; a.xs_h = (int)q->xs_h; a.xs_w = (int)q->xs_w; a.ys_n = q->ys_n; a.ys_h = q->ys_h; a.ys_w = q->ys_w;
    a.ws_slab = (int)q->ws_slab; a.ws_co = (int)q->ws_co; a.noise_sn = q->noise_stride_n;
    for (int t = 0; t < 9; t++) { a.tap_dy[t] = q->tap_dy[t]; a.tap_dx[t] = q->tap_dx[t]; a.tap_slab[t] = q->tap_slab[t]; }
    a.act = q->act; a.alpha = q->alpha; a.gain = q->gain; a.clamp = q->clamp;
    a.accumulate = q->accumulate;
    hipStream_t s = (hipStream_t)stream;
    const unsigned xb = (unsigned)x_extent_bytes(q), wb = (unsigned)w_extent_bytes(q);
    const bool t32 = (q->W % 32 == 0) && (q->H % 8 == 0);
    if (q->xdtype == SBG_BF16) return t32 ? launch_halo<bf16_mfma, 8, 32>(a, xb, wb, s) : launch_halo<bf16_mfma, 16, 16>(a, xb, wb, s);
    return t32 ? launch_halo<f16_mfma, 8, 32>(a, xb, wb, s) : launch_halo<f16_mfma, 16, 16>(a, xb, wb, s);
}
