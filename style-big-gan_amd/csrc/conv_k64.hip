// conv_k64.hip -- implicit-GEMM convolution on the gfx950 matrix cores, K-step 64, LDS-DMA staging in full 128-B lines.
//
// Same launch contract as conv_igemm.hip (tap list, see include/sbg_hip.h); this file holds the kernels that carry the
// flops of the StyleGAN2 step.  Design points (measured with rocprofv3 / ablation builds):
//   * The 128 x 128 x 32 kernel of conv_igemm.hip was bound by the L2 -> LDS stream (64-B row pieces = half a cache line
//     per request, 15.6 B per kflop).  Here a K-step is 64 channels deep, so every staged row is one whole 128-B line
//     (8 rows x 128 B per wave instruction) and the tile is 128 output channels x 256 pixels.
//   * conv_halo_ld_kernel (3x3 / stride 1 / output grid == input grid): a workgroup owns a TH x TW patch of one image; per
//     64-channel slice the (TH + 2) x (TW + 2) input halo is staged ONCE and all nine taps read shifted windows of it, so
//     a K-step fetches its 16 KB weight tile plus a ninth of the 43 KB halo: 5 B per kflop.  Persistent (one workgroup
//     per CU walks its tiles, the pipeline runs through tile boundaries) with dedicated loader waves.
//   * conv_k64_kernel (everything else: strided, transposed phases, 1x1): im2col on the fly, every wave stages its share.
//   * Compute waves run as two groups in ping-pong (one wave of each per SIMD): while one group reads its MFMA fragments
//     from LDS the other owns the matrix pipe; per-step control flow is compile-time (unrolled taps) or two-way scalar.
//   * Output channels are permuted inside each 32-channel block so that a lane ends up with 8 CONSECUTIVE channels of a
//     pixel: the epilogue stores 16 B (bf16) per lane, straight-line code specialised on dtype / fused tail.
//
// LDS images are [row][8 slots x 16 B] (128 B per row = 64 channels of one tap), written lane-linearly by
// `buffer_load_dwordx4 ... lds` (lane -> row = lane >> 3, slot = lane & 7).  The k-slot -> slot XOR swizzle
// slot = kslot ^ (row & 7) is applied on the SOURCE address and on the fragment reads: ds_read_b128 of 16 consecutive
// rows is conflict-free at any row alignment (SQ_LDS_BANK_CONFLICT = 0 measured), which the shifted halo windows need.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

using namespace sbgconv;

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define SBG_OOB_OFFSET 0x80000000u      // >= num_records of every descriptor built below (tensors < 2 GiB): reads as zeros

// LDS row R of the weight tile holds output channel c0 + chmap(R): inside a 32-row block, MFMA row m of the even / odd
// 16-row tile maps to channel 8 (m / 4) + 4 (tile & 1) + (m % 4), so accumulator tiles (2h, 2h + 1) of a lane hold
// channels 32 h + 8 fg + {0..3} and {4..7}.
static __device__ __forceinline__ int chmap(int R) { return (R & ~31) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3); }

template <int... Is, class F>
static __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
static __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }   // compile-time unrolled loop

template <int N> static __device__ __forceinline__ void wait_vmcnt_const() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// Straight-line fast path of the epilogue (layout of the accumulators: see conv_epilogue8 below): the output dtype and "no epilogue math" are template parameters, every 8-channel group of the
// wave is in range and 16-B aligned (checked by the caller), so the loops below carry no per-element guards and no dtype
// branches: leaky ReLU with alpha 0 / 1 covers ReLU / linear, clamp = med3 with an infinite bound when disabled.
// `lp`: the tile's epilogue parameters staged in LDS by a loader wave (halo kernel: [noise of the TH x TW tile, row-major | bias of the 128 tile
// channels | demodulation coefficients of (image, 128 tile channels)], 1 KiB each, fp32) -- the loads below then come from LDS (~100 cycles)
// instead of L2 / HBM (a round trip of 1-2 us under load, paid by both compute groups at every tile boundary); lp_c = channel offset of this wave
// inside the tile, lp_pix(j) = the pixel's index inside the tile.
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const float4_t lds_cfloat4;
struct LdsParams { lds_cfloat* base; int c; int tw; };
template <int TC, int TP, int YDT, bool PLAIN, bool NUNI, bool LP = false, class PixFn>
static __device__ __forceinline__ void conv_epilogue_fast(const ConvArgs& p, float4_t (&acc)[TC][TP], int cbase, int fg, PixFn pix, int64_t ybase,
                                                          const LdsParams* lp = nullptr, int y0 = 0, int x0 = 0)
{
    // NUNI: every pixel of the wave lies in one image (halo kernel), so the demodulation coefficients are per-h constants.
    // Every parameter load (noise per pixel, bias and demodulation coefficients per channel group) is issued up front and UNCONDITIONALLY -- an
    // absent term reads a valid dummy address (the first 16 bytes of x) and is replaced by a select -- so the wave waits for ONE round trip.
    // (With `if (p.noise) nz = p.noise[...]` per pixel the compiler emitted load, s_waitcnt vmcnt(0), branch join four times over, plus two
    // more waits for bias and coefficients: six exposed L2 / HBM latencies per tile, more than the arithmetic.)
    constexpr int TH2 = TC / 2;
    const float alpha = (p.act == SBG_ACT_LRELU) ? p.alpha : (p.act == SBG_ACT_RELU ? 0.f : 1.f);
    const float lsel = alpha <= 1.f ? __builtin_inff() : -__builtin_inff();      // leaky ReLU = med3(u, alpha u, +inf) = max for alpha <= 1, min (-inf) above
    const float cl = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    const float gain = p.gain;
    const float* const dummy = reinterpret_cast<const float*>(p.x);
    const bool has_nz = !PLAIN && p.noise != nullptr, has_b = !PLAIN && p.bias != nullptr, has_s = !PLAIN && p.oscale != nullptr;
    int64_t yoff[TP]; float nz[TP]; bool ok[TP]; int nn[TP];
#pragma unroll
    for (int j = 0; j < TP; j++) {
        int n, oy, ox;
        ok[j] = pix(j, n, oy, ox);                      // (an out-of-range pixel still decodes to valid coordinates)
        nn[j] = n;
        yoff[j] = ybase + (int64_t)blockIdx.y * p.y_split_stride + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w + cbase + 8 * fg;
        nz[j] = 0.f;
        if (!PLAIN) {
            if constexpr (LP) nz[j] = lp->base[(oy - y0) * lp->tw + (ox - x0)];
            else              nz[j] = *(has_nz ? p.noise + ((int64_t)n * p.noise_sn + (int64_t)oy * p.OW + ox) : dummy);
        }
    }
    float4_t b_lo[TH2], b_hi[TH2], s_lo[TH2], s_hi[TH2];
#pragma unroll
    for (int h = 0; h < TH2; h++) {
        b_lo[h] = b_hi[h] = float4_t{0.f, 0.f, 0.f, 0.f};
        s_lo[h] = s_hi[h] = float4_t{1.f, 1.f, 1.f, 1.f};
        if (!PLAIN) {
            if constexpr (LP) {
                lds_cfloat* b = lp->base + 256 + lp->c + 32 * h + 8 * fg;
                b_lo[h] = *(lds_cfloat4*)b; b_hi[h] = *(lds_cfloat4*)(b + 4);
                if (NUNI) { s_lo[h] = *(lds_cfloat4*)(b + 256); s_hi[h] = *(lds_cfloat4*)(b + 260); }
            } else {
            const float* b = has_b ? p.bias + cbase + 32 * h + 8 * fg : dummy;
            b_lo[h] = *reinterpret_cast<const float4_t*>(b); b_hi[h] = *reinterpret_cast<const float4_t*>(has_b ? b + 4 : dummy);
            if (NUNI) {
                const float* sc = has_s ? p.oscale + (int64_t)nn[0] * p.Cout + cbase + 32 * h + 8 * fg : dummy;
                s_lo[h] = *reinterpret_cast<const float4_t*>(sc); s_hi[h] = *reinterpret_cast<const float4_t*>(has_s ? sc + 4 : dummy);
            }
            }
        }
    }
    if (!PLAIN) {
#pragma unroll
        for (int j = 0; j < TP; j++) nz[j] = has_nz ? nz[j] : 0.f;
#pragma unroll
        for (int h = 0; h < TH2; h++) {
            if (!has_b) { b_lo[h] = float4_t{0.f, 0.f, 0.f, 0.f}; b_hi[h] = b_lo[h]; }
            if (!NUNI || !has_s) { s_lo[h] = float4_t{1.f, 1.f, 1.f, 1.f}; s_hi[h] = s_lo[h]; }
        }
    }
#pragma unroll
    for (int h = 0; h < TH2; h++) {
#pragma unroll
        for (int j = 0; j < TP; j++) {
            if (!ok[j]) continue;
            float4_t lo = acc[2 * h][j], hi = acc[2 * h + 1][j];
            if (!PLAIN) {
                float4_t sl = s_lo[h], sh = s_hi[h];
                if (!NUNI && has_s) {
                    const float* sc = p.oscale + (int64_t)nn[j] * p.Cout + cbase + 32 * h + 8 * fg;
                    sl = *reinterpret_cast<const float4_t*>(sc); sh = *reinterpret_cast<const float4_t*>(sc + 4);
                }
                lo = lo * sl + (nz[j] + b_lo[h]);
                hi = hi * sh + (nz[j] + b_hi[h]);
                const float4_t tl = lo * alpha, th = hi * alpha;       // vector forms: v_pk_mul_f32
#pragma unroll
                for (int e = 0; e < 4; e++) { lo[e] = __builtin_amdgcn_fmed3f(lo[e], tl[e], lsel); hi[e] = __builtin_amdgcn_fmed3f(hi[e], th[e], lsel); }
                lo = lo * gain; hi = hi * gain;
#pragma unroll
                for (int e = 0; e < 4; e++) { lo[e] = __builtin_amdgcn_fmed3f(lo[e], -cl, cl); hi[e] = __builtin_amdgcn_fmed3f(hi[e], -cl, cl); }
            }
            if (YDT == SBG_F32) {
                float* dst = (float*)p.y + yoff[j] + 32 * h;
                if (p.accumulate) { lo += *reinterpret_cast<float4_t*>(dst); hi += *reinterpret_cast<float4_t*>(dst + 4); }
                *reinterpret_cast<float4_t*>(dst) = lo;
                *reinterpret_cast<float4_t*>(dst + 4) = hi;
            } else {
                short8_t o;
#pragma unroll
                for (int e = 0; e < 4; e++) { o[e] = (short)f32_to_bf16_bits(lo[e]); o[4 + e] = (short)f32_to_bf16_bits(hi[e]); }
                *reinterpret_cast<short8_t*>((unsigned short*)p.y + yoff[j] + 32 * h) = o;
            }
        }
    }
}


// Epilogue: lane (fr, fg) holds, for the pixel of fragment column fr in segment j, channels cbase + 32 h + 8 fg + e with
// e = 0..3 in acc[2h][j] and e = 4..7 in acc[2h + 1][j].  pix(j, n, oy, ox) -> in range?
template <int TC, int TP, bool NUNI = false, class PixFn>
static __device__ __forceinline__ void conv_epilogue8(const ConvArgs& p, float4_t (&acc)[TC][TP], int cbase, int fg, PixFn pix, int64_t ybase = 0,
                                                      const LdsParams* lp = nullptr, int y0 = 0, int x0 = 0)
{
    constexpr int TH2 = TC / 2;
    const bool plain = (p.act <= SBG_ACT_LINEAR) && p.gain == 1.f && p.clamp < 0.f && !p.bias && !p.noise && !p.oscale;
    // fast path: the wave's whole channel range is valid, rows and per-channel vectors 16-B aligned, bf16 / fp32 output
    const bool fast = (cbase + 16 * TC <= p.Cout) && ((p.Cout & 7) == 0) && ((((uintptr_t)p.y) & 15) == 0) && (((p.ys_n | p.ys_h | p.ys_w) & 7) == 0)
                      && ((((uintptr_t)p.oscale) & 15) == 0) && ((((uintptr_t)p.bias) & 15) == 0) && p.ydtype != SBG_F16;
    if (fast) {
        if (p.ydtype == SBG_BF16) {
            if (plain) conv_epilogue_fast<TC, TP, SBG_BF16, true, NUNI>(p, acc, cbase, fg, pix, ybase);
            else if (lp) conv_epilogue_fast<TC, TP, SBG_BF16, false, NUNI, true>(p, acc, cbase, fg, pix, ybase, lp, y0, x0);
            else       conv_epilogue_fast<TC, TP, SBG_BF16, false, NUNI>(p, acc, cbase, fg, pix, ybase);
        } else {
            if (plain) conv_epilogue_fast<TC, TP, SBG_F32, true, NUNI>(p, acc, cbase, fg, pix, ybase);
            else if (lp) conv_epilogue_fast<TC, TP, SBG_F32, false, NUNI, true>(p, acc, cbase, fg, pix, ybase, lp, y0, x0);
            else       conv_epilogue_fast<TC, TP, SBG_F32, false, NUNI>(p, acc, cbase, fg, pix, ybase);
        }
        return;
    }
    float bias8[TH2][8];
#pragma unroll
    for (int h = 0; h < TH2; h++) {
        const int co = cbase + 32 * h + 8 * fg;
#pragma unroll
        for (int e = 0; e < 8; e++) bias8[h][e] = (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TP; j++) {
        int n, oy, ox;
        if (!pix(j, n, oy, ox)) continue;
        const int64_t yoff = ybase + (int64_t)blockIdx.y * p.y_split_stride + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        const float nz = (!plain && p.noise) ? p.noise[(int64_t)n * p.noise_sn + (int64_t)oy * p.OW + ox] : 0.f;
#pragma unroll
        for (int h = 0; h < TH2; h++) {
            const int co = cbase + 32 * h + 8 * fg;
            if (co >= p.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = acc[2 * h][j][e]; v[4 + e] = acc[2 * h + 1][j][e]; }
#pragma unroll 1
            for (int e = 0; e < 8; e++) {            // rolled: this path serves odd channel counts (ToRGB, tails), not the flops
                if (co + e >= p.Cout) break;
                float u = v[e];
                if (!plain) {
                    if (p.oscale) u *= p.oscale[(int64_t)n * p.Cout + co + e];
                    u += nz + bias8[h][e];
                    if (p.act == SBG_ACT_LRELU) u = (u > 0.f) ? u : u * p.alpha;
                    else if (p.act == SBG_ACT_RELU) u = (u > 0.f) ? u : 0.f;
                    u *= p.gain;
                    if (p.clamp >= 0.f) u = (u > -p.clamp && u < p.clamp) ? u : (u >= 0.f ? p.clamp : -p.clamp);
                }
                if (p.ydtype == SBG_F32) {
                    float* dst = (float*)p.y + yoff + co + e;
                    *dst = p.accumulate ? *dst + u : u;
                } else {
                    ((unsigned short*)p.y)[yoff + co + e] = (p.ydtype == SBG_BF16) ? f32_to_bf16_bits(u) : f32_to_f16_bits(u);
                }
            }
        }
    }
}

// Gather (im2col-on-the-fly) kernel: any tap list and stride; one 128 (or 64) x 256 (or 128) tile per workgroup, 8 waves.
template <class MF, int BC, int BP, int WGC, int WGP>
__global__ __launch_bounds__(512, 2) void conv_k64_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int NW = 8;
    static_assert(WGC * WGP == NW, "8 waves per workgroup");
    constexpr int WC = BC / WGC, WP = BP / WGP;        // wave tile
    constexpr int TC = WC / 16,  TP = WP / 16;         // 16 x 16 MFMA tiles per wave
    static_assert(WC % 32 == 0 && WP % 16 == 0 && BC % 64 == 0 && BP % 64 == 0, "tile shape");
    constexpr int IA = BC / 64;                        // weight pieces (8 rows x 128 B) per wave per K-step
    constexpr int IB = BP / 64;                        // gathered pixel pieces per wave per K-step
    constexpr int NSTAGE = 3;                          // LDS stages; the loads of step s + NSTAGE - 1 are issued in step s
    constexpr int LEAD = NSTAGE - 1;
    constexpr int A_BYTES = BC * 128, B_BYTES = BP * 128, STAGE = A_BYTES + B_BYTES;

#ifdef SBG_K64_DEBUG     // ablations for diagnosis: 1 = no MFMA, 2 = no DMA inside the K loop, 4 = no fragment reads, 8 = no epilogue
    const int dbg = p.debug;
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpY = wave >= 4;
    int bid = blockIdx.x;
    {   // XCD-aware tile order: workgroups b and b + 8 share an XCD (L2); give each XCD a contiguous run of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int c0 = (bid % p.ctiles) * BC, p0 = (bid / p.ctiles) * BP;     // one tile per workgroup

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);

    // ---- DMA lane coordinates: lane -> (row = 8 piece + (lane >> 3), slot = lane & 7), source k-slot = slot ^ (row & 7)
    const int lrow = lane >> 3;
    const int src_k = ((lane & 7) ^ lrow) * 8;          // channel offset inside the 64-channel slice
    unsigned a_base[IA];                                // byte offset of (channel row, src_k) in a weight slab, or out of range
#pragma unroll
    for (int i = 0; i < IA; i++) {
        const int co = c0 + chmap((wave * IA + i) * 8 + lrow);
        a_base[i] = (co < p.Cout) ? (unsigned)(co * (int)p.ws_co + src_k) * 2u : SBG_OOB_OFFSET;
    }
    const int kchunks = (p.Cin + 63) >> 6;
    const int nsteps_all = p.ntaps * kchunks;
    int nsteps = nsteps_all;
    // per-tap constants live in lane t of a VGPR and are fetched with v_readlane (no scalar-memory latency inside the K loop)
    const int tl = lane < p.ntaps ? lane : 0;
    const int tbl_dy = p.tap_dy[tl], tbl_dx = p.tap_dx[tl];
    const int tbl_wtap = p.tap_slab[tl] * (int)p.ws_slab * 2;

    // weights of (tap t, slice chunk) -> stage `stage`
    auto issue_w = [&](int t, int chunk, int stage) {
        unsigned char* st = smem + stage * STAGE;
        // the step's (tap, slab) offset is a scalar and rides in the instruction's scalar offset: a piece is the DMA alone.  Only the last slab of a
        // ragged Cin masks lanes (a masked-off lane reads out of range -> zeros; skipping it would leave stale bytes in LDS)
        const int wtap = __builtin_amdgcn_readlane(tbl_wtap, t) + chunk * 128;
        const bool ragged = chunk * 64 + 64 > p.Cin;          // (wave-uniform)
#pragma unroll
        for (int i = 0; i < IA; i++) {
            unsigned off = a_base[i];
            if (ragged) off = (chunk * 64 + src_k < p.Cin) ? off : SBG_OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + (wave * IA + i) * 1024), 16, off, wtap, 0, 0);
        }
    };

    // ---- MFMA coordinates -------------------------------------------------------------------------------------------
    const int wci = wave / WGP, wpi = wave % WGP;
    const int wc = wci * WC, wp = wpi * WP;
    const int fr = lane & 15, fg = lane >> 4;
    const int frag_off = fr * 128 + ((fg ^ (fr & 7)) << 4);          // k-sub 0; k-sub 1 = ^ 64
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    short8_t fa[2][TC], fb[2][TP];

    auto read_a = [&](int stage) {
        const unsigned char* sa = smem + stage * STAGE + wc * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TC; i++) fa[ks][i] = *reinterpret_cast<const short8_t*>(sa + i * 16 * 128 + (frag_off ^ (ks * 64)));
    };
    auto mma = [&]() {
        if (dbg & 1) return;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TC; i++)
#pragma unroll
                for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[ks][i], fb[ks][j], acc[i][j]);
    };

    // Ping-pong schedule: waves 0-3 (X) and 4-7 (Y) -- one of each per SIMD -- run half a K-step apart, separated by two
    // barriers per step: while one group issues its LDS reads and DMAs the other owns the matrix pipe.  Global barrier index:
    // X: B_a(s) = 2s, B_b(s) = 2s + 1;  Y: one extra barrier first, B_a(s) = 2s + 1, B_b(s) = 2s + 2.
    //   RAW: every wave's counted vmcnt wait for step s precedes a barrier with index <= 2s (X: before B_a(s); Y: before
    //        B_b(s - 1)), and a stage is read only behind B_a(s).
    //   WAR: reads of step s are retired (lgkmcnt(0)) before B_b(s) <= 2s + 2; stage s % NSTAGE is refilled by the loads of
    //        step s + NSTAGE, issued behind B_a(s + 1) >= 2s + 2.
    // Everything that decides a vmcnt count is a compile-time constant (unrolled taps) or a two-way scalar branch.

    {
        int b_iy0[IB], b_ix0[IB]; unsigned b_base[IB];
#pragma unroll
        for (int i = 0; i < IB; i++) {
            const int pix = p0 + (wave * IB + i) * 8 + lrow;
            const bool ok = pix < p.P;
            const int pp = ok ? pix : 0;
            const int ox = pp % p.OW, t = pp / p.OW, oy = t % p.OH, n = t / p.OH;
            b_iy0[i] = ok ? oy * p.stride : -(1 << 28);            // invalid rows fail every range test below
            b_ix0[i] = ox * p.stride;
            b_base[i] = (unsigned)(n * (int)p.xs_n + src_k) * 2u;
        }
        // Steps are tap-major, so the gathered pixel of a lane (and whether it lies in the image) changes only once per kchunks steps: its byte offset
        // is worked out when the tap changes (x_off[], ~12 vector instructions per piece) and the slab within the tap rides in the instruction's scalar
        // offset.  These waves issue the MFMAs too; at Cin = 512 seven of eight steps now issue their four pixel pieces with no arithmetic at all.
        unsigned x_off[IB]; int x_tap = -1;
        auto issue_x = [&](int t, int chunk, int stage) {          // gathered pixels of (tap t, slice chunk) -> stage
            unsigned char* st = smem + stage * STAGE + A_BYTES;
            if (t != x_tap) {                                      // (wave-uniform)
                x_tap = t;
                const int dy = __builtin_amdgcn_readlane(tbl_dy, t), dx = __builtin_amdgcn_readlane(tbl_dx, t);
#pragma unroll
                for (int i = 0; i < IB; i++) {
                    const int iy = b_iy0[i] + dy, ix = b_ix0[i] + dx;
                    const bool ok = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
                    x_off[i] = ok ? b_base[i] + (unsigned)(iy * (int)p.xs_h + ix * (int)p.xs_w) * 2u : SBG_OOB_OFFSET;
                }
            }
            const bool ragged = chunk * 64 + 64 > p.Cin;           // (wave-uniform)
#pragma unroll
            for (int i = 0; i < IB; i++) {
                unsigned off = x_off[i];
                if (ragged) off = (chunk * 64 + src_k < p.Cin) ? off : SBG_OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(st + (wave * IB + i) * 1024), 16, off, chunk * 128, 0, 0);
            }
        };
        auto read_b = [&](int stage) {
            const unsigned char* sb = smem + stage * STAGE + A_BYTES + wp * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int j = 0; j < TP; j++) fb[ks][j] = *reinterpret_cast<const short8_t*>(sb + j * 16 * 128 + (frag_off ^ (ks * 64)));
        };
        constexpr int C = IA + IB;                       // DMA instructions per wave per step
        static_assert(LEAD == 2, "gather pipeline: two steps of loads in flight");
        // steps are tap-major: step = t * kchunks + chunk; (it, ic) = coordinates of the step whose loads are issued next
        // split K: workgroup blockIdx.y of gridDim.y takes steps [s_begin, s_begin + nsteps)
        int s_begin = 0;
        if (p.ksplit > 1) {
            const int per = (nsteps_all + p.ksplit - 1) / p.ksplit;
            s_begin = blockIdx.y * per;
            nsteps = nsteps_all - s_begin < per ? nsteps_all - s_begin : per;
            if (nsteps < 0) nsteps = 0;
        }
        int it = s_begin / kchunks, ic = s_begin - it * kchunks, wstage = 0;
        auto issue_next = [&]() {
            issue_w(it, ic, wstage); issue_x(it, ic, wstage);
            wstage = wstage == NSTAGE - 1 ? 0 : wstage + 1;
            if (++ic == kchunks) { ic = 0; it++; }
        };
        if (nsteps > 0) issue_next();
        if (nsteps > 1) { issue_next(); wait_vmcnt_const<C>(); } else wait_vmcnt_const<0>();
        if (grpY) __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int s = 0; s < nsteps; s++) {
            if (!grpY) { if (s + 1 < nsteps) wait_vmcnt_const<C>(); else wait_vmcnt_const<0>(); }     // step s; step s + 1 may be in flight
            __builtin_amdgcn_s_barrier();                // B_a
            if (!(dbg & 4)) { read_a(stage); read_b(stage); }
            if (s + LEAD < nsteps && !(dbg & 2)) issue_next();
            if (grpY) { if (s + 2 < nsteps) wait_vmcnt_const<C>(); else wait_vmcnt_const<0>(); }      // step s + 1; step s + 2 may be in flight
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();                // B_b
            __builtin_amdgcn_sched_barrier(0);
            mma();
            __builtin_amdgcn_sched_barrier(0);
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
        if (!grpY) __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    if (dbg & 8) return;
    conv_epilogue8<TC, TP>(p, acc, c0 + wc, fg, [&](int j, int& n, int& oy, int& ox) {
        const int pix = p0 + wp + 16 * j + fr;
        const int pp = pix < p.P ? pix : 0;
        ox = pp % p.OW; const int t = pp / p.OW; oy = t % p.OH; n = t / p.OH;
        return pix < p.P;
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo kernel with loader waves.  12 waves: 0-3 (X) and 4-7 (Y) compute in ping-pong as above but issue NO vector-memory
// instruction inside the K loop; waves 8-9 stream the weight tiles, waves 10-11 the halo.  Why: (1) a compute wave's
// non-MFMA phase shrinks to `barrier, 16 ds_read_b128, barrier`; (2) vmcnt retires in order, so in the 8-wave kernel a halo
// piece that misses L2 (537 MB activations do not fit the 256 MB Infinity Cache) stalled the weight stream queued behind it --
// here the halo loaders have a whole slice (nine K-steps) of lead and the weight loaders never wait for HBM.
// Every wave executes exactly 2 S + 1 workgroup barriers (S = K-steps of all tiles of this workgroup), index g = 0 .. 2S:
//   X: B_a(s) = 2s, B_b(s) = 2s + 1, final 2S;   Y: first 0, B_a(s) = 2s + 1, B_b(s) = 2s + 2;
//   weight loader: per step { wait W(s) landed; 2s; issue W(s + 3) into the stage step s - 1 used (its reads retired before
//                  2s); 2s + 1 }, final 2S;      halo loader: per slice c { vmcnt(0): halo(c) landed; 18c; issue halo(c + 1) into
//                  the buffer slice c - 1 used; 18c + 1 .. 18c + 17 }, final 2S.
template <class MF, int TH, int TW>
__global__ __launch_bounds__(768, 3) void conv_halo_ld_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int BC = 128, BP = 256, WGP = 4;
    constexpr int WC = 64, WP = 64, TC = 4, TP = 4;
    static_assert(TH * TW == BP, "halo tile = TH x TW pixels");
    constexpr int NSTAGE = 4, STAGE = BC * 128;        // weight stages: the loads of step s + 3 are issued while step s computes
    constexpr int PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
    constexpr int HPIECES = (NPIX + 7) / 8, HALO_BYTES = HPIECES * 1024;
    constexpr int NT = 9, SEG = TW / 16;
    constexpr int WPIECES = BC / 8 / 2;                // weight pieces per weight loader per step
    constexpr int HPL = (HPIECES + 1) / 2;             // halo pieces per halo loader per slice

#ifdef SBG_K64_DEBUG     // ablations for diagnosis: 1 = no MFMA, 4 = no fragment reads, 8 = no epilogue, 16 = no B_b barrier wait... (timing only)
    const int dbg = p.debug;
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sH = smem + NSTAGE * STAGE;
    unsigned char* const sP = sH + 2 * HALO_BYTES;     // epilogue parameters of the tile being computed: two buffers (tile parity) of 3 KiB
    constexpr int PARAM_BYTES = 3072;
#ifdef SBG_K64_STAMPS    // diagnosis build (scratch/halo_stamps.py): lane 0 of waves 0, 4, 8, 10 of workgroup 0 writes its clock in front of and behind every
                         // barrier into the head of y (4096 records per wave pair); the outputs of that launch are garbage
    unsigned long long* const stamp_base = (unsigned long long*)p.y + (threadIdx.x >> 7) * 4096;
    const bool stamp_on = blockIdx.x == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 4 || (threadIdx.x >> 6) == 8 || (threadIdx.x >> 6) == 10);
    int stamp_i = 0;
#define SBG_BARRIER() do { if (stamp_on && stamp_i < 4000) stamp_base[stamp_i] = __builtin_amdgcn_s_memtime(); stamp_i++; __builtin_amdgcn_s_barrier(); \
                           if (stamp_on && stamp_i < 4000) stamp_base[stamp_i] = __builtin_amdgcn_s_memtime(); stamp_i++; } while (0)
#else
#define SBG_BARRIER() __builtin_amdgcn_s_barrier()
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware tile order: workgroups b and b + 8 share an XCD (L2); give each XCD a contiguous run of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tiles_x = p.OW / TW, tiles_y = p.OH / TH;
    const int ntiles = p.ptiles * p.ctiles, G = gridDim.x;
    const int my_tiles = (ntiles - bid + G - 1) / G;                  // tiles bid, bid + G, ...  (the grid never exceeds ntiles)
    const int kchunks = (p.Cin + 63) >> 6;
    const int nslices = my_tiles * kchunks;                            // slice = one 64-channel slab of one tile = nine K-steps
    struct TileC { int c0, tn, y0, x0; };
    auto decode = [&](int tile) -> TileC {
        TileC r;
        const int ct_ = tile % p.ctiles; int pt_ = tile / p.ctiles;
        const int tx = pt_ % tiles_x; pt_ /= tiles_x;
        const int ty = pt_ % tiles_y;
        r.c0 = ct_ * BC; r.tn = pt_ / tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
        return r;
    };
    // tile + G without divisions: G in the mixed radix (ctiles, tiles_x, tiles_y), added with carries.  (decode()'s four scalar divisions are
    // ~1000 cycles; the compute waves ran them between a tile's last MFMA and the barrier their partner group was waiting at.)
    const TileC gstep = decode(G);                       // digits of G, scaled like the coordinates
    auto advance = [&](TileC t_) -> TileC {
        t_.c0 += gstep.c0; int carry = t_.c0 >= p.ctiles * BC; t_.c0 -= carry ? p.ctiles * BC : 0;
        t_.x0 += gstep.x0 + (carry ? TW : 0); carry = t_.x0 >= tiles_x * TW; t_.x0 -= carry ? tiles_x * TW : 0;
        t_.y0 += gstep.y0 + (carry ? TH : 0); carry = t_.y0 >= tiles_y * TH; t_.y0 -= carry ? tiles_y * TH : 0;
        t_.tn += gstep.tn + carry;
        return t_;
    };
    const int lrow = lane >> 3;
    const int src_k = ((lane & 7) ^ lrow) * 8;          // DMA lane -> (row = 8 piece + lrow, slot = lane & 7), source k-slot = slot ^ (row & 7)

    if (wave >= 10) {
        // ---------------------------------------------------------------- halo loader (waves 10, 11) ----------------------
        const int lh = wave - 10;
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
        // epilogue parameters of tile `t_` -> parameter buffer `buf` (wave 10 only): the noise of the TH x TW tile (one 16-B piece per lane, row-major),
        // the bias of the tile's 128 channels and the demodulation coefficients of (image, those channels) (lanes 0-31).  Issued during the tile's LAST
        // slice; this wave's vmcnt(0) in front of the next slice's first barrier covers them, and the tile's epilogue runs behind that barrier.
        auto issue_params = [&](int t_, int buf) {
            const TileC pc = decode(t_);
            unsigned char* dst = sP + buf * PARAM_BYTES;
            if (p.noise) {
                __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc((void*)p.noise, 0, 0x7fffffff, 0x00020000);
                constexpr int LPR = TW / 4;
                const int r = lane / LPR, cx = (lane - r * LPR) * 4;
                const unsigned off = (unsigned)(pc.tn * (int)p.noise_sn + (pc.y0 + r) * p.OW + pc.x0 + cx) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(nr, (lds_void_ptr)dst, 16, off, 0, 0, 0);
            }
            const int co = pc.c0 + 4 * lane;
            const unsigned okm = 0u - (unsigned)((lane < 32) & (co < p.Cout));
            if (p.bias) {
                __amdgpu_buffer_rsrc_t br_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(br_, (lds_void_ptr)(dst + 1024), 16, (((unsigned)co * 4u) & okm) | (SBG_OOB_OFFSET & ~okm), 0, 0, 0);
            }
            if (p.oscale) {
                __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)p.oscale, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (lds_void_ptr)(dst + 2048), 16, (((unsigned)(pc.tn * p.Cout + co) * 4u) & okm) | (SBG_OOB_OFFSET & ~okm), 0, 0, 0);
            }
        };
        // The loader's HPL pieces of a slice with everything per-lane worked out ONCE: rel[i] = byte offset of piece i's halo pixel (py, px) and k-slot from the halo's
        // origin, flag bits {py == 0, py == PH - 1, px == 0, px == PW - 1, no pixel} (5 per piece, 6 pieces per word).  Per slice the origin's byte offset
        // and the tile's edge mask are scalars, so an issue is ~5 vector instructions instead of ~25 (a division by PW, two range tests, three multiplies).
        // The loader waves share their SIMDs with the compute waves and were the last to arrive at every second barrier (scratch/halo_stamps.py):
        // +3 .. 4.5 % (scratch/kbench_ab.py).
        int rel[HPL]; unsigned ef[(HPL + 5) / 6];
#pragma unroll
        for (int w_ = 0; w_ < (HPL + 5) / 6; w_++) ef[w_] = 0;
#pragma unroll
        for (int i = 0; i < HPL; i++) {
            const int piece = lh * HPL + i, pp = piece * 8 + lrow;
            const int py = pp / PW, px = pp - py * PW;
            rel[i] = (py * (int)p.xs_h + px * (int)p.xs_w + src_k) * 2;
            const unsigned f = (unsigned)(py == 0) | ((unsigned)(py == PH - 1) << 1) | ((unsigned)(px == 0) << 2) | ((unsigned)(px == PW - 1) << 3)
                             | ((unsigned)((piece >= HPIECES) | (pp >= NPIX)) << 4);
            ef[i / 6] |= f << (5 * (i % 6));
        }
        auto issue_halo = [&](auto i_tag, int hbase, unsigned emask, bool kok, int buf) {
            constexpr int i = decltype(i_tag)::value;
            const int piece = lh * HPL + i;
            if (piece >= HPIECES) return;                // (wave-uniform)
            const bool bad = (((ef[i / 6] >> (5 * (i % 6))) & emask) != 0u) | !kok;
            const unsigned off = bad ? SBG_OOB_OFFSET : (unsigned)(rel[i] + hbase);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(sH + buf * HALO_BYTES + piece * 1024), 16, off, 0, 0, 0);
        };
        auto halo_base = [&](const TileC& t_, int chunk_) { return (t_.tn * (int)p.xs_n + (t_.y0 - 1) * (int)p.xs_h + (t_.x0 - 1) * (int)p.xs_w + chunk_ * 64) * 2; };
        auto edge_mask = [&](const TileC& t_) {
            const unsigned m = (unsigned)(t_.y0 == 0) | ((unsigned)(t_.y0 + TH == p.IH) << 1) | ((unsigned)(t_.x0 == 0) << 2) | ((unsigned)(t_.x0 + TW == p.IW) << 3) | 16u;
            return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
        };
        int ctile = bid, cchunk = 0, cpar = 0;          // tile / chunk of the slice being computed, parity of that tile's ordinal
        int tile = bid, chunk = 0;
        TileC tc = decode(tile);
        {
            const int hbase = halo_base(tc, 0); const unsigned emask = edge_mask(tc);
            static_for<HPL>([&](auto it_) { issue_halo(it_, hbase, emask, src_k < p.Cin, 0); });
        }
        // The 22 pieces of the next slice go out TWO per half-step (behind each of the slice's first eleven barriers), not as one burst behind
        // the first: an LDS-DMA instruction costs its wave ~100+ cycles beside the compute waves' fragment reads, every wave of the workgroup
        // meets at the next barrier, and a burst of 22 made that one half-step ~4x as long as the 32 MFMAs it should hide behind.  Measured
        // (scratch/kbench_ab.py, interleaved rounds on one device, [64, C, R, R] (*) [C, C, 3, 3]): C = 128 @ 256^2 1070 -> 1160 TF,
        // 256 @ 128^2 1154 -> 1300, 512 @ 64^2 1214 -> 1389, 512 @ 32^2 1273 -> 1390; three or four per half-step 1090-1110 / 1220-1250 /
        // 1280-1310.  (One piece behind each of the later barriers instead of two behind the first eleven: no difference.)
        constexpr int per = 2;                           // pieces per half-step
        for (int c = 0; c < nslices; c++) {
            wait_vmcnt_const<0>();                       // halo(c) has landed
            SBG_BARRIER();                // 18c
            const bool more = c + 1 < nslices && !(dbg & 2) && !(dbg & 32);       // 32: halo loads only
            if (more) {
                if (++chunk == kchunks) { chunk = 0; tile += G; tc = advance(tc); }
            }
            const int hbase = halo_base(tc, chunk); const unsigned emask = edge_mask(tc);
            const bool kok = chunk * 64 + src_k < p.Cin;
            static_for<17>([&](auto it_) {
                constexpr int i = decltype(it_)::value;
                if (more) {
                    if constexpr (per * i < HPL)     issue_halo(std::integral_constant<int, (per * i < HPL ? per * i : 0)>{}, hbase, emask, kok, (c + 1) & 1);
                    if constexpr (per * i + 1 < HPL) issue_halo(std::integral_constant<int, (per * i + 1 < HPL ? per * i + 1 : 0)>{}, hbase, emask, kok, (c + 1) & 1);
                }
                if (i == 12 && lh == 0 && p.lds_params && cchunk == kchunks - 1) issue_params(ctile, cpar);
                if (i == 16 && p.lds_params && c + 1 == nslices) wait_vmcnt_const<0>();      // the LAST tile's parameters have no later slice whose first barrier would cover them
                SBG_BARRIER();
            });
            if (++cchunk == kchunks) { cchunk = 0; ctile += G; cpar ^= 1; }
        }
        SBG_BARRIER();                    // 2S
        return;
    }
    if (wave >= 8) {
        // ---------------------------------------------------------------- weight loader (waves 8, 9) ----------------------
        const int lw = wave - 8;
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
        const int tl = lane < p.ntaps ? lane : 0;
        const int tbl_wtap = p.tap_slab[tl] * (int)p.ws_slab * 2;
        unsigned a_base[WPIECES];
        auto weight_rows = [&](int c0_) {
#pragma unroll
            for (int i = 0; i < WPIECES; i++) {
                const int co = c0_ + chmap((lw * WPIECES + i) * 8 + lrow);
                a_base[i] = (co < p.Cout) ? (unsigned)(co * (int)p.ws_co + src_k) * 2u : SBG_OOB_OFFSET;
            }
        };
        // (it, ichunk, itile, istage): coordinates of the step whose weights are issued next
        int it = 0, ichunk = 0, itile = bid, istage = 0;
        weight_rows(decode(itile).c0);
        unsigned char* st = nullptr; unsigned kokm = 0, wtap = 0;
        auto issue_prep = [&]() {
            st = smem + istage * STAGE + lw * WPIECES * 1024;
            kokm = 0u - (unsigned)(ichunk * 64 + src_k < p.Cin);
            wtap = (unsigned)(__builtin_amdgcn_readlane(tbl_wtap, it) + ichunk * 128);
        };
        // Whole 64-channel slabs (Cin % 64 == 0): the step's (tap, slab) offset is a scalar and rides in the instruction's scalar offset, so a piece
        // is the DMA and nothing else (a_base[i] stays 0x80000000 for a row past Cout: out of range with or without the scalar part).  The six vector
        // instructions per piece this removes (48 per step, in waves that share their SIMDs with the compute waves) were worth +7 .. 9 %:
        // [64, C, R, R] (*) [C, C, 3, 3], interleaved rounds: C = 128 @ 256^2 1187 -> 1270 TF, 256 @ 128^2 1321 -> 1439, 512 @ 64^2 1401 -> 1526.
        const bool whole_k = (p.Cin & 63) == 0;
        auto issue_part = [&](int lo, int hi) {
            if (whole_k) {
#pragma unroll
                for (int i = 0; i < WPIECES; i++)
                    if (i >= lo && i < hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + i * 1024), 16, a_base[i], wtap, 0, 0);
                return;
            }
#pragma unroll
            for (int i = 0; i < WPIECES; i++) {
                if (i < lo || i >= hi) continue;
                const unsigned okm = kokm & (0u - (unsigned)(a_base[i] != SBG_OOB_OFFSET));
                const unsigned off = ((a_base[i] + wtap) & okm) | (SBG_OOB_OFFSET & ~okm);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + i * 1024), 16, off, 0, 0, 0);
            }
        };
        auto issue_done = [&]() {
            istage = (istage + 1) & (NSTAGE - 1);
            if (++it == NT) {
                it = 0;
                if (++ichunk == kchunks) { ichunk = 0; itile += G; if (itile < ntiles) weight_rows(decode(itile).c0); }
            }
        };
        auto issue_next = [&]() { issue_prep(); issue_part(0, WPIECES); issue_done(); };
        constexpr bool split = false;                    // (half of a step's pieces behind each of its two barriers: +2..4 % alone, none beside the spread halo loads)
        const int S = nslices * NT;
        int issued = 0;                                  // steps issued so far
        for (; issued < 4 && issued < S; issued++) issue_next();
        for (int s = 0; s < S; s++) {
            // W(s) has landed once at most the steps issued after it are outstanding
            const int newer = issued - 1 - s;            // 0 .. 3
            if (newer >= 3) wait_vmcnt_const<3 * WPIECES>();
            else if (newer == 2) wait_vmcnt_const<2 * WPIECES>();
            else if (newer == 1) wait_vmcnt_const<1 * WPIECES>();
            else wait_vmcnt_const<0>();
            SBG_BARRIER();                // 2s
            const bool go = s >= 1 && issued < S && !(dbg & 2) && !(dbg & 16);      // step s + 3 -> the stage step s - 1 was read from (16: weight loads only)
            if (go) { issue_prep(); issue_part(0, split ? WPIECES / 2 : WPIECES); }
            SBG_BARRIER();                // 2s + 1
            if (go) { if (split) issue_part(WPIECES / 2, WPIECES); issue_done(); issued++; }
            else if (s >= 1 && issued < S) issued++;
        }
        SBG_BARRIER();                    // 2S
        return;
    }

    // -------------------------------------------------------------------- compute waves (0-7) ----------------------------
    const bool grpY = wave >= 4;
    const int wci = wave >> 2, wpi = wave & 3;           // waves 0-3 and 4-7 each cover both channel halves? no: wave = wci * 4 + wpi
    const int wc = wci * WC;
    const int fr = lane & 15, fg = lane >> 4;
    const int frag_off = fr * 128 + ((fg ^ (fr & 7)) << 4);          // k-sub 0; k-sub 1 = ^ 64
    const int tl = lane < p.ntaps ? lane : 0;
    const int tbl_shift = p.tap_dy[tl] * PW + p.tap_dx[tl];
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    short8_t fa[2][TC], fb[2][TP];
    int seg_pp[TP];                                      // patch pixel of (segment j, lane fr) for tap (0, 0)
#pragma unroll
    for (int j = 0; j < TP; j++) {
        const int sg = wpi * TP + j, r = sg / SEG, cseg = (sg - r * SEG) * 16;
        seg_pp[j] = (r + 1) * PW + cseg + fr + 1;
    }
    int stage = 0, par = 0, chunk = 0, tile = bid;
    TileC cur = decode(tile);
    // The epilogue of a finished tile is deferred into the first step of the next tile, behind that step's B_a: there it runs
    // beside the partner group's MFMA phase instead of holding both groups at a barrier (the stores are the exposed part of a
    // short-K tile: 18 K-steps at Cin = 128).
    bool pend = false;
    TileC done = cur;
    int tpar = 0, done_par = 0;                          // parity of the current / finished tile's ordinal: its parameter buffer
    auto epilogue = [&](const TileC& tc) __attribute__((always_inline)) {
        const LdsParams lp{(lds_cfloat*)(sP + done_par * PARAM_BYTES), wc, TW};
        if (!(dbg & 8))
        conv_epilogue8<TC, TP, true>(p, acc, tc.c0 + wc, fg, [&](int j, int& n, int& oy, int& ox) {
            const int sg = wpi * TP + j, r = sg / SEG, cseg = (sg - r * SEG) * 16;
            n = tc.tn; oy = tc.y0 + r; ox = tc.x0 + cseg + fr;
            return true;
        }, 0, p.lds_params ? &lp : nullptr, tc.y0, tc.x0);
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    };
    if (grpY) SBG_BARRIER();              // 0
    for (int c = 0; c < nslices; c++) {
        const unsigned char* hb = sH + par * HALO_BYTES;
        static_for<NT>([&](auto tap_tag) {
            constexpr int t = decltype(tap_tag)::value;
            const int shift = __builtin_amdgcn_readlane(tbl_shift, t);
            const unsigned char* sa = smem + stage * STAGE + wc * 128;
            SBG_BARRIER();                // B_a: the stage and the halo buffer of this step have landed
            if (t == 0 && pend) { epilogue(done); pend = false; }
            if (!(dbg & 4)) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TC; i++) fa[ks][i] = *reinterpret_cast<const short8_t*>(sa + i * 16 * 128 + (frag_off ^ (ks * 64)));
#pragma unroll
            for (int j = 0; j < TP; j++) {
                const int pp = seg_pp[j] + shift;
                const int o = pp * 128 + ((fg ^ (pp & 7)) << 4);
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fb[ks][j] = *reinterpret_cast<const short8_t*>(hb + (o ^ (ks * 64)));
            }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): fragments are in registers, this wave no longer reads the stage
            SBG_BARRIER();                // B_b
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 1))
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int i = 0; i < TC; i++)
#pragma unroll
                    for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[ks][i], fb[ks][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            stage = (stage + 1) & (NSTAGE - 1);
        });
        par ^= 1;
        if (++chunk < kchunks) continue;
        done = cur; done_par = tpar; tpar ^= 1;
        // X defers its epilogue behind the next step's B_a (Y is then in its last MFMA phase).  With a fused tail (its arithmetic makes the epilogue
        // ~40 % longer) Y runs its own right away, behind its last MFMA, in the interval in which X runs X's: the two groups' tail arithmetic and
        // stores side by side take less than one after the other (deferred like X's, Y's epilogue filled a second ~3850-cycle interval while X
        // waited at the next barrier; scratch/halo_stamps.py).  Measured, interleaved rounds: with tail +3 / +1.7 / +1 % (128 / 256 / 512 channels),
        // plain stores -1 % -- a plain launch keeps both deferred.
        if (grpY && p.lds_params) epilogue(done); else pend = true;
        chunk = 0; tile += G;
        if (tile < ntiles) cur = advance(cur);
    }
    if (pend) epilogue(done);
    if (!grpY) SBG_BARRIER();             // 2S
}

// ---------------------------------------------------------------------------------------------------------------------
#undef SBG_BARRIER
// Gather kernel with loader waves: the 128 x 256 tile of conv_k64_kernel in the persistent 12-wave structure of
// conv_halo_ld_kernel.  Waves 8-9 stream the weight tile, waves 10-11 gather the 256 pixel rows of the tap (im2col on the
// fly, out-of-image rows at an out-of-range offset); three 48 KB stages, the loads of step s + 2 are issued while step s
// computes.  Tiles bid, bid + G, ... run through one pipeline, the epilogue of a tile is deferred into the next tile's first
// step.  Transposed-convolution phases and strided convolutions have 2-36 K-steps per tile, so the per-tile prologue /
// epilogue of the one-tile-per-workgroup kernel was a third of their time.
// Barrier indices as in conv_halo_ld_kernel; every loader: per step { wait step s landed; 2s; issue step s + 2 into the stage
// step s - 1 used (its reads retired before 2s); 2s + 1 }, final 2S.
template <class MF>
__global__ __launch_bounds__(768, 3) void conv_gather_ld_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int BC = 128, BP = 256, WGP = 4;
    constexpr int WC = 64, WP = 64, TC = 4, TP = 4;
    constexpr int NSTAGE = 3, A_BYTES = BC * 128, B_BYTES = BP * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WPIECES = BC / 8 / 2;                // weight pieces per weight loader per step (8)
    constexpr int XPIECES = BP / 8 / 2;                // pixel pieces per pixel loader per step (16)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware tile order: workgroups b and b + 8 share an XCD (L2); give each XCD a contiguous run of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ntiles = p.ptiles * p.ctiles, G = gridDim.x;
    const int my_tiles = (ntiles - bid + G - 1) / G;                  // tiles bid, bid + G, ...  (the grid never exceeds ntiles)
    const int kchunks = (p.Cin + 63) >> 6;
    // tile -> (channel tile, phase, pixel tile): the phases of one pixel tile are neighbours in the tile order, so they run at the same
    // time on CUs of one XCD and share the input rows through its L2.  K-steps of a tile, tap-major: step = t * kchunks + chunk.
    // The phases have different K lengths (4 / 2 / 2 / 1 taps), and tile + G keeps r % nphase, so the phase is rotated with the pixel
    // tile: every workgroup then walks all phases in turn (equal work) -- still a bijection on (pixel tile, phase).
    auto tile_phase = [&](int tile) { const int r = tile / p.ctiles; return (r % p.nphase + (r / p.nphase) / p.ph_rot_div) % p.nphase; };
    int S = 0;
    for (int t_ = bid; t_ < ntiles; t_ += G) S += p.ph_ntaps[tile_phase(t_)] * kchunks;
    const int lrow = lane >> 3;
    const int src_k = ((lane & 7) ^ lrow) * 8;          // DMA lane -> (row = 8 piece + lrow, slot = lane & 7), source k-slot = slot ^ (row & 7)
    const int tl = lane < p.ntaps ? lane : 0;

    if (wave >= 8) {
        // ---------------------------------------------------------------- loader waves -----------------------------------
        const bool xload = wave >= 10;                  // waves 10, 11 gather pixels; 8, 9 stream weights
        const int lw = (wave - 8) & 1;
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
        const int tbl_dy = p.tap_dy[tl], tbl_dx = p.tap_dx[tl];
        const int tbl_wtap = p.tap_slab[tl] * (int)p.ws_slab * 2;
        unsigned a_base[WPIECES];
        int b_iy0[XPIECES], b_ix0[XPIECES]; unsigned b_base[XPIECES];
        int ph_t0 = 0, ph_nt = 1;                       // taps of the issue tile's phase
        auto tile_state = [&](int tile) {
            const int ph = tile_phase(tile);
            const int c0 = (tile % p.ctiles) * BC, p0 = (tile / p.ctiles / p.nphase) * BP;
            const int OW_ = p.ph_OW[ph], OH_ = p.ph_OH[ph], P_ = p.ph_P[ph];
            ph_t0 = p.ph_tap0[ph]; ph_nt = p.ph_ntaps[ph];
            if (!xload) {
#pragma unroll
                for (int i = 0; i < WPIECES; i++) {
                    const int co = c0 + chmap((lw * WPIECES + i) * 8 + lrow);
                    a_base[i] = (co < p.Cout) ? (unsigned)(co * (int)p.ws_co + src_k) * 2u : SBG_OOB_OFFSET;
                }
            } else {
#pragma unroll
                for (int i = 0; i < XPIECES; i++) {
                    const int pix = p0 + (lw * XPIECES + i) * 8 + lrow;
                    const bool ok = pix < P_;
                    const int pp = ok ? pix : 0;
                    const int ox = pp % OW_, t = pp / OW_, oy = t % OH_, n = t / OH_;
                    b_iy0[i] = ok ? oy * p.stride : -(1 << 28);            // invalid rows fail every range test below
                    b_ix0[i] = ox * p.stride;
                    b_base[i] = (unsigned)(n * (int)p.xs_n + src_k) * 2u;
                }
            }
        };
        // (it, ic, itile, istage): coordinates of the step whose loads are issued next
        int it = 0, ic = 0, itile = bid, istage = 0;
        tile_state(itile);
        // the loads of one step in two parts (half = 0, 1; half < 0: everything): the two halves go behind the step's two barriers
        auto issue_part = [&](int half) {
            const unsigned kokm = 0u - (unsigned)(ic * 64 + src_k < p.Cin);
            if (!xload) {
                unsigned char* st = smem + istage * STAGE + lw * WPIECES * 1024;
                const unsigned wtap = (unsigned)(__builtin_amdgcn_readlane(tbl_wtap, ph_t0 + it) + ic * 128);
#pragma unroll
                for (int i = 0; i < WPIECES; i++) {
                    if (half >= 0 && (i >= WPIECES / 2) != (half == 1)) continue;
                    const unsigned okm = kokm & (0u - (unsigned)(a_base[i] != SBG_OOB_OFFSET));      // branch-free: a masked-off lane would leave stale LDS bytes
                    const unsigned off = ((a_base[i] + wtap) & okm) | (SBG_OOB_OFFSET & ~okm);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + i * 1024), 16, off, 0, 0, 0);
                }
            } else {
                unsigned char* st = smem + istage * STAGE + A_BYTES + lw * XPIECES * 1024;
                const int dy = __builtin_amdgcn_readlane(tbl_dy, ph_t0 + it), dx = __builtin_amdgcn_readlane(tbl_dx, ph_t0 + it);
#pragma unroll
                for (int i = 0; i < XPIECES; i++) {
                    if (half >= 0 && (i >= XPIECES / 2) != (half == 1)) continue;
                    const int iy = b_iy0[i] + dy, ix = b_ix0[i] + dx;
                    const unsigned okm = kokm & (0u - (unsigned)(((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW)));
                    const unsigned real = b_base[i] + (unsigned)(iy * (int)p.xs_h + ix * (int)p.xs_w + ic * 64) * 2u;
                    const unsigned off = (real & okm) | (SBG_OOB_OFFSET & ~okm);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(st + i * 1024), 16, off, 0, 0, 0);
                }
            }
        };
        auto issue_done = [&]() {
            istage = istage == NSTAGE - 1 ? 0 : istage + 1;
            if (++ic == kchunks) {
                ic = 0;
                if (++it == ph_nt) { it = 0; itile += G; if (itile < ntiles) tile_state(itile); }
            }
        };
        auto issue_next = [&]() { issue_part(-1); issue_done(); };
        constexpr bool split = true;                     // half of a step's pieces behind each of its two barriers (+0.5..1 %)
        int issued = 0;                                  // steps issued so far
        for (; issued < NSTAGE && issued < S; issued++) issue_next();
        for (int s = 0; s < S; s++) {
            // step s has landed once at most the steps issued after it are outstanding (same instruction count per step)
            const int newer = issued - 1 - s;            // 0 .. 2
            if (xload) {
                if (newer >= 2) wait_vmcnt_const<2 * XPIECES>(); else if (newer == 1) wait_vmcnt_const<XPIECES>(); else wait_vmcnt_const<0>();
            } else {
                if (newer >= 2) wait_vmcnt_const<2 * WPIECES>(); else if (newer == 1) wait_vmcnt_const<WPIECES>(); else wait_vmcnt_const<0>();
            }
            __builtin_amdgcn_s_barrier();                // 2s
            const bool go = s >= 1 && issued < S;        // step s + 2 -> the stage step s - 1 was read from
            if (go) issue_part(split ? 0 : -1);
            __builtin_amdgcn_s_barrier();                // 2s + 1
            if (go) { if (split) issue_part(1); issue_done(); issued++; }
        }
        __builtin_amdgcn_s_barrier();                    // 2S
        return;
    }

    // -------------------------------------------------------------------- compute waves (0-7) ----------------------------
    const bool grpY = wave >= 4;
    const int wci = wave >> 2, wpi = wave & 3;
    const int wc = wci * WC, wp = wpi * WP;
    const int fr = lane & 15, fg = lane >> 4;
    const int frag_off = fr * 128 + ((fg ^ (fr & 7)) << 4);          // k-sub 0; k-sub 1 = ^ 64
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    short8_t fa[2][TC], fb[2][TP];
    int stage = 0, tile = bid, left = p.ph_ntaps[tile_phase(bid)] * kchunks;      // steps left in the tile being computed
    bool pend = false;
    int done_tile = tile;
    auto epilogue = [&](int t_) __attribute__((always_inline)) {
        const int ph = tile_phase(t_);
        const int c0 = (t_ % p.ctiles) * BC, p0 = (t_ / p.ctiles / p.nphase) * BP;
        const int OW_ = p.ph_OW[ph], OH_ = p.ph_OH[ph], P_ = p.ph_P[ph];
        conv_epilogue8<TC, TP>(p, acc, c0 + wc, fg, [&](int j, int& n, int& oy, int& ox) {
            const int pix = p0 + wp + 16 * j + fr;
            const int pp = pix < P_ ? pix : 0;
            ox = pp % OW_; const int t = pp / OW_; oy = t % OH_; n = t / OH_;
            return pix < P_;
        }, p.ph_yoff[ph]);
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    };
    if (grpY) __builtin_amdgcn_s_barrier();              // 0
    for (int s = 0; s < S; s++) {
        const unsigned char* sa = smem + stage * STAGE + wc * 128;
        const unsigned char* sb = smem + stage * STAGE + A_BYTES + wp * 128;
        __builtin_amdgcn_s_barrier();                    // B_a: the stage of this step has landed
        if (pend) { epilogue(done_tile); pend = false; } // previous tile's stores run beside the partner group's MFMA phase
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TC; i++) fa[ks][i] = *reinterpret_cast<const short8_t*>(sa + i * 16 * 128 + (frag_off ^ (ks * 64)));
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < TP; j++) fb[ks][j] = *reinterpret_cast<const short8_t*>(sb + j * 16 * 128 + (frag_off ^ (ks * 64)));
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): fragments are in registers, this wave no longer reads the stage
        __builtin_amdgcn_s_barrier();                    // B_b
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < TC; i++)
#pragma unroll
                for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[ks][i], fb[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        if (--left == 0) { pend = true; done_tile = tile; tile += G; if (tile < ntiles) left = p.ph_ntaps[tile_phase(tile)] * kchunks; }
    }
    if (pend) epilogue(done_tile);
    if (!grpY) __builtin_amdgcn_s_barrier();             // 2S
}

template <class MF>
static int launch_gather_ld(ConvArgs& a, unsigned x_bytes, unsigned w_bytes, hipStream_t stream)
{
    constexpr int lds = 3 * (128 * 128 + 256 * 128);
    a.ctiles = (a.Cout + 127) / 128;
    if (a.nphase <= 1) {            // the whole launch as one phase
        a.nphase = 1; a.ph_tap0[0] = 0; a.ph_ntaps[0] = a.ntaps; a.ph_OH[0] = a.OH; a.ph_OW[0] = a.OW; a.ph_P[0] = a.P; a.ph_yoff[0] = 0;
    }
    int pmax = 0;
    for (int i = 0; i < a.nphase; i++) if (a.ph_P[i] > pmax) pmax = a.ph_P[i];
    a.ptiles = ((pmax + 255) / 256) * a.nphase;      // every phase gets the pixel tiles of the largest one (the others' last tile may be empty)
    int64_t nblk = (int64_t)a.ptiles * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        ncu = n;
    }
    if (nblk > ncu) nblk = ncu;     // persistent: one workgroup per CU, each walks tiles b, b + grid, ...
    a.ph_rot_div = (int)(nblk / ((int64_t)a.ctiles * a.nphase)); if (a.ph_rot_div < 1) a.ph_rot_div = 1;
    auto kern = conv_gather_ld_kernel<MF>;
    if (!SBG_RAISE_LDS_ONCE(kern, lds))
        return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds);
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    double macs = 0.0, outpix = 0.0;
    for (int i = 0; i < a.nphase; i++) { macs += (double)a.ph_P[i] * a.ph_ntaps[i]; outpix += a.ph_P[i]; }
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * macs * a.Cout * (double)a.Cin,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * outpix * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {(int)outpix, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, 4128256 + (a.nphase > 1 ? 1000000 * a.nphase : 0)});
    SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(768), lds, stream, a, x_bytes, w_bytes);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

template <class MF, int TH, int TW>
static int launch_halo_ld(ConvArgs& a, unsigned x_bytes, unsigned w_bytes, hipStream_t stream)
{
    constexpr int HPIECES = ((TH + 2) * (TW + 2) + 7) / 8;
    constexpr int lds = 4 * 128 * 128 + 2 * HPIECES * 1024 + 2 * 3072;
    {   // the epilogue parameters go through LDS when the epilogue's 16-B fast path applies and the noise rows are 16-B aligned
        const bool plain = (a.act <= SBG_ACT_LINEAR) && a.gain == 1.f && a.clamp < 0.f && !a.bias && !a.noise && !a.oscale;
        const bool fast = ((a.Cout & 7) == 0) && ((((uintptr_t)a.y) & 15) == 0) && (((a.ys_n | a.ys_h | a.ys_w) & 7) == 0)
                          && ((((uintptr_t)a.oscale) & 15) == 0) && ((((uintptr_t)a.bias) & 15) == 0) && a.ydtype != SBG_F16;
        const bool nz_ok = !a.noise || (((((uintptr_t)a.noise) & 15) == 0) && (a.noise_sn & 3) == 0 && (a.OW & 3) == 0
                                        && (int64_t)a.N * (a.noise_sn > 0 ? a.noise_sn : 0) + (int64_t)a.OH * a.OW < (1ll << 28));
        a.lds_params = !plain && fast && nz_ok && (int64_t)a.N * a.Cout < (1ll << 28) && !((a.debug >> 8) & 32);      // experiment bit 32: parameters from global memory
    }
    static_assert(lds <= 160 * 1024, "LDS budget");
    a.ctiles = (a.Cout + 127) / 128;
    a.ptiles = a.N * (a.OH / TH) * (a.OW / TW);
    int64_t nblk = (int64_t)a.ptiles * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        ncu = n;
    }
    if (nblk > ncu) nblk = ncu;     // persistent: one workgroup per CU (the LDS footprint admits no more), each walks tiles b, b + grid, ...
    auto kern = conv_halo_ld_kernel<MF, TH, TW>;
    if (!SBG_RAISE_LDS_ONCE(kern, lds))
        return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds);
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * a.P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * a.P * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {a.P, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, 3128256});
    SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(768), lds, stream, a, x_bytes, w_bytes);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

template <class MF, int BC, int BP, int WGC, int WGP>
static int launch_k64(ConvArgs& a, unsigned x_bytes, unsigned w_bytes, hipStream_t stream)
{
    constexpr int lds = 3 * (BC * 128 + BP * 128);
    static_assert(lds <= 160 * 1024, "LDS budget");
    a.ctiles = (a.Cout + BC - 1) / BC;
    a.ptiles = (a.P + BP - 1) / BP;
    const int64_t nblk = (int64_t)a.ptiles * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    auto kern = conv_k64_kernel<MF, BC, BP, WGC, WGP>;
    if (!SBG_RAISE_LDS_ONCE(kern, lds))
        return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds);
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * a.P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * a.P * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {a.P, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, 1000000 + BC * 1000 + BP});
    SBG_LAUNCH(kern, dim3((unsigned)nblk, (unsigned)(a.ksplit > 1 ? a.ksplit : 1)), dim3(512), lds, stream, a, x_bytes, w_bytes);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

// y[i] (+)= sum_k ws[k][i] in a fixed order; 16 lanes share an output element (slabs k, k + 16, ...), then a shuffle tree.
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(const float* ws, float* y, int64_t n, int ksplit, int accumulate)
{
    const int sub = threadIdx.x & 15;
    const int64_t step = (int64_t)gridDim.x * 16;
    for (int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < n; i += step) {
        float s = 0.f;
        for (int k = sub; k < ksplit; k += 16) s += ws[(int64_t)k * n + i];
        for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (sub == 0) y[i] = accumulate ? y[i] + s : s;
    }
}

// Slab reduction with the fused epilogue: y[p][co] = clamp(act(sum_k ws[k][p][co] * oscale[n, co] + noise[n, pixel] + bias[co]) * gain), any output
// dtype, dense channel-minor y.  Lets the few-tile / long-K launches of the 16-bit 4x4 .. 8x8 blocks (16-64 workgroups for 72 K-steps) split K as
// well: 8 channels per lane, 16-B loads per slab.
__global__ __launch_bounds__(256) void conv_ksplit_reduce_epi_kernel(const float* ws, ConvArgs p, int64_t n, int ksplit)
{
    const int64_t groups = n >> 3;                      // n = P * Cout, Cout % 8 == 0
    const int hw = p.OH * p.OW;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        const int64_t i = g << 3;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ksplit; k++) {
            float v[8];
            Vec8<float>::ld(ws + (int64_t)k * n + i, v);
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += v[e];
        }
        const int64_t pix = i / p.Cout; const int co = (int)(i - pix * p.Cout);
        const int img = (int)(pix / hw), pp = (int)(pix - (int64_t)img * hw);
        const float nz = p.noise ? p.noise[(int64_t)img * p.noise_sn + pp] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float u = acc[e];
            if (p.oscale) u *= p.oscale[(int64_t)img * p.Cout + co + e];
            u += nz + (p.bias ? p.bias[co + e] : 0.f);
            if (p.act == SBG_ACT_LRELU) u = (u > 0.f) ? u : u * p.alpha;
            else if (p.act == SBG_ACT_RELU) u = (u > 0.f) ? u : 0.f;
            u *= p.gain;
            if (p.clamp >= 0.f) u = (u > -p.clamp && u < p.clamp) ? u : (u >= 0.f ? p.clamp : -p.clamp);
            acc[e] = u;
        }
        if (p.ydtype == SBG_F32)       Vec8<float>::st((float*)p.y + i, acc);
        else if (p.ydtype == SBG_BF16) Vec8<bf16_s>::st((bf16_s*)p.y + i, acc);
        else                           Vec8<f16_s>::st((f16_s*)p.y + i, acc);
    }
}

template <class MF>
static int dispatch_k64(ConvArgs& a, int level, unsigned xb, unsigned wb, hipStream_t stream)
{
    if (a.nphase > 1) return launch_gather_ld<MF>(a, xb, wb, stream);       // multi-phase launches exist only in the persistent gather kernel
    // halo kernel: stride 1, nine taps with |offset| <= 1, output grid == input grid, tile-aligned, enough tiles to fill the chip
    bool halo = level >= 2 && a.stride == 1 && a.ntaps == 9 && a.OH == a.IH && a.OW == a.IW;
    for (int t = 0; halo && t < 9; t++) halo = a.tap_dy[t] >= -1 && a.tap_dy[t] <= 1 && a.tap_dx[t] >= -1 && a.tap_dx[t] <= 1;
    const int64_t tiles256 = (int64_t)((a.P + 255) / 256) * ((a.Cout + 127) / 128);
    if (halo && a.Cout > 64 && tiles256 >= 256) {
        const int rc8 = sbg_conv_halo8_dispatch(a, std::is_same<MF, bf16_mfma>::value, xb, wb, stream);
        if (rc8 != -1) return rc8;
        if (a.OW % 32 == 0 && a.OH % 8 == 0)  return launch_halo_ld<MF, 8, 32>(a, xb, wb, stream);
        if (a.OW % 16 == 0 && a.OH % 16 == 0) return launch_halo_ld<MF, 16, 16>(a, xb, wb, stream);
    }
    if (a.Cout <= 64) return launch_k64<MF, 64, 256, 1, 8>(a, xb, wb, stream);
    if (a.ksplit > 1) return launch_k64<MF, 128, 128, 2, 4>(a, xb, wb, stream);        // the caller wraps this launch with the slab reduction
    if (tiles256 < 256) return launch_k64<MF, 128, 128, 2, 4>(a, xb, wb, stream);
    // short reductions (transposed-conv phases: 1-4 taps) gain from the persistent pipeline; measured: +4..10 % at <= 8 K-steps per tile,
    // -5 % at 18+ (the 8-wave kernel's in-wave DMA issue overlaps better there)
    static const char* egl = sbg_env("SBG_K64_GATHER_LD");        // experiment switch: 0 never, 1 always
    const int ksteps = a.ntaps * ((a.Cin + 63) >> 6);
    if (egl ? atoi(egl) != 0 : (ksteps >= 2 && ksteps <= 8)) return launch_gather_ld<MF>(a, xb, wb, stream);      // 1 step: store-bound, the plain kernel wins
    return launch_k64<MF, 128, 256, 2, 4>(a, xb, wb, stream);
}

} // namespace

// Split-K is worth it when the output tiles alone leave most CUs idle and the reduction is long: returns the split actually used.
static int plan_ksplit(const ConvArgs& a, int ksplit)
{
    if (ksplit <= 1 || a.Cout <= 64) return 1;
    const int nsteps = a.ntaps * ((a.Cin + 63) >> 6);
    int k = ksplit;
    if (k > nsteps / 4) k = nsteps / 4;             // at least four K-steps per workgroup
    return k < 2 ? 1 : k;
}

extern "C" int64_t sbg_conv2d_igemm_workspace(const sbg_conv_params* q)
{
    if (!q || q->ksplit <= 1) return 0;
    return (int64_t)q->ksplit * q->N * q->OH * q->OW * q->Cout * (int64_t)sizeof(float);
}

int sbg_conv_k64_dispatch(ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, void* workspace, int ksplit, hipStream_t stream)
{
    if (x_bytes >= (int64_t)SBG_OOB_OFFSET || w_bytes >= (int64_t)SBG_OOB_OFFSET) return -1;
    if (a.xs_n < 0 || a.xs_h < 0 || a.xs_w < 0 || a.ws_slab < 0 || a.ws_co < 0) return -1;
    // experiment switch: SBG_CONV_K64 = 0 off / 1 gather only / 2 (default) gather + halo
    const char* e1 = sbg_env("SBG_CONV_K64");
    const int level = e1 ? atoi(e1) : 2;
    if (level <= 0) return -1;
#ifdef SBG_K64_DEBUG
    { const char* e3 = sbg_env("SBG_K64_ABL"); a.debug = (a.debug & ~255) | (e3 ? atoi(e3) & 255 : 0); }
#endif
    a.debug = (a.debug & 255) | (sbg_experiment() << 8);      // variants under A/B test (sbg_experiment_set)
    const unsigned xb = (unsigned)x_bytes, wb = (unsigned)w_bytes;
    const int64_t y_numel = (int64_t)a.P * a.Cout;
    const bool dense_y = a.ys_w == a.Cout && a.ys_h == (int64_t)a.OW * a.Cout && a.ys_n == (int64_t)a.OH * a.OW * a.Cout;
    const bool plain = (a.act <= SBG_ACT_LINEAR) && a.gain == 1.f && a.clamp < 0.f && !a.bias && !a.noise && !a.oscale;
    const bool simple = a.ydtype == SBG_F32 && plain;                      // plain fp32 sum (may accumulate); otherwise the epilogue rides in the reduction
    const bool epi_ok = !a.accumulate && (a.Cout & 7) == 0 && (((uintptr_t)a.y) & 15) == 0;
    const int k = (workspace && dense_y && (simple || epi_ok)) ? plan_ksplit(a, ksplit) : 1;
    if (k > 1) {
        ConvArgs b = a;
        void* y = a.y; const int acc = a.accumulate;
        b.y = workspace; b.accumulate = 0; b.ksplit = k; b.y_split_stride = y_numel;
        b.ydtype = SBG_F32; b.act = SBG_ACT_LINEAR; b.gain = 1.f; b.clamp = -1.f; b.bias = nullptr; b.noise = nullptr; b.oscale = nullptr;    // raw fp32 slabs
        const int rc = bf16 ? dispatch_k64<bf16_mfma>(b, level, xb, wb, stream) : dispatch_k64<f16_mfma>(b, level, xb, wb, stream);
        if (rc != SBG_OK) return rc;
        if (simple) {
            unsigned grid = (unsigned)((y_numel + 15) / 16); if (grid > 4096) grid = 4096;
            SBG_LAUNCH(conv_ksplit_reduce_kernel, dim3(grid), dim3(256), 0, stream, (const float*)workspace, (float*)y, y_numel, k, acc);
        } else {
            unsigned grid = (unsigned)((y_numel / 8 + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
            SBG_LAUNCH(conv_ksplit_reduce_epi_kernel, dim3(grid), dim3(256), 0, stream, (const float*)workspace, a, y_numel, k);
        }
        SBG_HIP_LAUNCH_CHECK();
        return SBG_OK;
    }
    if (bf16) return dispatch_k64<bf16_mfma>(a, level, xb, wb, stream);
    return dispatch_k64<f16_mfma>(a, level, xb, wb, stream);
}
