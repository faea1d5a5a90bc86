// conv_up2.hip -- all four phases of a stride-2 3x3 transposed convolution in ONE pass over the input (gfx950).
//
// Replaces, for the up-sampling convolutions of the generator and the data gradients of the discriminator's strided convolutions, the
// multi-phase launch of conv_gather_ld_kernel (conv_k64.hip), which treats every (phase, tap) as its own gathered 256-pixel tile:
// 48 KB staged per 64-deep K-step, 11.4 B/kflop, 590-690 TFLOP/s.  The reference computes the same thing with cuDNN's
// conv_transpose2d (torch_utils/ops/conv2d_resample.py:117-131 -> conv2d_gradfix.py:36-48).
//
// Phase (a, b) of y = conv_transpose2d(x, w, stride 2) is a small convolution of x on the phase grid: y[2 oy + a', 2 ox + b'] =
// sum over the phase's taps of w_tap . x[oy + dy, ox + dx] with dy, dx in a 2 x 2 window that is THE SAME for all phases, and the nine taps
// split 4 / 2 / 2 / 1 over the phases.  So a workgroup stages the (8 + 1) x (32 + 1) input halo of an 8 x 32 tile of the phase grid once
// per 64-channel slice (38 KB) and runs all nine tap products from it, each into the accumulators of its phase: 64 output channels x 256
// pixels x 4 phases per workgroup.  Per slice: 38 KB of halo + 72 KB of weights for 18.9 MFLOP = 5.8 B/kflop.
//
// Structure (the one the weight-gradient rows kernel settled on): 8 waves, every wave loads and computes; LDS-DMA (`buffer_load ... lds`)
// fills a ring of four 16 KB weight stages (two taps each) three steps ahead and the other halo buffer during the first three steps of a
// slice; fragment reads are inline assembly with counted waits (lds_asm.h) so the compiler does not guard them with vmcnt(0); one barrier per
// step, five steps (2 + 2 + 2 + 2 + 1 taps) per slice.  Every wave issues the same number of DMA instructions in a given step (spare slots
// write zeros into a dump KiB), so every vmcnt count is an immediate:  step k of a slice issues [2 halo pieces if k < 3][2 weight pieces];
// the weights of step s were the last thing step s - 3 issued, so they (and every halo piece before them) have landed once at most
// cnt(s - 2) + cnt(s - 1) loads are outstanding.
//
// Accumulator layout and channel permutation of the weight rows are those of conv_k64.hip, so its epilogue is reused per phase.
#include "conv_common.h"
#include "lds_asm.h"
#include <cstdlib>
#include <utility>

using namespace sbgconv;

// from conv_k64.hip (same translation-unit-local helpers, restated: 30 lines are cheaper than a shared header with the epilogue's templates)
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define SBG_OOB_OFFSET 0x80000000u

static __device__ __forceinline__ int chmap(int R) { return (R & ~31) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3); }

template <int OFF>
static __device__ __forceinline__ void lds_read128_issue(short8_t& d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
static __device__ __forceinline__ void lds_wait6(short8_t& a0, short8_t& a1, short8_t& b0, short8_t& b1, short8_t& b2, short8_t& b3)
{
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "n"(N) : "memory");
}

// store of one phase: lane (fr, fg) holds, for the pixel of fragment column fr in segment j, channels cbase + 8 fg + e with e = 0..3 in
// acc[0][j] and e = 4..7 in acc[1][j] (conv_k64.hip's layout with TC = 2)
template <int YDT>
static __device__ __forceinline__ void store_phase(const ConvArgs& p, float4_t (&acc)[2][4], int cbase, int fg, int n, int oy0, int ox0, int pg, int fr,
                                                   int OHp, int OWp, int64_t ybase)
{
    if (cbase + 8 * fg >= p.Cout) return;                // (Cout is a multiple of 8: a lane's eight channels are in or out together)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int oy = oy0 + pg * 2 + (j >> 1), ox = ox0 + (j & 1) * 16 + fr;
        if (oy >= OHp || ox >= OWp) continue;
        const int64_t yoff = ybase + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w + cbase + 8 * fg;
        float4_t lo = acc[0][j], hi = acc[1][j];
        if (YDT == SBG_F32) {
            float* dst = (float*)p.y + yoff;
            if (p.accumulate) { lo += *reinterpret_cast<float4_t*>(dst); hi += *reinterpret_cast<float4_t*>(dst + 4); }
            *reinterpret_cast<float4_t*>(dst) = lo;
            *reinterpret_cast<float4_t*>(dst + 4) = hi;
        } else {
            short8_t o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o[e]     = (short)(YDT == SBG_BF16 ? f32_to_bf16_bits(lo[e]) : f32_to_f16_bits(lo[e]));
                o[4 + e] = (short)(YDT == SBG_BF16 ? f32_to_bf16_bits(hi[e]) : f32_to_f16_bits(hi[e]));
            }
            *reinterpret_cast<short8_t*>((unsigned short*)p.y + yoff) = o;
        }
    }
}

// grid: (N * tiles_y * tiles_x) * ctiles workgroups, channel tile fastest (the channel tiles of one pixel tile share its halo in L2)
template <class MF, int YDT>
__global__ __launch_bounds__(512) void conv_up2_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes, int tiles_y, int tiles_x, int dymin, int dxmin)
{
    constexpr int TH = 8, TW = 32, PW = TW + 1, PH = TH + 1, NPIX = PW * PH;       // 297 halo pixels
    constexpr int HPIECES = (NPIX + 7) / 8, HALO = HPIECES * 1024;                  // 38 pieces of 8 pixels x 128 B
    constexpr int WSTAGE = 2 * 64 * 128, NRING = 4, LEAD = 3;                       // a weight stage = two taps x 64 rows x 128 B
    constexpr int RING0 = 2 * HALO, DUMP = RING0 + NRING * WSTAGE;                   // LDS: halo 0 | halo 1 | ring | dump KiB
    constexpr int NSTEP = 5;                                                         // steps per slice: taps {0,1} {2,3} {4,5} {6,7} {8}
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int ct = b % p.ctiles; b /= p.ctiles;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int c0 = ct * 64, oy0 = ty * TH, ox0 = tx * TW;
    const int kchunks = (p.Cin + 63) >> 6;
    const int S = kchunks * NSTEP;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
    const int lrow = lane >> 3;
    const int src_k = ((lane & 7) ^ lrow) * 8;          // DMA lane -> (row = 8 piece + lrow, slot = lane & 7), source k-slot = slot ^ (row & 7)

    // ---- weight rows of this wave: piece `wave` (rows 8 wave + lrow) of every tap tile; per tap the slab offset
    unsigned wrow;                                       // byte offset of (channel row, src_k) inside a slab, or out of range
    {
        const int co = c0 + chmap(wave * 8 + lrow);
        wrow = (co < p.Cout) ? (unsigned)(co * (int)p.ws_co + src_k) * 2u : SBG_OOB_OFFSET;
    }
    const int tl = lane < 9 ? lane : 0;
    const int tbl_wtap = p.tap_slab[tl] * (int)p.ws_slab * 2;      // lane t holds the slab byte offset of tap t (v_readlane, no scalar memory in the loop)
    // ---- halo pieces of this wave: slots q = 16 k + 2 wave + e for steps k = 0, 1, 2 (in-loop) and q = 8 e + wave, e = 0..4 (first slice)
    auto halo_src = [&](int q) -> unsigned {             // byte offset of (pixel 8 q + lrow, src_k) of slice 0, or out of range
        const int hp = q * 8 + lrow;
        const int r = hp / PW, c = hp - r * PW;
        const int iy = oy0 + dymin + r, ix = ox0 + dxmin + c;
        const bool ok = (q < HPIECES) & (hp < NPIX) & ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
        return ok ? (unsigned)(n * (int)p.xs_n + iy * (int)p.xs_h + ix * (int)p.xs_w + src_k) * 2u : SBG_OOB_OFFSET;
    };
    auto dma_x = [&](unsigned src, int chunk, unsigned lds_off) {
        const unsigned okm = 0u - (unsigned)((src != SBG_OOB_OFFSET) & (chunk * 64 + src_k < p.Cin));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(smem + lds_off), 16, ((src + (unsigned)chunk * 128u) & okm) | (SBG_OOB_OFFSET & ~okm), 0, 0, 0);
    };
    // the two weight pieces of step (chunk, k): taps 2k, 2k + 1 (k == 4: tap 8 and a spare)
    auto dma_w_piece = [&](int chunk, int k, int slot, bool live, int e) {
        {
            const int t = 2 * k + e;
            const bool ok = live & (t < 9) & (wrow != SBG_OOB_OFFSET) & (chunk * 64 + src_k < p.Cin);
            const unsigned okm = 0u - (unsigned)ok;
            const unsigned off = wrow + (unsigned)__builtin_amdgcn_readlane(tbl_wtap, t < 9 ? t : 0) + (unsigned)chunk * 128u;
            unsigned char* dst = (t < 9) ? smem + RING0 + slot * WSTAGE + e * 8192 + wave * 1024 : smem + DUMP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)dst, 16, (off & okm) | (SBG_OOB_OFFSET & ~okm), 0, 0, 0);
        }
    };
    auto dma_w = [&](int chunk, int k, int slot, bool live) { dma_w_piece(chunk, k, slot, live, 0); dma_w_piece(chunk, k, slot, live, 1); };
    // `late`: a step's four DMA instructions go out one behind each group of eight MFMAs, not together in front of the step's first fragment
    // read: all eight waves leave the barrier together, so four back-to-back LDS-DMA issues (~100 cycles each) kept the matrix pipe idle at the
    // head of every step.  Measured (scratch/kbench_ab.py, one device, interleaved): [64,256,128^2] -> 128 725 -> 781 TF, [64,512,64^2] -> 256
    // 841 -> 936, [64,512,32^2] -> 512 826 -> 918 (border launch included).  
    constexpr bool late = true;              // (the early form is in the history: as a run-time switch it cost nine spilled registers)

    // ---- prologue: halo of slice 0 (five slots per wave), weights of steps 0, 1, 2
#pragma unroll
    for (int e = 0; e < 5; e++) {
        const int q = 8 * e + wave;
        dma_x(halo_src(q), 0, q < HPIECES ? (unsigned)(q * 1024) : (unsigned)DUMP);
    }
#pragma unroll
    for (int k = 0; k < LEAD; k++) dma_w(0, k, k, true);

    // ---- MFMA coordinates: wave = (pixel group pg: tile rows 2 pg, 2 pg + 1) x (channel half cg: 32 channels)
    const int cg = wave & 1, pg = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void_ptr)smem);
    // a fragments: row R = 32 cg + 16 i + fr of a tap tile, k-slot fg + 4 ks at position slot ^ (R & 7): byte offsets for ks = 0, 1
    const unsigned offA0 = (unsigned)((32 * cg + fr) * 128 + ((fg ^ (fr & 7)) << 4)), offA1 = offA0 ^ 64u;
    // b fragments: halo pixel hp = (2 pg + (j >> 1) + sy) * PW + 16 (j & 1) + fr + sx for the four window shifts sh = 2 sy + sx, same swizzle
    // by hp & 7 (ks = 1: ^ 64 before the buffer base is added).  Which shift a tap reads is fixed by the canonical tap order the launcher
    // establishes: phase 0 = shifts 0, 1, 2, 3; phase 1 = 1, 3; phase 2 = 2, 3; phase 3 = 3.
    unsigned offB[4][4];
#pragma unroll
    for (int sh = 0; sh < 4; sh++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int hp = (2 * pg + (j >> 1) + (sh >> 1)) * PW + 16 * (j & 1) + fr + (sh & 1);
            offB[sh][j] = (unsigned)(hp * 128 + ((fg ^ (hp & 7)) << 4));
        }

    float4_t acc[4][2][4];                               // [phase][channel fragment][pixel fragment]
#pragma unroll
    for (int ph = 0; ph < 4; ph++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[ph][i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    int slot = 0;                                        // ring slot of the current step
    for (int c = 0; c < kchunks; c++) {
        const unsigned hb = (unsigned)((c & 1) * HALO);  // this slice's halo buffer
        sbg_static_for<NSTEP>([&](auto kt) {
            constexpr int k = decltype(kt)::value;
            // cnt(k) = 4 for k < 3, else 2;  allowed outstanding = cnt(k - 2) + cnt(k - 1) (indices mod 5)
            constexpr int allowed = (k == 0) ? 4 : (k == 1) ? 6 : (k == 2) ? 8 : (k == 3) ? 8 : 6;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(allowed) : "memory");
            __builtin_amdgcn_s_barrier();
            // ---- loads: [halo pieces of slice c + 1 -> the other buffer] [weights of step s + 3 -> the slot step s - 1 used].  `late`: one
            // DMA instruction behind each group of eight MFMAs instead of all four in front of the step's first fragment read (experiment)
            auto dma_halo_piece = [&](int e) {
                const int q = 16 * k + 2 * wave + e;
                const bool live = (c + 1 < kchunks) & (q < HPIECES);
                dma_x(live ? halo_src(q) : SBG_OOB_OFFSET, c + 1, live ? (unsigned)(((c + 1) & 1) * HALO + q * 1024) : (unsigned)DUMP);
            };
            constexpr int k3 = (k + LEAD) % NSTEP;
            const int c3 = c + (k + LEAD) / NSTEP;
            if constexpr (!late) {
                if constexpr (k < 3) { dma_halo_piece(0); dma_halo_piece(1); }
                dma_w(c3, k3, (slot + LEAD) & (NRING - 1), c3 < kchunks);
            }
            // ---- this step's taps
            const unsigned abase = lds_base + (unsigned)(RING0 + slot * WSTAGE), bbase = lds_base + hb;
            constexpr int NTAP = (k < 4) ? 2 : 1;
            constexpr int ph = (k < 2) ? 0 : (k < 4 ? k - 1 : 3);       // taps 0-3 -> phase 0, 4-5 -> 1, 6-7 -> 2, 8 -> 3
            short8_t fa[2][2], fb[2][4];
            // groups g = 2 e + ks: a fragments (i = 0, 1) and b fragments (j = 0..3) of tap 2 k + e, k-half ks; the next group is in flight
            // while the current one multiplies
            auto issue_grp = [&](auto gt) {
                constexpr int g = decltype(gt)::value, e = g >> 1, ks = g & 1, buf = g & 1, t = 2 * k + e;
                constexpr int sh = (t < 4) ? t : (t == 4 ? 1 : t == 6 ? 2 : 3);
                const unsigned aa = abase + (ks ? offA1 : offA0);
                lds_read128_issue<e * 8192>(fa[buf][0], aa);
                lds_read128_issue<e * 8192 + 16 * 128>(fa[buf][1], aa);
#pragma unroll
                for (int j = 0; j < 4; j++) lds_read128_issue<0>(fb[buf][j], (offB[sh][j] ^ (unsigned)(ks * 64)) + bbase);
            };
            constexpr int NG = 2 * NTAP;
            issue_grp(std::integral_constant<int, 0>{});
            sbg_static_for<NG>([&](auto gt) {
                constexpr int g = decltype(gt)::value, buf = g & 1;
                if constexpr (g + 1 < NG) issue_grp(std::integral_constant<int, g + 1>{});
                lds_wait6<(g + 1 < NG ? 6 : 0)>(fa[buf][0], fa[buf][1], fb[buf][0], fb[buf][1], fb[buf][2], fb[buf][3]);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[ph][i][j] = Mfma<MF>::run(fa[buf][i], fb[buf][j], acc[ph][i][j]);
                if constexpr (late) {
                    // group g of NG carries DMA instructions [g * 4 / NG, (g + 1) * 4 / NG) of the step's four: weight piece 0, 1, halo piece 0, 1
                    constexpr int d0 = g * 4 / NG, d1 = (g + 1) * 4 / NG;
                    sbg_static_for<4>([&](auto dt) {
                        constexpr int d = decltype(dt)::value;
                        if constexpr (d >= d0 && d < d1) {
                            if constexpr (d < 2) dma_w_piece(c3, k3, (slot + LEAD) & (NRING - 1), c3 < kchunks, d);
                            else if constexpr (k < 3) dma_halo_piece(d - 2);
                        }
                    });
                }
            });
            slot = (slot + 1) & (NRING - 1);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the spare loads of the last steps: nothing of this workgroup's LDS may be written after it ends

    // ---- epilogue: phase ph -> output pixels (oy, ox) of its grid, offset ph_yoff, strides already those of the phase grid
#pragma unroll
    for (int ph = 0; ph < 4; ph++)
        store_phase<YDT>(p, acc[ph], c0 + 32 * cg, fg, n, oy0, ox0, pg, fr, p.ph_OH[ph], p.ph_OW[ph], p.ph_yoff[ph]);
}

template <class MF>
static int launch_up2(const ConvArgs& a, unsigned x_bytes, unsigned w_bytes, int Hm, int Wm, int dymin, int dxmin, hipStream_t stream)
{
    constexpr int lds = 2 * 38 * 1024 + 4 * 16384 + 1024;
    const int tiles_y = Hm / 8, tiles_x = Wm / 32;
    const int64_t nblk = (int64_t)a.N * tiles_y * tiles_x * a.ctiles;
    if (nblk > INT32_MAX || nblk < 1) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0, pix = (double)a.N * Hm * Wm;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * 9.0 * pix * a.Cout * (double)a.Cin,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * 9 * a.Cout * (double)a.Cin + ys * 4.0 * pix * (double)a.Cout,
                      {(int)(4.0 * pix), a.Cout, a.Cin, 9, 2, Hm, 9064256});      // 9xxxxxx = conv_up2_kernel (profiles/summarize.py)
#define SBG_UP2_LAUNCH(YDT) do { auto kern = conv_up2_kernel<MF, YDT>; \
        if (!SBG_RAISE_LDS_ONCE(kern, lds)) return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds); \
        SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(512), lds, stream, a, x_bytes, w_bytes, tiles_y, tiles_x, dymin, dxmin); } while (0)
    if (a.ydtype == SBG_F32) SBG_UP2_LAUNCH(SBG_F32); else if (a.ydtype == SBG_BF16) SBG_UP2_LAUNCH(SBG_BF16); else SBG_UP2_LAUNCH(SBG_F16);
#undef SBG_UP2_LAUNCH
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

// Returns SBG_OK / an error, or -1 when the launch is not a 4 / 2 / 2 / 1-tap stride-2 transposed convolution this kernel covers.  On
// success the region [0, Hm) x [0, Wm) of every phase grid has been written; `border` receives the (up to four) rectangles that remain --
// the last row of the phases that have Hm + 1 rows, the last column of those with Wm + 1 columns -- as phases of an ordinary launch.
int sbg_conv_up2_dispatch(ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, sbg_conv_params* border, const sbg_conv_params* q, hipStream_t stream)
{
    static const char* off = sbg_env("SBG_CONV_NO_UP2");
    if (off) return -1;
    if (a.nphase != 4 || a.stride != 1 || a.ksplit > 1 || a.accumulate) return -1;
    if (a.ph_ntaps[0] != 4 || a.ph_ntaps[1] != 2 || a.ph_ntaps[2] != 2 || a.ph_ntaps[3] != 1) return -1;
    if ((a.Cout % 8) != 0 || (a.Cin % 8) != 0 || a.Cout < 64 || a.Cin < 64) return -1;
    if (x_bytes >= (int64_t)SBG_OOB_OFFSET || w_bytes >= (int64_t)SBG_OOB_OFFSET) return -1;
    if ((((uintptr_t)a.y) & 15) != 0 || ((a.ys_n | a.ys_h | a.ys_w) & 7) != 0) return -1;
    int dymin = a.tap_dy[0], dxmin = a.tap_dx[0];
    for (int t = 0; t < 9; t++) { if (a.tap_dy[t] < dymin) dymin = a.tap_dy[t]; if (a.tap_dx[t] < dxmin) dxmin = a.tap_dx[t]; }
    for (int t = 0; t < 9; t++) if (a.tap_dy[t] - dymin > 1 || a.tap_dx[t] - dxmin > 1) return -1;
    {   // canonical tap order: inside every phase by (dy, dx); the window shifts must then read 0 1 2 3 | 1 3 | 2 3 | 3 (the kernel hard-codes them)
        static const int want[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};
        for (int i = 0; i < 4; i++) {
            if (a.ph_tap0[i] != (i == 0 ? 0 : i == 1 ? 4 : i == 2 ? 6 : 8)) return -1;
            for (int u = a.ph_tap0[i]; u < a.ph_tap0[i] + a.ph_ntaps[i]; u++)
                for (int v = u + 1; v < a.ph_tap0[i] + a.ph_ntaps[i]; v++)
                    if (a.tap_dy[v] < a.tap_dy[u] || (a.tap_dy[v] == a.tap_dy[u] && a.tap_dx[v] < a.tap_dx[u])) {
                        std::swap(a.tap_dy[u], a.tap_dy[v]); std::swap(a.tap_dx[u], a.tap_dx[v]); std::swap(a.tap_slab[u], a.tap_slab[v]);
                    }
        }
        for (int t = 0; t < 9; t++) if ((a.tap_dy[t] - dymin) * 2 + (a.tap_dx[t] - dxmin) != want[t]) return -1;
    }
    int Hm = a.ph_OH[0], Wm = a.ph_OW[0];
    for (int i = 1; i < 4; i++) { if (a.ph_OH[i] < Hm) Hm = a.ph_OH[i]; if (a.ph_OW[i] < Wm) Wm = a.ph_OW[i]; }
    if (Hm < 8 || Wm < 32 || (Hm % 8) != 0 || (Wm % 32) != 0) return -1;
    for (int i = 0; i < 4; i++) {
        if (a.ph_OH[i] > Hm + 1 || a.ph_OW[i] > Wm + 1) return -1;
        if ((a.ph_yoff[i] & 7) != 0) return -1;
    }
    // ---- the border rectangles, as phases of a second launch: pixel (u, v) of rectangle (r0, c0) of phase i is pixel (r0 + u, c0 + v) of
    // that phase's grid, so its taps are the phase's taps shifted by (r0, c0) and its outputs start r0 rows / c0 columns further
    *border = *q;
    border->nphase = 0; border->ntaps = 0;
    for (int i = 0; i < 4; i++) {
        for (int side = 0; side < 2; side++) {
            int r0, c0, rh, rw;
            if (side == 0) { if (a.ph_OH[i] == Hm) continue; r0 = Hm; c0 = 0; rh = 1; rw = a.ph_OW[i]; }          // the extra row, all columns
            else           { if (a.ph_OW[i] == Wm) continue; r0 = 0; c0 = Wm; rh = Hm; rw = 1; }                  // the extra column, the main rows
            if (border->nphase >= 4) return -1;
            const int ph = border->nphase;
            int nt = 0;
            for (int t = 0; t < a.ph_ntaps[i]; t++) {
                const int dy = a.tap_dy[a.ph_tap0[i] + t] + r0, dx = a.tap_dx[a.ph_tap0[i] + t] + c0;
                if (dy + rh - 1 < 0 || dy >= a.IH || dx + rw - 1 < 0 || dx >= a.IW) continue;                     // reads only zeros
                if (border->ntaps >= SBG_MAX_TAPS) return -1;
                border->tap_dy[border->ntaps] = dy; border->tap_dx[border->ntaps] = dx; border->tap_slab[border->ntaps] = a.tap_slab[a.ph_tap0[i] + t];
                border->ntaps++; nt++;
            }
            if (nt == 0) return -1;                      // (would need a zero fill: not a shape this path meets)
            border->ph_ntaps[ph] = nt; border->ph_oh[ph] = rh; border->ph_ow[ph] = rw;
            border->ph_yoff[ph] = a.ph_yoff[i] + (int64_t)r0 * a.ys_h + (int64_t)c0 * a.ys_w;
            border->nphase++;
        }
    }
    a.ctiles = (a.Cout + 63) / 64;
    const int rc = bf16 ? launch_up2<bf16_mfma>(a, (unsigned)x_bytes, (unsigned)w_bytes, Hm, Wm, dymin, dxmin, stream)
                        : launch_up2<f16_mfma>(a, (unsigned)x_bytes, (unsigned)w_bytes, Hm, Wm, dymin, dxmin, stream);
    return rc;
}
