// conv_igemm.hip -- implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}).
//
// One kernel serves forward convolution, data gradient (a convolution with flipped / transposed weights) and each of
// the four sub-pixel phases of a stride-2 transposed convolution: the host describes the launch as a tap list
//     y[n, oy*ysh + .., ox*ysw + .., co] (+)= sum_t sum_ci x[n, oy*stride + dy[t], ox*stride + dx[t], ci] * w[slab[t]][co][ci]
// (see include/sbg_hip.h).  This replaces the cuDNN calls behind conv2d_gradfix.conv2d / conv_transpose2d
// (stylegan2ada/torch_utils/ops/conv2d_gradfix.py:107-137).
//
// GEMM view: D[co][pixel] = sum_k W[co][k] * X[k][pixel], k = (tap, ci).  The weights are the MFMA A operand and the
// gathered activations the B operand, so a lane ends up holding 4 consecutive output channels of one pixel -- a
// contiguous 8-B (bf16) / 16-B (fp32) channel-minor store.
//
// Workgroup = 256 lanes = 4 waves (64-wide), tile BC output channels x BP output pixels, K-step 32 (one MFMA deep).
// Activations are channel-minor, so for a fixed tap the 32-channel slice of a pixel is 64 contiguous bytes: every lane
// stages 16-B pieces global -> VGPR -> LDS (issue early / write late, 2 LDS stages, one barrier per K-step).
// LDS image per operand: [k-group g = 0..3][row][8 x 16-bit] (16-B cells).  A ds_read_b128 of MFMA fragments touches 16
// rows that are distinct mod 16 -> 16 distinct 16-B slots of the 256-B bank row: conflict-free; the staging writes put 8
// consecutive rows of one k-group in each 8-lane store group: 128 contiguous bytes, conflict-free.
#include "conv_common.h"
#include <cstdlib>

using namespace sbgconv;

namespace {

// lane -> (row within a 16-row staging group, k-group): 8 consecutive lanes share a k-group and walk 8 rows.
static __device__ __forceinline__ int stage_row16(int lane) { return (lane & 7) | ((lane >> 5) << 3); }
static __device__ __forceinline__ int stage_kgrp(int lane)  { return (lane >> 3) & 3; }

// Epilogue shared by both main loops: lane holds channels c0 + wc + 16 i + 4 fg + {0..3} of pixel p0 + wp + 16 j + fr.
template <int TC, int TP>
static __device__ __forceinline__ void conv_epilogue(const ConvArgs& p, float4_t (&acc)[TC][TP], int c0, int p0, int wc, int wp, int fr, int fg)
{
    // per-channel terms depend on the channel tile only: fetch them once (16-B loads when the 4 channels are in range)
    float4_t bias4[TC];
    bool full4[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {
        const int co = c0 + wc + 16 * i + 4 * fg;
        full4[i] = (co + 4 <= p.Cout);
        bias4[i] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; e++) if (co + e < p.Cout) bias4[i][e] = p.bias[co + e];
        }
    }
    const bool plain = (p.act <= SBG_ACT_LINEAR) && p.gain == 1.f && p.clamp < 0.f && !p.bias && !p.noise && !p.oscale;
#pragma unroll
    for (int j = 0; j < TP; j++) {
        const int pix = p0 + wp + 16 * j + fr;
        if (pix >= p.P) continue;
        const int ox = pix % p.OW, t = pix / p.OW, oy = t % p.OH, n = t / p.OH;
        const int64_t yoff = (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        const float nz = (!plain && p.noise) ? p.noise[(int64_t)n * p.noise_sn + (int64_t)oy * p.OW + ox] : 0.f;
#pragma unroll
        for (int i = 0; i < TC; i++) {
            const int co = c0 + wc + 16 * i + 4 * fg;
            if (co >= p.Cout) continue;
            float4_t v = acc[i][j];
            if (!plain) {
                if (p.oscale) {
                    const float* sc = p.oscale + (int64_t)n * p.Cout + co;
#pragma unroll
                    for (int e = 0; e < 4; e++) if (co + e < p.Cout) v[e] *= sc[e];
                }
                v += nz;
                v += bias4[i];
                if (p.act == SBG_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (v[e] > 0.f) ? v[e] : v[e] * p.alpha;
                } else if (p.act == SBG_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (v[e] > 0.f) ? v[e] : 0.f;
                }
                v *= p.gain;
                if (p.clamp >= 0.f) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (v[e] > -p.clamp && v[e] < p.clamp) ? v[e] : (v[e] >= 0.f ? p.clamp : -p.clamp);
                }
            }
            const bool full = full4[i];
            if (p.ydtype == SBG_F32) {
                float* dst = (float*)p.y + yoff + co;
                if (full && ((((uintptr_t)dst) & 15) == 0)) {
                    float4_t o = v;
                    if (p.accumulate) { float4_t old = *reinterpret_cast<float4_t*>(dst); o += old; }
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

template <class MF, int BC, int BP, int WGC, int WGP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p)
{
    static_assert(WGC * WGP == 4, "4 waves per workgroup");
    constexpr int WC = BC / WGC, WP = BP / WGP;        // wave tile
    constexpr int TC = WC / 16,  TP = WP / 16;         // 16x16 MFMA tiles per wave
    constexpr int RA = BC / 64,  RB = BP / 64;         // staging rows per lane (16 rows per wave-instruction, 4 waves)
    static_assert(BC % 64 == 0 && BP % 64 == 0, "tile must be a multiple of 64");

    // LDS: 2 stages x { A [4][BC] cells, B [4][BP] cells }, cell = 16 B.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = 4 * BC * 16, B_BYTES = 4 * BP * 16, STAGE = A_BYTES + B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups b and b+8 share an XCD (L2); give each XCD a contiguous run of tiles.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ct = bid % p.ctiles, pt = bid / p.ctiles;
    const int c0 = ct * BC, p0 = pt * BP;

    // ---- per-lane staging coordinates -------------------------------------------------------------------------
    const int srow = wave * 16 + stage_row16(lane), sg = stage_kgrp(lane);
    // B operand rows = output pixels: decode (n, oy, ox) once.
    int  b_iy0[RB], b_ix0[RB]; int64_t b_base[RB]; bool b_ok[RB];
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const int pix = p0 + srow + 64 * i;
        b_ok[i] = pix < p.P;
        const int pp = b_ok[i] ? pix : 0;
        const int ox = pp % p.OW, t = pp / p.OW, oy = t % p.OH, n = t / p.OH;
        b_iy0[i] = oy * p.stride; b_ix0[i] = ox * p.stride; b_base[i] = (int64_t)n * p.xs_n;
    }
    // A operand rows = output channels.
    int64_t a_off[RA]; bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; i++) {
        const int co = c0 + srow + 64 * i;
        a_ok[i] = co < p.Cout;
        a_off[i] = (int64_t)(a_ok[i] ? co : 0) * p.ws_co;
    }

    const int kchunks = (p.Cin + 31) >> 5;
    const int nsteps = p.ntaps * kchunks;

    short8_t ra[RA], rb[RB];
    auto issue_loads = [&](int step) {
        const int t = step / kchunks, ck = (step - t * kchunks) * 32 + sg * 8;
        const bool kok = ck < p.Cin;
        const int dy = p.tap_dy[t], dx = p.tap_dx[t];
        const unsigned short* wslab = p.w + (int64_t)p.tap_slab[t] * p.ws_slab + ck;
#pragma unroll
        for (int i = 0; i < RA; i++) {
            short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kok && a_ok[i]) v = *reinterpret_cast<const short8_t*>(wslab + a_off[i]);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int iy = b_iy0[i] + dy, ix = b_ix0[i] + dx;
            short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kok && b_ok[i] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
                v = *reinterpret_cast<const short8_t*>(p.x + b_base[i] + (int64_t)iy * p.xs_h + (int64_t)ix * p.xs_w + ck);
            rb[i] = v;
        }
    };
    auto write_stage = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE;
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < RA; i++) *reinterpret_cast<short8_t*>(sa + (sg * BC + srow + 64 * i) * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; i++) *reinterpret_cast<short8_t*>(sb + (sg * BP + srow + 64 * i) * 16) = rb[i];
    };

    // ---- MFMA coordinates ---------------------------------------------------------------------------------------
    const int wc = (wave / WGP) * WC, wp = (wave % WGP) * WP;
    const int fr = lane & 15, fg = lane >> 4;
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    issue_loads(0);
    write_stage(0);
    __syncthreads();
    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        if (s + 1 < nsteps) issue_loads(s + 1);
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + A_BYTES;
        short8_t fa[TC], fb[TP];
#pragma unroll
        for (int i = 0; i < TC; i++) fa[i] = *reinterpret_cast<const short8_t*>(sa + (fg * BC + wc + 16 * i + fr) * 16);
#pragma unroll
        for (int j = 0; j < TP; j++) fb[j] = *reinterpret_cast<const short8_t*>(sb + (fg * BP + wp + 16 * j + fr) * 16);
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[i], fb[j], acc[i][j]);
        if (s + 1 < nsteps) write_stage(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<TC, TP>(p, acc, c0, p0, wc, wp, fr, fg);
}

template <class MF, int BC, int BP, int WGC, int WGP>
static int launch_conv(ConvArgs& a, hipStream_t stream)
{
    a.ptiles = (a.P + BP - 1) / BP;
    a.ctiles = (a.Cout + BC - 1) / BC;
    const int64_t nblk = (int64_t)a.ptiles * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    constexpr int lds = 2 * (4 * BC * 16 + 4 * BP * 16);
    auto kern = conv_igemm_kernel<MF, BC, BP, WGC, WGP>;
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * a.P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * a.P * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {a.P, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, BC * 1000 + BP});
    SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(256), lds, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA main loop (default).  The register-staged loop above keeps ONE K-step of loads in flight, and under load a
// global -> LDS round trip costs several K-steps of MFMA time, so waves sit in s_waitcnt (rocprofv3: SQ_WAIT_ANY 58 % of
// wave cycles, matrix pipe 22 % busy).  Here every stage is filled by `buffer_load_dwordx4 ... lds` (no VGPR staging, no
// ds_write), NSTAGE = 4 LDS stages deep with 3 K-steps of loads in flight behind a COUNTED s_waitcnt vmcnt(N) and one
// raw s_barrier per K-step.  Out-of-image taps, ragged rows and channel tails are fetched at an out-of-range buffer
// offset, which the hardware range check turns into zeros -- padding costs no branches.
// LDS image per operand and stage: [row][4 slots of 16 B] written lane-linearly (16 rows x 4 slots = 1 KiB per
// wave-instruction, so a row's 64 B come from one 64-B global segment); the k-group -> slot XOR swizzle
// slot = g ^ ((-(row >> 2)) & 3) is applied on the SOURCE address and on the fragment read (conflict-free ds_read_b128).

typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define SBG_OOB_OFFSET 0x80000000u      // >= num_records of every descriptor built below (tensors < 2 GiB)

template <class MF, int BC, int BP, int WGC, int WGP, int NSTAGE, int MINW>
__global__ __launch_bounds__(WGC * WGP * 64, MINW) void conv_igemm_dma_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int NW = WGC * WGP;                      // waves per workgroup (4 or 8)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int WC = BC / WGC, WP = BP / WGP;
    constexpr int TC = WC / 16,  TP = WP / 16;
    constexpr int DEPTH = NSTAGE - 1;
    constexpr int A_BYTES = BC * 64, B_BYTES = BP * 64, STAGE = A_BYTES + B_BYTES;
    static_assert(BC % (16 * NW) == 0 && BP % (16 * NW) == 0, "every wave stages whole 16-row pieces of both operands");
    constexpr int IA = BC / (16 * NW), IB = BP / (16 * NW);   // DMA instructions per wave per stage (16 rows each)
    constexpr int PER_STEP = IA + IB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int ct = bid % p.ctiles, pt = bid / p.ctiles;
    const int c0 = ct * BC, p0 = pt * BP;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);

    // ---- per-lane DMA coordinates: lane -> (row = 16-row group base + lane / 4, slot = lane & 3), source k-group = slot ^ swz
    const int lrow = lane >> 2, lslot = lane & 3;
    const int src_g = lslot ^ ((-(lrow >> 2)) & 3);
    int b_iy0[IB], b_ix0[IB]; unsigned b_base[IB];
#pragma unroll
    for (int i = 0; i < IB; i++) {
        const int pix = p0 + (wave * IB + i) * 16 + lrow;
        const bool ok = pix < p.P;
        const int pp = ok ? pix : 0;
        const int ox = pp % p.OW, t = pp / p.OW, oy = t % p.OH, n = t / p.OH;
        b_iy0[i] = ok ? oy * p.stride : -(1 << 28);           // invalid rows fail every range test below
        b_ix0[i] = ox * p.stride;
        b_base[i] = (unsigned)(n * (int)p.xs_n) * 2u;
    }
    unsigned a_base[IA];
#pragma unroll
    for (int i = 0; i < IA; i++) {
        const int co = c0 + (wave * IA + i) * 16 + lrow;
        a_base[i] = (co < p.Cout) ? (unsigned)(co * (int)p.ws_co) * 2u : SBG_OOB_OFFSET;
    }
    const int kchunks = (p.Cin + 31) >> 5;
    const int nsteps = p.ntaps * kchunks;

    auto issue = [&](int step) {
        const int t = step / kchunks, ck = (step - t * kchunks) * 32 + src_g * 8;
        const bool kok = ck < p.Cin;
        unsigned char* st = smem + (step % NSTAGE) * STAGE;
        const int dy = p.tap_dy[t], dx = p.tap_dx[t];
        const unsigned wtap = (unsigned)(p.tap_slab[t] * (int)p.ws_slab + ck) * 2u;
#pragma unroll
        for (int i = 0; i < IA; i++) {
            // branch-free select: a masked-off lane would leave stale bytes in LDS instead of zeros
            const unsigned okm = 0u - (unsigned)(kok & (a_base[i] != SBG_OOB_OFFSET));
            const unsigned off = ((a_base[i] + wtap) & okm) | (SBG_OOB_OFFSET & ~okm);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(st + (wave * IA + i) * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IB; i++) {
            const int iy = b_iy0[i] + dy, ix = b_ix0[i] + dx;
            const unsigned okm = 0u - (unsigned)(kok & ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW));
            const unsigned real = b_base[i] + (unsigned)(iy * (int)p.xs_h + ix * (int)p.xs_w + ck) * 2u;
            const unsigned off = (real & okm) | (SBG_OOB_OFFSET & ~okm);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(st + A_BYTES + (wave * IB + i) * 1024), 16, off, 0, 0, 0);
        }
    };

    const int wc = (wave / WGP) * WC, wp = (wave % WGP) * WP;
    const int fr = lane & 15, fg = lane >> 4;
    const int rd_slot = fg ^ ((-(fr >> 2)) & 3);
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < DEPTH; s++) if (s < nsteps) issue(s);

    for (int s = 0; s < nsteps; s++) {
        // this wave's loads of step s have landed once at most (steps issued after s) * PER_STEP remain outstanding
        const int ahead = (nsteps - 1 - s < DEPTH - 1) ? nsteps - 1 - s : DEPTH - 1;
        if (DEPTH >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER_STEP) : "memory");
        else if (ahead >= 1)          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * PER_STEP) : "memory");
        else                          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // every wave's part of stage s is in LDS; every wave is done reading stage s - 1
        if (s + DEPTH < nsteps) issue(s + DEPTH);       // refills the stage read in step s - 1
        const unsigned char* sa = smem + (s % NSTAGE) * STAGE;
        const unsigned char* sb = sa + A_BYTES;
        short8_t fa[TC], fb[TP];
#pragma unroll
        for (int i = 0; i < TC; i++) fa[i] = *reinterpret_cast<const short8_t*>(sa + (wc + 16 * i + fr) * 64 + rd_slot * 16);
#pragma unroll
        for (int j = 0; j < TP; j++) fb[j] = *reinterpret_cast<const short8_t*>(sb + (wp + 16 * j + fr) * 64 + rd_slot * 16);
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[i], fb[j], acc[i][j]);
    }
    conv_epilogue<TC, TP>(p, acc, c0, p0, wc, wp, fr, fg);
}

template <class MF, int BC, int BP, int WGC, int WGP, int NSTAGE, int MINW>
static int launch_conv_dma(ConvArgs& a, unsigned x_bytes, unsigned w_bytes, hipStream_t stream)
{
    a.ptiles = (a.P + BP - 1) / BP;
    a.ctiles = (a.Cout + BC - 1) / BC;
    const int64_t nblk = (int64_t)a.ptiles * a.ctiles;
    if (nblk > INT32_MAX) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    constexpr int lds = NSTAGE * (BC * 64 + BP * 64);
    auto kern = conv_igemm_dma_kernel<MF, BC, BP, WGC, WGP, NSTAGE, MINW>;
    if (lds > 64 * 1024 && !SBG_RAISE_LDS_ONCE(kern, lds))
        return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds);
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * a.P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * a.P * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {a.P, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, BC * 1000 + BP});
    SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(WGC * WGP * 64), lds, stream, a, x_bytes, w_bytes);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

template <class MF>
static int dispatch_conv(ConvArgs& a, int64_t x_bytes, int64_t w_bytes, bool allow_dma, hipStream_t stream)
{
    // LDS-DMA pipeline whenever both operands fit a 2 GiB buffer descriptor (the out-of-range sentinel sits at 2 GiB)
    const bool dma = allow_dma && x_bytes < (int64_t)SBG_OOB_OFFSET && w_bytes < (int64_t)SBG_OOB_OFFSET;
    // tile choice: few output channels -> pixel-heavy tile; otherwise 128 x 128.
    if (dma) {
        const char* v = sbg_env("SBG_CONV_TILE");      // experiment switch (tile / pipeline-depth variants)
        const int variant = v ? atoi(v) : 0;
        if (a.Cout <= 64) return launch_conv_dma<MF, 64, 256, 1, 4, 4, 1>(a, (unsigned)x_bytes, (unsigned)w_bytes, stream);
        if (variant == 1) return launch_conv_dma<MF, 128, 256, 2, 4, 3, 4>(a, (unsigned)x_bytes, (unsigned)w_bytes, stream);   // 8 waves, 2 WG/CU
        if (variant == 2) return launch_conv_dma<MF, 128, 256, 2, 4, 4, 2>(a, (unsigned)x_bytes, (unsigned)w_bytes, stream);   // 8 waves, 1 WG/CU
        if (variant == 3) return launch_conv_dma<MF, 128, 128, 2, 4, 4, 4>(a, (unsigned)x_bytes, (unsigned)w_bytes, stream);   // 8 waves, 128^2
        return launch_conv_dma<MF, 128, 128, 2, 2, 4, 2>(a, (unsigned)x_bytes, (unsigned)w_bytes, stream);
    }
    if (a.Cout <= 64)  return launch_conv<MF, 64, 256, 1, 4>(a, stream);
    return launch_conv<MF, 128, 128, 2, 2>(a, stream);
}

} // namespace

static int phase_min_tiles()        // experiment switch: smallest tile count for which a transposed convolution runs as ONE multi-phase launch
{
    static const char* e = sbg_env("SBG_PHASE_MIN_TILES");
    return e ? atoi(e) : 16;            // 256 -> 16: the 4x4 .. 32x32 up-sampling layers as one launch instead of four latency-bound ones (-1.4 ms per step)
}

extern "C" int sbg_conv2d_igemm(const sbg_conv_params* q, sbg_stream_t stream)
{
    SBG_CHECK(q && q->x && q->w && q->y, "conv2d_igemm: null pointer");
    SBG_CHECK(q->xdtype == SBG_BF16 || q->xdtype == SBG_F16, "conv2d_igemm: x/w must be bf16 or f16 (fp32 inputs are split by the host)");
    SBG_CHECK(q->ydtype == SBG_F32 || q->ydtype == SBG_BF16 || q->ydtype == SBG_F16, "conv2d_igemm: bad output dtype");
    SBG_CHECK(q->N >= 0 && q->IH >= 1 && q->IW >= 1 && q->OH >= 1 && q->OW >= 1 && q->Cin >= 8 && q->Cout >= 1, "conv2d_igemm: bad sizes");
    SBG_CHECK((q->Cin % 8) == 0, "conv2d_igemm: Cin must be a multiple of 8 (pad on the host)");
    SBG_CHECK(q->ntaps >= 1 && q->ntaps <= SBG_MAX_TAPS, "conv2d_igemm: 1..%d taps", SBG_MAX_TAPS);
    SBG_CHECK(q->stride >= 1, "conv2d_igemm: stride must be >= 1");
    SBG_CHECK(!q->accumulate || q->ydtype == SBG_F32, "conv2d_igemm: accumulate needs an fp32 output");
    SBG_CHECK(q->ksplit <= 1 || q->workspace != nullptr, "conv2d_igemm: ksplit > 1 needs a workspace");
    SBG_CHECK(q->act == 0 || q->act == SBG_ACT_LINEAR || q->act == SBG_ACT_RELU || q->act == SBG_ACT_LRELU, "conv2d_igemm: fused activation must be linear, relu or lrelu");
    SBG_CHECK(sbg_aligned16(q->x) && sbg_aligned16(q->w), "conv2d_igemm: x and w must be 16-byte aligned");
    SBG_CHECK((q->xs_n % 8) == 0 && (q->xs_h % 8) == 0 && (q->xs_w % 8) == 0 && (q->ws_slab % 8) == 0 && (q->ws_co % 8) == 0,
              "conv2d_igemm: pixel / row strides must be multiples of 8 elements");
    const int64_t P = (int64_t)q->N * q->OH * q->OW;
    SBG_CHECK(P <= INT32_MAX, "conv2d_igemm: too many output pixels");
    if (P == 0) return SBG_OK;

    ConvArgs a;
    a.x = (const unsigned short*)q->x; a.w = (const unsigned short*)q->w; a.y = q->y; a.oscale = q->oscale;
    a.bias = q->bias; a.noise = q->noise; a.noise_sn = q->noise_stride_n;
    a.act = q->act; a.alpha = q->alpha; a.gain = (q->act == 0 && q->gain == 0.f) ? 1.f : q->gain; a.clamp = (q->act == 0 && q->clamp == 0.f && q->gain == 0.f) ? -1.f : q->clamp;
    a.ydtype = q->ydtype;
    a.N = q->N; a.IH = q->IH; a.IW = q->IW; a.Cin = q->Cin; a.Cout = q->Cout; a.OH = q->OH; a.OW = q->OW;
    a.xs_n = q->xs_n; a.xs_h = q->xs_h; a.xs_w = q->xs_w; a.ys_n = q->ys_n; a.ys_h = q->ys_h; a.ys_w = q->ys_w;
    a.ws_slab = q->ws_slab; a.ws_co = q->ws_co;
    a.stride = q->stride; a.ntaps = q->ntaps;
    for (int t = 0; t < SBG_MAX_TAPS; t++) { a.tap_dy[t] = q->tap_dy[t]; a.tap_dx[t] = q->tap_dx[t]; a.tap_slab[t] = q->tap_slab[t]; }
    a.accumulate = q->accumulate;
    a.P = (int)P; a.ptiles = a.ctiles = 0; a.debug = sbg_experiment() << 8; a.lds_params = 0; a.ksplit = 1; a.y_split_stride = 0;
    a.nphase = 1; a.ph_rot_div = 1;
    for (int i = 0; i < 4; i++) { a.ph_tap0[i] = 0; a.ph_ntaps[i] = 0; a.ph_OH[i] = 0; a.ph_OW[i] = 0; a.ph_P[i] = 0; a.ph_yoff[i] = 0; }
    hipStream_t s = (hipStream_t)stream;
    int maxslab = 0;
    for (int t = 0; t < q->ntaps; t++) { SBG_CHECK(q->tap_slab[t] >= 0, "conv2d_igemm: negative weight slab"); if (q->tap_slab[t] > maxslab) maxslab = q->tap_slab[t]; }
    const int64_t x_bytes = 2 * ((int64_t)(q->N - 1) * q->xs_n + (int64_t)(q->IH - 1) * q->xs_h + (int64_t)(q->IW - 1) * q->xs_w + q->Cin);
    const int64_t w_bytes = 2 * ((int64_t)maxslab * q->ws_slab + (int64_t)(q->Cout - 1) * q->ws_co + q->Cin);
    const bool allow_dma = sbg_env("SBG_CONV_NO_DMA") == nullptr && q->xs_n >= 0 && q->xs_h >= 0 && q->xs_w >= 0 && q->ws_slab >= 0 && q->ws_co >= 0;
    if (q->nphase > 1) {
        // phases: one persistent launch when the shape fits that kernel, otherwise one launch per phase through this same entry point
        SBG_CHECK(q->nphase <= 4, "conv2d_igemm: at most 4 phases");
        const bool plain = (q->act == 0 || q->act == SBG_ACT_LINEAR) && (q->gain == 1.f || q->gain == 0.f) && q->clamp < 0.f && !q->bias && !q->noise && !q->oscale;
        SBG_CHECK(plain && (q->ksplit <= 1), "conv2d_igemm: phases exclude the fused epilogue and the K split");
        int t0 = 0; int64_t ptot = 0;
        for (int i = 0; i < q->nphase; i++) {
            SBG_CHECK(q->ph_ntaps[i] >= 1 && q->ph_oh[i] >= 1 && q->ph_ow[i] >= 1, "conv2d_igemm: bad phase %d", i);
            a.ph_tap0[i] = t0; a.ph_ntaps[i] = q->ph_ntaps[i]; a.ph_OH[i] = q->ph_oh[i]; a.ph_OW[i] = q->ph_ow[i];
            a.ph_P[i] = q->N * q->ph_oh[i] * q->ph_ow[i]; a.ph_yoff[i] = q->ph_yoff[i];
            t0 += q->ph_ntaps[i]; ptot += a.ph_P[i];
        }
        SBG_CHECK(t0 <= q->ntaps, "conv2d_igemm: phases use %d taps, %d given", t0, q->ntaps);
        {   // few channels: one streaming launch over all phases (conv_thin.hip)
            a.nphase = q->nphase;
            const int rc = sbg_conv_thin_dispatch(a, q->xdtype == SBG_BF16, s);
            if (rc >= 0) return rc;
            a.nphase = 1;
        }
        if (allow_dma && q->nphase == 4) {      // stride-2 3x3 transposed convolution: every phase from one staged halo (conv_up2.hip), then its border
            a.nphase = 4;
            sbg_conv_params border;
            const int rc = sbg_conv_up2_dispatch(a, q->xdtype == SBG_BF16, x_bytes, w_bytes, &border, q, s);
            if (rc > 0) return rc;
            if (rc == SBG_OK) return border.nphase > 0 ? sbg_conv2d_igemm(&border, stream) : SBG_OK;
            a.nphase = 1;
        }
        const int64_t tiles = ((ptot / q->nphase + 255) / 256) * q->nphase * ((q->Cout + 127) / 128);
        if (allow_dma && q->Cout > 64 && tiles >= phase_min_tiles() && x_bytes < (int64_t)0x80000000u && w_bytes < (int64_t)0x80000000u && sbg_env("SBG_CONV_NO_PHASES") == nullptr) {
            a.nphase = q->nphase;
            const int rc = sbg_conv_k64_dispatch(a, q->xdtype == SBG_BF16, x_bytes, w_bytes, nullptr, 1, s);
            if (rc >= 0) return rc;
            a.nphase = 1;
        }
        for (int i = 0; i < q->nphase; i++) {
            sbg_conv_params one = *q;
            one.nphase = 0; one.OH = q->ph_oh[i]; one.OW = q->ph_ow[i]; one.ntaps = q->ph_ntaps[i];
            one.y = (char*)q->y + q->ph_yoff[i] * (q->ydtype == SBG_F32 ? 4 : 2);
            for (int t = 0; t < one.ntaps; t++) { one.tap_dy[t] = q->tap_dy[a.ph_tap0[i] + t]; one.tap_dx[t] = q->tap_dx[a.ph_tap0[i] + t]; one.tap_slab[t] = q->tap_slab[a.ph_tap0[i] + t]; }
            const int rc = sbg_conv2d_igemm(&one, stream);
            if (rc != SBG_OK) return rc;
        }
        return SBG_OK;
    }
    if (q->ksplit <= 1) {   // few channels on both sides: the streaming kernel of conv_thin.hip
        const int rc = sbg_conv_thin_dispatch(a, q->xdtype == SBG_BF16, s);
        if (rc >= 0) return rc;
    }
    if (allow_dma) {      // K-step-64 kernels (conv_k64.hip) take every launch whose operands fit a 2 GiB buffer descriptor
        const int rc = sbg_conv_k64_dispatch(a, q->xdtype == SBG_BF16, x_bytes, w_bytes, q->workspace, q->ksplit, s);
        if (rc >= 0) return rc;
    }
    if (q->xdtype == SBG_BF16) return dispatch_conv<bf16_mfma>(a, x_bytes, w_bytes, allow_dma, s);
    return dispatch_conv<f16_mfma>(a, x_bytes, w_bytes, allow_dma, s);
}
