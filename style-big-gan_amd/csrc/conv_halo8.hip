// conv_halo8.hip -- 3x3 / stride-1 / pad-same convolution (forward and data gradient) on the gfx950 matrix cores: the halo-staged tile of
// conv_k64.hip's conv_halo_ld_kernel in an 8-wave structure whose tile boundaries cost (almost) nothing.
//
// Why a second kernel.  conv_halo_ld_kernel (12 waves: two compute groups in ping-pong + four loader waves, 168 registers per wave) pays ~3.7 us
// (plain) to ~5 us (fused tail) at every tile boundary, whatever the tile's length: fitted over 18 / 36 / 72 K-steps per tile, T = n * 0.70 us +
// 3.7..5.1 us (round 3, scratch/kbench_ab.py).  At 128 input channels -- the [N,128,256,256] (*) [128,128,3,3] layer the MFMA target is stated for --
// a tile has 18 K-steps, so a quarter of the kernel is tile boundary: the 64 KiB of output go out as 64 store instructions that the CU retires at
// ~16 B/clk (transaction-bound: 16 half lines per instruction), first for one compute group, then for the other, each time with the other
// group's 32 MFMAs as the only cover; with 168 registers there is no room to keep a finished tile while the next one starts.
//
// Structure here: 8 waves x 256 registers, every wave loads and computes (conv_up2.hip's skeleton), and TWO accumulator sets.  A finished tile
// stays in its set while the next tile accumulates into the other one, and is drained in eight units -- one (channel group, pixel segment) pair:
// tail arithmetic, 16-bit conversion, ONE 16-B store per lane -- behind the MFMAs of the next tile's first eight K-steps.  Nothing waits for the
// store tail any more, and the tail's arithmetic rides in the matrix pipe's shadow.
//   * K-step = (tap, 64-channel slice) as before: 16 KiB weight stage (ring of four, LDS-DMA three steps ahead) + the staged (TH+2) x (TW+2) halo
//     of the slice (two buffers; the next slice's 43 pieces go out one per wave in steps 0-5 of the current slice).
//   * Per step and wave: two groups (k halves) of 8 fragment reads (inline asm, counted lgkmcnt) and 16 MFMAs; the step's barrier sits BETWEEN the
//     groups: [reads g1] [wait g0, 16 MFMA, 2-3 DMA instructions behind them] [wait g1] [vmcnt(n); barrier] [reads g0 of the NEXT step] [16 MFMA,
//     drain unit].  So every fragment read has 16 MFMAs of cover and the LDS-DMA instructions never sit in front of a read (round 3: a burst of
//     LDS-DMA issues ahead of the fragment reads was what bounded every kernel of this family).
//   * Every wave issues the same number of vector-memory instructions in a given step (spare slots write zeros into a dump KiB): all vmcnt
//     counts are immediates.  Step t of a slice issues 2 weight pieces, and for t <= 5 one halo piece, for t == 6 one piece of the finished
//     tile's epilogue parameters (noise tile / bias / demodulation coefficients -> 3 KiB of LDS, read back by the drain units); a drain unit adds
//     one store.  At the barrier of step s the loads of step s - 2 and older have landed when at most [store(s-2)] + ops(s-1) + dma(s) are
//     outstanding.
// Output: bf16 / fp32, plain or with the fused tail  y = clamp(lrelu(acc * oscale[n, c] + noise[n, pixel] + bias[c]) * gain)  of conv_k64.hip's
// epilogue; accumulator layout, channel permutation (chmap) and LDS images are those of conv_k64.hip.  fp32 read-modify-write outputs, f16
// outputs and launches whose tail parameters are not 16-B aligned stay with conv_halo_ld_kernel.
#include "conv_common.h"
#include "lds_asm.h"
#include <cstdlib>
#include <utility>

using namespace sbgconv;

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
#define SBG_OOB_OFFSET 0x80000000u

static __device__ __forceinline__ int chmap(int R) { return (R & ~31) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3); }

template <int OFF>
static __device__ __forceinline__ void lds_read128(short8_t& d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
static __device__ __forceinline__ void lds_read128f(float4_t& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
static __device__ __forceinline__ void lds_read32f(float& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
template <int N>
static __device__ __forceinline__ void lds_wait8(short8_t (&a)[4], short8_t (&b)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}
static __device__ __forceinline__ void lds_wait_params(float (&nz)[4], float4_t& b0, float4_t& b1, float4_t& s0, float4_t& s1)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nz[0]), "+v"(nz[1]), "+v"(nz[2]), "+v"(nz[3]), "+v"(b0), "+v"(b1), "+v"(s0), "+v"(s1) :: "memory");
}
// a wave-uniform value made opaque to the optimiser at this point, at no cost: whatever is computed from it is computed HERE, not hoisted out of the
// tile / slice loop into registers that stay live around it (the loop pins 192 of the 256: two accumulator sets and two fragment groups)
static __device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// vmcnt(base + extra), extra in 0..4 (wave-uniform): the drain stores of the two previous steps ride in the same counter
template <int BASE> static __device__ __forceinline__ void wait_vm_plus(int extra)
{
    if (extra == 0) wait_vm<BASE>(); else if (extra == 1) wait_vm<BASE + 1>(); else if (extra == 2) wait_vm<BASE + 2>();
    else if (extra == 3) wait_vm<BASE + 3>(); else wait_vm<BASE + 4>();
}

template <class MF, int TH, int TW, int YDT, bool TAIL>
__global__ __launch_bounds__(512) void conv_halo8_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes)
{
    constexpr int NT = 9, NRING = 4, LEAD = 3, STAGE = 128 * 128;
    constexpr int PW = TW + 2, PH = TH + 2, NPIX = PW * PH, HPIECES = (NPIX + 7) / 8, HALO = HPIECES * 1024, SEG = TW / 16;
    constexpr int RING0 = 0, H0 = NRING * STAGE, P0 = H0 + 2 * HALO, PARAM = 3072, DUMP = P0 + 2 * PARAM;
    static_assert(TH * TW == 256 && HPIECES <= 48, "tile = 256 pixels; the next halo goes out as one piece per wave in six steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void_ptr)smem);

    // Every wave of this kernel carries MFMAs, fragment reads AND the load / store address arithmetic; a SIMD issues one vector instruction at a
    // time for its two waves and an MFMA holds the issue port for 8 of its 16 cycles, so a K-step (32 MFMAs per wave) has room for ~60 other vector
    // instructions per wave before the kernel turns issue-bound (first version of this kernel: ~170 per step, 0.91 us per step against 0.72 for the
    // 12-wave kernel whose compute waves do no address arithmetic).  Hence: everything per-lane that does not change is computed ONCE into a
    // register (rel[], eflags, a_base, fg16, ...), what changes per step or tile is scalar and reaches the memory instructions through the
    // SGPR-offset operand or one add, and constant deltas between a lane's four pixel fragments ride in the LDS instructions' offset fields.
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware tile order: workgroups b and b + 8 share an XCD (L2); give each XCD a contiguous run of tiles
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int G = gridDim.x;
    const int ctiles = p.ctiles, ntiles = p.ptiles * p.ctiles;
    const int tiles_x = p.OW / TW, tiles_y = p.OH / TH;
    const int my_tiles = (ntiles - bid + G - 1) / G;                   // tiles bid, bid + G, ... (the grid never exceeds ntiles)
    const int kchunks = p.Cin >> 6;                                     // (Cin is a multiple of 64: the launcher checks)
    const int Cout = p.Cout, IH = p.IH, IW = p.IW, OW = p.OW;
    const int xs_n = (int)p.xs_n, xs_h = (int)p.xs_h, xs_w = (int)p.xs_w, ws_co = (int)p.ws_co;
    const int ys_n = (int)p.ys_n, ys_h = (int)p.ys_h, ys_w = (int)p.ys_w;
    struct TileC { int c0, tn, y0, x0; };
    auto decode = [&](int tile) -> TileC {
        TileC r;
        const int ct_ = tile % ctiles; int pt_ = tile / ctiles;
        const int tx = pt_ % tiles_x; pt_ /= tiles_x;
        const int ty = pt_ % tiles_y;
        r.c0 = ct_ * 128; r.tn = pt_ / tiles_y; r.y0 = ty * TH; r.x0 = tx * TW;
        return r;
    };
    // tile + G without divisions: G in the mixed radix (ctiles, tiles_x, tiles_y), added with carries (the divisions of decode() cost ~1000 cycles
    // per tile boundary, in every wave, with the matrix pipe idle)
    const TileC gstep = decode(G);                       // (c0 / 128, x0 / TW, y0 / TH, tn) digits of G, scaled like the coordinates
    auto advance = [&](TileC t_) -> TileC {
        t_.c0 += gstep.c0; int carry = t_.c0 >= ctiles * 128; t_.c0 -= carry ? ctiles * 128 : 0;
        t_.x0 += gstep.x0 + (carry ? TW : 0); carry = t_.x0 >= tiles_x * TW; t_.x0 -= carry ? tiles_x * TW : 0;
        t_.y0 += gstep.y0 + (carry ? TH : 0); carry = t_.y0 >= tiles_y * TH; t_.y0 -= carry ? tiles_y * TH : 0;
        t_.tn += gstep.tn + carry;
        return t_;
    };

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
    const int lrow = lane >> 3;
    const int src_k = ((lane & 7) ^ lrow) * 8;          // DMA lane -> (row = 8 piece + lrow, slot = lane & 7), source k-slot = slot ^ (row & 7)
    // taps come in canonical order t = 3 (dy + 1) + (dx + 1) with slab(t) = slab(0) + t * (slab(1) - slab(0)) (the launcher sorts and checks): the halo
    // shift of a tap is a compile-time constant and its weight slab a scalar multiply -- no per-tap tables in registers
    const int W0 = p.tap_slab[0] * (int)p.ws_slab * 2, DW = (p.tap_slab[1] - p.tap_slab[0]) * (int)p.ws_slab * 2;
    auto tap_shift = [](int t) { return (t / 3 - 1) * PW + (t % 3 - 1); };

    // ---- roles for the loads: waves 0-3 stream the weight stages (4 pieces each per step), waves 4-7 the halo (11 pieces each per slice, in its first
    // three steps) and the tile's epilogue parameters.  All eight compute alike; but vmcnt retires in order, and a halo piece that comes from HBM
    // (537 MB of activations do not fit the Infinity Cache) in front of a weight piece would hold up the weight stream's counted waits -- with
    // every wave issuing both, each slice stalled on its six halo steps (first version: 2.3 us per tile, all of it exposed HBM latency).
    const int wrole = wave >> 2, w4 = wave & 3;
    // weight wave: rows 8 w4 + lrow + 32 e, e = 0..3, of the 128-row tile (chmap(R + 32) = chmap(R) + 32).  a_base = byte offset of (row, src_k) inside
    // a tile's rows of a slab; tile, tap, slice and e reach the loads through the scalar offset.  Rows beyond Cout
    // do not exist (Cout is a multiple of 128: the launcher checks -- the scalar offset is not part of the buffer range check).
    const unsigned a_base = (unsigned)(chmap(w4 * 8 + lrow) * ws_co + src_k) * 2u;
    // coordinates of the slice that holds step s + LEAD (the step whose weights are issued in step s); past the last tile the stream stays on it
    // (the loads then fetch bytes nobody reads, from valid addresses)
    int ichunk = 0, itile = bid, ic0 = decode(bid).c0;
    auto dma_w = [&](int tap, int slot) {
        const int soff = W0 + tap * DW + ichunk * 128 + ic0 * ws_co * 2;
#pragma unroll
        for (int e = 0; e < 4; e++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_ptr)(smem + RING0 + slot * STAGE + (w4 + 4 * e) * 1024), 16, a_base, soff + e * 32 * ws_co * 2, 0, 0);
    };
    auto advance_issue_slice = [&]() {                   // the issue stream enters the next slice
        if (++ichunk == kchunks) {
            ichunk = 0;
            if (itile + G < ntiles) { itile += G; ic0 += gstep.c0; if (ic0 >= ctiles * 128) ic0 -= ctiles * 128; }
        }
    };
    // ---- halo pieces of a halo wave: q = 4 k + w4, k = 0..10 (q >= HPIECES: a spare that writes zeros into the dump KiB).  Per lane and piece, once:
    // rel[k] = byte offset of its halo pixel (py, px) and k-slot from the halo's origin; flag bits {py == 0, py == PH - 1, px == 0, px == PW - 1,
    // no pixel} (5 per piece: k < 6 in eflo, the others in efhi).  Per slice: the origin's byte offset and the tile's edge mask (scalars) -- a lane
    // whose flags meet the mask reads zeros.
    constexpr int NHP = 11;
    static_assert(4 * NHP >= HPIECES, "eleven pieces per halo wave cover the halo");
    int rel[NHP]; unsigned eflo = 0, efhi = 0;
#pragma unroll
    for (int k = 0; k < NHP; k++) {
        const int q = 4 * k + w4, pp = q * 8 + lrow;
        const int py = pp / PW, px = pp - py * PW;
        rel[k] = (py * xs_h + px * xs_w + src_k) * 2;
        const unsigned f = (unsigned)(py == 0) | ((unsigned)(py == PH - 1) << 1) | ((unsigned)(px == 0) << 2) | ((unsigned)(px == PW - 1) << 3)
                         | ((unsigned)((q >= HPIECES) | (pp >= NPIX)) << 4);
        if (k < 6) eflo |= f << (5 * k); else efhi |= f << (5 * (k - 6));
    }
    auto dma_halo = [&](auto k_tag, int hbase, unsigned emask, int buf) {
        constexpr int k = decltype(k_tag)::value;
        const int q = 4 * k + w4;
        const bool bad = (((k < 6 ? eflo >> (5 * k) : efhi >> (5 * (k - 6)))) & emask) != 0u;
        const unsigned off = bad ? SBG_OOB_OFFSET : (unsigned)(rel[k] + hbase);
        const unsigned dst = (q < HPIECES) ? (unsigned)(H0 + buf * HALO + q * 1024) : (unsigned)DUMP;      // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(smem + dst), 16, off, 0, 0, 0);
    };
    auto halo_base = [&](const TileC& t_, int chunk) { return (t_.tn * xs_n + (t_.y0 - 1) * xs_h + (t_.x0 - 1) * xs_w + chunk * 64) * 2; };
    auto edge_mask = [&](const TileC& t_) {
        const unsigned m = (unsigned)(t_.y0 == 0) | ((unsigned)(t_.y0 + TH == IH) << 1) | ((unsigned)(t_.x0 == 0) << 2) | ((unsigned)(t_.x0 + TW == IW) << 3) | 16u;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)m);      // (wave-uniform; says so to the compiler)
    };
    // ---- epilogue parameters of tile pc -> parameter buffer `buf` (halo waves): w4 = 0 the noise of the TH x TW tile (row-major, one 16-B piece per lane),
    // w4 = 1 the bias of the tile's 128 channels, w4 = 2 the demodulation coefficients of (image, those channels) (lanes 0-31); w4 = 3 (and every halo
    // wave when `live` is false) writes zeros into the dump KiB so that all halo waves issue one instruction
    auto dma_params = [&](const TileC& pc, bool live, int buf) {
        unsigned dst = (unsigned)DUMP;                   // byte offset inside smem
        unsigned off = SBG_OOB_OFFSET;
        const void* base = p.x;
        if (TAIL && live) {
            if (w4 == 0 && p.noise) {
                constexpr int LPR = TW / 4;
                const int r = lane / LPR, cx = (lane - r * LPR) * 4;
                base = p.noise; dst = (unsigned)(P0 + buf * PARAM);
                off = (unsigned)(pc.tn * (int)p.noise_sn + (pc.y0 + r) * OW + pc.x0 + cx) * 4u;
            } else if ((w4 == 1 && p.bias) || (w4 == 2 && p.oscale)) {
                const int co = pc.c0 + 4 * lane;
                const bool ok = (lane < 32) & (co < Cout);
                base = (w4 == 1) ? (const void*)p.bias : (const void*)p.oscale;
                dst = (unsigned)(P0 + buf * PARAM + (w4 == 1 ? 1024 : 2048));
                off = ok ? (unsigned)((w4 == 1 ? 0 : pc.tn * Cout) + co) * 4u : SBG_OOB_OFFSET;
            }
        }
        __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (lds_void_ptr)(smem + dst), 16, off, 0, 0, 0);
    };

    // ---- MFMA coordinates: wave = (channel half wci) x (pixel quarter wpi), 64 x 64 per wave
    const int wci = wave >> 2, wpi = wave & 3;
    const int wc = wci * 64;
    const int fr = lane & 15, fg = lane >> 4;
    const unsigned offA0 = (unsigned)(wc * 128 + fr * 128 + ((fg ^ (fr & 7)) << 4));       // k half 0; k half 1: ^ 64
    // segment j of this wave = rows / 16-pixel column blocks (2 wpi + (j >> 1), j & 1) of the tile for TW = 32, (4 wpi + j, 0) for TW = 16.  The halo
    // pixel of (segment j, lane fr) is that of segment 0 plus DJ(j) = {0, 16, PW, PW + 16} / {0, PW, 2 PW, 3 PW}: the multiple of 8 rides in the read's
    // offset field, the rest (0 or 2 for TW = 32; 0, 2, 4, 6 for TW = 16) selects one of NB swizzled bases per step
    const int seg_pp0 = ((SEG == 2 ? 2 * wpi : 4 * wpi) + 1) * PW + fr + 1;               // halo pixel of (segment 0, lane fr) for tap shift 0
    const unsigned fg16 = (unsigned)fg << 4;
    constexpr int NB = (SEG == 2) ? 2 : 4;
    const int row0 = (SEG == 2 ? 2 * wpi : 4 * wpi);                                      // tile row of segment 0
    auto seg_dpix = [&](int j) { return row0 * ys_h + (SEG == 2 ? (j >> 1) * ys_h + (j & 1) * 16 * ys_w : j * ys_h); };   // (wave-uniform)
    const int lane_yoff = fr * ys_w + 8 * fg;            // output element offset of (lane's pixel column, lane's channel group) inside a segment
    auto fr4 = [&]() { return (unsigned)(seg_pp0 - (row0 + 1) * PW - 1) * 4u; };          // 4 fr, from a register that lives anyway

    // acc: the tile being computed.  holdp: the finished tile, already through its tail and converted (two 16-bit values per register), drained -- one
    // 16-B store per lane and step -- behind the next tile's first eight steps.  (Keeping the finished tile in fp32 and running the tail in the drain
    // units does not fit: 64 + 64 accumulators + 64 fragment registers + ~45 others spill, and a spill reload in the loop is a vmcnt(0).  Two
    // accumulator sets that swap roles per tile double the loop body and the allocator then shuffles 128 registers at every join.)
    float4_t acc[4][4];                                  // [channel fragment][pixel fragment]
    int4_t holdp[2][4];                                  // [channel group h][pixel fragment] = channels 32 h + 8 fg + 0..7 as 16-bit pairs
    short8_t fa[2][4], fb[2][4];                         // [k half = group][fragment]

    // ---- drain: unit d = (h = d >> 2: channel fragments 2h, 2h + 1 = channels 32 h + 8 fg + 0..7; j = d & 3) of accumulator set DS
    const float alpha = (p.act == SBG_ACT_LRELU) ? p.alpha : (p.act == SBG_ACT_RELU ? 0.f : 1.f);
    const float lsel = alpha <= 1.f ? __builtin_inff() : -__builtin_inff();      // leaky ReLU = med3(u, alpha u, +inf) = max for alpha <= 1, min (-inf) above
    const float cl = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    const float gain = p.gain;
    const bool has_nz = TAIL && p.noise != nullptr, has_b = TAIL && p.bias != nullptr, has_s = TAIL && p.oscale != nullptr;
    int tpar = 0;                                        // parameter buffer of the current tile
    unsigned y_done = 0;                                 // element offset of its origin: n * ys_n + y0 * ys_h + x0 * ys_w + c0 + wc  (wave-uniform)
    auto pack8 = [](const float4_t& lo, const float4_t& hi) {
        short8_t o;
#pragma unroll
        for (int e = 0; e < 4; e++) { o[e] = (short)f32_to_bf16_bits(lo[e]); o[4 + e] = (short)f32_to_bf16_bits(hi[e]); }
        return __builtin_bit_cast(int4_t, o);
    };
    const bool nostore = (p.debug >> 8) & 512;           // timing experiment: the drain units skip their store (wrong results)
    auto drain_unit = [&](auto d_tag) {
        constexpr int d = decltype(d_tag)::value, h = d >> 2, j = d & 3;
        const unsigned yo = (unsigned)opaque_s((int)(y_done + (unsigned)seg_dpix(j) + 32u * h)) + (unsigned)lane_yoff;
        if (!nostore) *reinterpret_cast<int4_t*>((unsigned short*)p.y + yo) = holdp[h][j];
    };
    // the finished tile leaves the accumulators: tail (parameters from the tile's LDS block: they landed before the last step's barrier), conversion.
    // The next tile's first MFMAs start from a zero operand, not from zeroed registers.
    auto retire_tile = [&]() {
        if constexpr (TAIL) {
            const unsigned pb = lds_base + (unsigned)(P0 + tpar * PARAM);
            float nz[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int dsg = (SEG == 2) ? (j >> 1) * TW + (j & 1) * 16 : j * TW;          // pixel of segment j relative to segment 0, inside the tile
                lds_read32f(nz[j], pb + (unsigned)((row0 * TW + dsg) * 4) + fr4());
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float4_t b0, b1, s0, s1;
                const unsigned cb = pb + 1024u + (unsigned)((wc + 32 * h) * 4) + fg16 * 2u;
                lds_read128f(b0, cb); lds_read128f(b1, cb + 16u); lds_read128f(s0, cb + 1024u); lds_read128f(s1, cb + 1040u);
                lds_wait_params(nz, b0, b1, s0, s1);
                if (!has_b) { b0 = float4_t{0.f, 0.f, 0.f, 0.f}; b1 = b0; }
                if (!has_s) { s0 = float4_t{1.f, 1.f, 1.f, 1.f}; s1 = s0; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float nzj = has_nz ? nz[j] : 0.f;
                    float4_t lo = acc[2 * h][j] * s0 + (nzj + b0), hi = acc[2 * h + 1][j] * s1 + (nzj + b1);
                    const float4_t tl_ = lo * alpha, th_ = hi * alpha;
#pragma unroll
                    for (int e = 0; e < 4; e++) { lo[e] = __builtin_amdgcn_fmed3f(lo[e], tl_[e], lsel); hi[e] = __builtin_amdgcn_fmed3f(hi[e], th_[e], lsel); }
                    lo = lo * gain; hi = hi * gain;
#pragma unroll
                    for (int e = 0; e < 4; e++) { lo[e] = __builtin_amdgcn_fmed3f(lo[e], -cl, cl); hi[e] = __builtin_amdgcn_fmed3f(hi[e], -cl, cl); }
                    holdp[h][j] = pack8(lo, hi);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int j = 0; j < 4; j++) holdp[h][j] = pack8(acc[2 * h][j], acc[2 * h + 1][j]);
        }
    };
    constexpr int ST = 1;                                // store instructions of a drain unit

    // ---- fragment reads of a step (ring slot, halo buffer, tap shift): both k halves share the step's NB swizzled bases
    //   base[k] = halo buffer + 128 t0 + ((fg ^ ((t0 + 2k) & 7)) << 4),  t0 = seg_pp0 + shift;  k half 1: ^ 64
    unsigned bbase[NB];
    auto step_bases = [&](unsigned hb, int shift) {
        const unsigned t0 = (unsigned)(seg_pp0 + shift);
        const unsigned row = lds_base + hb + (t0 << 7), u = t0 << 4;
#pragma unroll
        for (int k = 0; k < NB; k++) bbase[k] = row + (fg16 ^ ((u + 32u * k) & 0x70u));
    };
    auto read_group = [&](auto g_tag, int slot) {
        constexpr int g = decltype(g_tag)::value;
        const unsigned aa = lds_base + (unsigned)(RING0 + slot * STAGE) + (g ? (offA0 ^ 64u) : offA0);
        lds_read128<0>(fa[g][0], aa); lds_read128<2048>(fa[g][1], aa); lds_read128<4096>(fa[g][2], aa); lds_read128<6144>(fa[g][3], aa);
        if constexpr (SEG == 2) {
            const unsigned b0 = g ? (bbase[0] ^ 64u) : bbase[0], b1 = g ? (bbase[1] ^ 64u) : bbase[1];
            lds_read128<0>(fb[g][0], b0); lds_read128<16 * 128>(fb[g][1], b0);
            lds_read128<PW * 128>(fb[g][2], b1); lds_read128<(PW + 16) * 128>(fb[g][3], b1);
        } else {
            lds_read128<0>(fb[g][0], g ? (bbase[0] ^ 64u) : bbase[0]);
            lds_read128<PW * 128>(fb[g][1], g ? (bbase[1] ^ 64u) : bbase[1]);
            lds_read128<2 * PW * 128>(fb[g][2], g ? (bbase[2] ^ 64u) : bbase[2]);
            lds_read128<3 * PW * 128>(fb[g][3], g ? (bbase[3] ^ 64u) : bbase[3]);
        }
    };

    int tile = bid;
    TileC cur = decode(tile), nxt = advance(cur);
    // ---- prologue: halo of slice 0 (six slots per wave), weights of steps 0, 1, 2; then the first step's first fragment group
    {
        const int hb0 = halo_base(cur, 0); const unsigned em0 = edge_mask(cur);
        if (wrole == 1) { sbg_static_for<NHP>([&](auto kt) { dma_halo(kt, hb0, em0, 0); }); wait_vm<0>(); }
        else {
#pragma unroll
            for (int k = 0; k < LEAD; k++) dma_w(k, k);
            wait_vm<4>();                                // everything but the weights of step 2
        }
    }
    __builtin_amdgcn_s_barrier();
    step_bases((unsigned)H0, opaque_s(tap_shift(0)));
    read_group(std::integral_constant<int, 0>{}, 0);

    int slot = 0, par = 0;                               // ring slot of the current step, halo buffer of the current slice
    int st1 = 0, st2 = 0;                                // store instructions issued by the drain unit of the previous step / the one before
    bool pending = false;                                // the other accumulator set holds a finished tile

    // timing experiment (bits 512 + 1024: no output is stored; wave 0 of workgroup 0 writes its clock at every barrier into the head of y)
    const bool stamp = ((p.debug >> 8) & 1024) && nostore && blockIdx.x == 0 && wave == 0 && lane == 0;
    unsigned long long* stamps = (unsigned long long*)p.y; int sidx = 0;
    // one tile; the previous one drains from `hold` behind its first eight steps
    auto run_tile = [&]() {
        for (int chunk = 0; chunk < kchunks; chunk++) {
            // the slice after this one (past the last tile: this tile's first slice again -- valid addresses, nobody reads the bytes): its halo is
            // staged during this slice's first six steps
            int hbase; unsigned emask;
            if (chunk + 1 < kchunks) { hbase = halo_base(cur, chunk + 1); emask = edge_mask(cur); }
            else {
                const TileC hc = (tile + G < ntiles) ? nxt : cur;
                hbase = halo_base(hc, 0); emask = edge_mask(hc);
            }
            const bool last_chunk = chunk == kchunks - 1;
            const bool first_chunk = chunk == 0;
            const bool drain_here = pending && first_chunk;
            const unsigned hb = (unsigned)(H0 + par * HALO), hb_next = (unsigned)(H0 + (par ^ 1) * HALO);
            sbg_static_for<NT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                // A: second k half of this step (its bases were set with the first half's, in E of the previous step)
                read_group(std::integral_constant<int, 1>{}, slot);
                __builtin_amdgcn_sched_barrier(0);
                // B: first k half; the tile's very first products start from zero
                lds_wait8<8>(fa[0], fb[0]);
                if (t == 0 && first_chunk) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[i][j] = Mfma<MF>::run(fa[0][i], fb[0][j], float4_t{0.f, 0.f, 0.f, 0.f});
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[i][j] = Mfma<MF>::run(fa[0][i], fb[0][j], acc[i][j]);
                }
                // the step's loads, behind those MFMAs: weights of step s + 3 -> the slot step s - 1 used; a halo piece of the next slice / the tile's parameters
                if constexpr (t == 6) advance_issue_slice();           // step s + 3 is the first step of the next slice
                if (wrole == 0) dma_w((t + LEAD) % NT, (slot + LEAD) & (NRING - 1));
                else {
                    if constexpr (t <= 2) {
                        const int hb_ = opaque_s(hbase); const unsigned em_ = (unsigned)opaque_s((int)emask);
                        sbg_static_for<(t < 2 ? 4 : NHP - 8)>([&](auto kt) { dma_halo(std::integral_constant<int, 4 * t + decltype(kt)::value>{}, hb_, em_, par ^ 1); });
                    }
                    if constexpr (t == 6) dma_params(cur, last_chunk, tpar);
                }
                __builtin_amdgcn_sched_barrier(0);
                // C: this wave is done with the stage of step s
                lds_wait8<0>(fa[1], fb[1]);
                // D: the loads of step s - 2 (the weights of step s + 1, every halo piece and parameter before them) have landed on all waves
                // (weight waves: the weights of step s + 1 were issued in step s - 2; halo waves: the next slice's halo and the parameters, before the
                // slice's last barrier; the drain stores of the two previous steps ride in the same counters)
                if (wrole == 0) wait_vm_plus<8>(st1 + st2);
                else if constexpr (t == 8) wait_vm_plus<0>(st1 + st2);
                if (stamp) { if (sidx < 400) stamps[sidx] = __builtin_amdgcn_s_memtime(); sidx++; }       // before the barrier: this wave's arrival
                __builtin_amdgcn_s_barrier();
                if (stamp) { if (sidx < 400) stamps[sidx] = __builtin_amdgcn_s_memtime(); sidx++; }       // behind it: the last wave's arrival
                __builtin_amdgcn_sched_barrier(0);
                // E: first k half of the NEXT step (the first step of the next TILE gets its reads behind the retiring of this one: the tail's
                // parameters and temporaries then have the 32 fragment registers to themselves)
                if (t < 8 || !last_chunk) {
                    constexpr int tn_ = (t + 1) % NT;
                    step_bases(tn_ == 0 ? hb_next : hb, opaque_s(tap_shift(tn_)));
                    read_group(std::integral_constant<int, 0>{}, (slot + 1) & (NRING - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                // F: second k half, then one drain unit of the finished tile
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = Mfma<MF>::run(fa[1][i], fb[1][j], acc[i][j]);
                st2 = st1; st1 = 0;
                if constexpr (t < 8) {
                    if (drain_here) { drain_unit(std::integral_constant<int, t>{}); st1 = nostore ? 0 : ST; }
                }
                slot = (slot + 1) & (NRING - 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (drain_here) pending = false;
            par ^= 1;
        }
        // the tile is complete: it moves to `holdp` and drains behind the next tile
        retire_tile();
        step_bases((unsigned)(H0 + par * HALO), opaque_s(tap_shift(0)));
        read_group(std::integral_constant<int, 0>{}, slot);
        pending = true; tpar ^= 1;
        y_done = (unsigned)(cur.tn * ys_n + cur.y0 * ys_h + cur.x0 * ys_w + cur.c0 + wc);
        tile += G;
        cur = nxt; nxt = advance(nxt);
    };

    for (int ti = 0; ti < my_tiles; ti++) run_tile();
    wait_vm<0>();                                        // spare loads of the last steps: nothing of this workgroup's LDS may be written after it ends
    __builtin_amdgcn_s_waitcnt(0xC07F);                  // the read-ahead of the step that does not exist
    if (pending) sbg_static_for<8>([&](auto dt) { drain_unit(dt); });     // the last tile
}

template <class MF, int TH, int TW>
static int launch_halo8(ConvArgs& a, unsigned x_bytes, unsigned w_bytes, bool tail, hipStream_t stream)
{
    constexpr int HPIECES = ((TH + 2) * (TW + 2) + 7) / 8;
    constexpr int lds = 4 * 128 * 128 + 2 * HPIECES * 1024 + 2 * 3072 + 1024;
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
    if (nblk > ncu) nblk = ncu;     // persistent: one workgroup per CU, each walks tiles b, b + grid, ...
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * a.P * a.Cout * (double)a.Cin * a.ntaps,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * a.P * (double)a.Cout,
                      {a.P, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, 5128256});      // 5xxxxxx = conv_halo8_kernel (profiles/summarize.py)
#define SBG_HALO8_LAUNCH(YDT, TAIL) do { auto kern = conv_halo8_kernel<MF, TH, TW, YDT, TAIL>; \
        if (!SBG_RAISE_LDS_ONCE(kern, lds)) return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds); \
        SBG_LAUNCH(kern, dim3((unsigned)nblk), dim3(512), lds, stream, a, x_bytes, w_bytes); } while (0)
    if (tail) SBG_HALO8_LAUNCH(SBG_BF16, true); else SBG_HALO8_LAUNCH(SBG_BF16, false);
#undef SBG_HALO8_LAUNCH
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

// Returns SBG_OK / an error, or -1 when the launch is not one this kernel covers (the caller then uses conv_halo_ld_kernel).  The caller has
// established: stride 1, nine taps with |offset| <= 1, output grid == input grid, Cout > 64, enough tiles to fill the chip.
int sbg_conv_halo8_dispatch(ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, hipStream_t stream)
{
    static const char* off = sbg_env("SBG_CONV_NO_HALO8");
    if (off || !((a.debug >> 8) & 256)) return -1;                      // opt-in (experiment bit 256): see the header -- not faster than the 12-wave kernel yet
    if (a.accumulate || a.ksplit > 1 || a.ydtype != SBG_BF16 || a.nphase > 1) return -1;      // (fp32 outputs: the few 4x4 .. 8x8 fp32 blocks stay with the 12-wave kernel)
    if ((a.Cout & 127) || (a.Cin & 63) || ((((uintptr_t)a.y) & 15) != 0) || (((a.ys_n | a.ys_h | a.ys_w) & 7) != 0)) return -1;
    if ((int64_t)a.N * a.ys_n >= (1ll << 31) || a.xs_n * (int64_t)a.N >= (1ll << 30)) return -1;          // 32-bit element offsets in the kernel
    const bool plain = (a.act <= SBG_ACT_LINEAR) && a.gain == 1.f && a.clamp < 0.f && !a.bias && !a.noise && !a.oscale;
    if (!plain) {
        const bool ok = ((((uintptr_t)a.oscale) & 15) == 0) && ((((uintptr_t)a.bias) & 15) == 0)
                        && (!a.noise || (((((uintptr_t)a.noise) & 15) == 0) && (a.noise_sn & 3) == 0 && (a.OW & 3) == 0
                                         && (int64_t)a.N * (a.noise_sn > 0 ? a.noise_sn : 0) + (int64_t)a.OH * a.OW < (1ll << 28)))
                        && (int64_t)a.N * a.Cout < (1ll << 28);
        if (!ok) return -1;
    }
    {   // canonical tap order t = 3 (dy + 1) + (dx + 1); the slabs must then be an arithmetic sequence (as stored, or flipped)
        for (int u = 0; u < 9; u++)
            for (int v = u + 1; v < 9; v++)
                if (a.tap_dy[v] < a.tap_dy[u] || (a.tap_dy[v] == a.tap_dy[u] && a.tap_dx[v] < a.tap_dx[u])) {
                    std::swap(a.tap_dy[u], a.tap_dy[v]); std::swap(a.tap_dx[u], a.tap_dx[v]); std::swap(a.tap_slab[u], a.tap_slab[v]);
                }
        for (int t = 0; t < 9; t++) {
            if (a.tap_dy[t] != t / 3 - 1 || a.tap_dx[t] != t % 3 - 1) return -1;
            if (a.tap_slab[t] != a.tap_slab[0] + t * (a.tap_slab[1] - a.tap_slab[0])) return -1;
        }
    }
    const unsigned xb = (unsigned)x_bytes, wb = (unsigned)w_bytes;
    if (a.OW % 32 == 0 && a.OH % 8 == 0)
        return bf16 ? launch_halo8<bf16_mfma, 8, 32>(a, xb, wb, !plain, stream) : launch_halo8<f16_mfma, 8, 32>(a, xb, wb, !plain, stream);
    if (a.OW % 16 == 0 && a.OH % 16 == 0)
        return bf16 ? launch_halo8<bf16_mfma, 16, 16>(a, xb, wb, !plain, stream) : launch_halo8<f16_mfma, 16, 16>(a, xb, wb, !plain, stream);
    return -1;
}
