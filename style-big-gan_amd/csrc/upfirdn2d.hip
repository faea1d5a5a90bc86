// upfirdn2d.hip -- pad -> zero-insert upsample -> 2-D FIR -> decimate, any up/down/pad/filter/strides.
//
// Semantics: stylegan2ada/torch_utils/ops/upfirdn2d.py:120-208 (`upfirdn2d`, `_upfirdn2d_ref`) and the output-size
// rule of upfirdn2d.cpp:32-33.  Per axis, with u the zero-inserted + padded signal,
//     y[o] = gain * sum_k u[o*down + k] * g[k],   g = f (flip_filter) or reversed f (default: true convolution)
// and u[X] is x[(X - pad0) / up] when (X - pad0) is a non-negative multiple of `up` inside the input, else 0; so only
// taps k == (pad0 - o*down) mod up contribute and the kernels below visit exactly those (no multiplies by inserted zeros).
//
// gfx950 design: HBM-bound streaming op (algorithmic bytes = (numel_in + numel_out) * sizeof(T)).
//  * channel-minor activations (the layout the conv kernels produce): one lane = 8 channels (16 B) of one output pixel,
//    consecutive lanes walk channels then x, so every tap is a fully coalesced 16-B-per-lane load and the 4x4 / 2x2
//    effective footprint is re-used out of L1/L2 (each input line is touched by <= fh*fw/(up*up) neighbouring lanes).
//  * planar (NCHW) activations: one lane = one output pixel of one channel, lanes walk x (coalesced along rows).
//  * the filter lives in LDS (<= 1024 taps) and is read with wave-uniform addresses (broadcast, conflict-free).
#include "sbg_common.h"
#include "lds_asm.h"
#include <cstdlib>

namespace {

struct UpfirdnArgs {
    const void* x; const float* f; void* y;
    int upx, upy, downx, downy, padx0, pady0, flip; float gain;
    int inW, inH, C, N; int64_t isx, isy, isc, isn;
    int fw, fh, fsx, fsy;
    int outW, outH; int64_t osx, osy, osc, osn;
    int64_t total;    // number of lane work items
    // fused tail (matrix-core FIR only): y = clamp(act(fir * gain * oscale[n, c] + noise[n, pixel] + bias[c]) * act_gain)
    const float* oscale; const float* noise; int64_t noise_sn; const float* bias;
    int tail; float alpha, act_gain, clamp;
    // backward tail (sliding-window matrix-core FIR only; tail == 2): the result is the gradient w.r.t. the OUTPUT y of a bias_act (`dact_y`, saved,
    // same geometry as this launch's output); the kernel multiplies by the activation's slope at y -- dact_gpos for y > 0, dact_gneg below, zero on the
    // clamp rails |y| >= dact_rail -- and adds up the result per channel (the bias gradient): dact_part[workgroup][64]
    const void* dact_y; float* dact_part; float dact_gpos, dact_gneg, dact_rail;
    const float* post;      // forward tail only: the finished value is multiplied by post[n, c] (the style modulation of the layer that reads it next)
};

#define SBG_UPFIRDN_MAX_LDS_TAPS 1024

static __device__ __forceinline__ int pos_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

template <class T, int VEC>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirdnArgs p)
{
    __shared__ float sf[SBG_UPFIRDN_MAX_LDS_TAPS];
    const int ntaps = p.fw * p.fh;
    const bool f_lds = ntaps <= SBG_UPFIRDN_MAX_LDS_TAPS;
    if (f_lds) {
        // stage the filter in visiting order: sf[ky*fw + kx] multiplies u[.. + ky][.. + kx]
        for (int t = threadIdx.x; t < ntaps; t += blockDim.x) {
            int ky = t / p.fw, kx = t - ky * p.fw;
            int fy = p.flip ? ky : p.fh - 1 - ky, fx = p.flip ? kx : p.fw - 1 - kx;
            sf[t] = p.f[fy * p.fsy + fx * p.fsx];
        }
        __syncthreads();
    }
    const T* px = (const T*)p.x; T* py = (T*)p.y;
    const int cvecs = (VEC == 8) ? (p.C >> 3) : p.C;
    const bool cminor = (VEC == 8) || (p.isc == 1 && p.C > 1);
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < p.total; idx += step) {
        // decompose the work item: channel-minor -> (c fastest, then x, y, n); planar -> (x fastest, then y, c, n)
        int c, ox, oy, n;
        int64_t r = idx;
        if (cminor) { c = (int)(r % cvecs); r /= cvecs; ox = (int)(r % p.outW); r /= p.outW; oy = (int)(r % p.outH); n = (int)(r / p.outH); }
        else        { ox = (int)(r % p.outW); r /= p.outW; oy = (int)(r % p.outH); r /= p.outH; c = (int)(r % cvecs); n = (int)(r / cvecs); }
        if (VEC == 8) c <<= 3;

        const int baseX = ox * p.downx - p.padx0, baseY = oy * p.downy - p.pady0;   // u-coordinate of tap 0, in input*up units
        const int kx0 = pos_mod(-baseX, p.upx), ky0 = pos_mod(-baseY, p.upy);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) acc[j] = 0.f;
        const T* xin = px + n * p.isn + c * p.isc;
        for (int ky = ky0; ky < p.fh; ky += p.upy) {
            const int iy = (baseY + ky) / p.upy;       // exact: (baseY + ky) is a multiple of upy
            if (iy < 0 || iy >= p.inH) continue;
            for (int kx = kx0; kx < p.fw; kx += p.upx) {
                const int ix = (baseX + kx) / p.upx;
                if (ix < 0 || ix >= p.inW) continue;
                float fv;
                if (f_lds) fv = sf[ky * p.fw + kx];
                else { int fy = p.flip ? ky : p.fh - 1 - ky, fx = p.flip ? kx : p.fw - 1 - kx; fv = p.f[fy * p.fsy + fx * p.fsx]; }
                const T* src = xin + iy * p.isy + ix * p.isx;
                if (VEC == 8) {
                    float v[8];
                    Vec8<T>::ld(src, v);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += v[j] * fv;
                } else {
                    acc[0] += Elem<T>::ld(src) * fv;
                }
            }
        }
        T* dst = py + n * p.osn + c * p.osc + oy * p.osy + ox * p.osx;
        if (VEC == 8) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = acc[j] * p.gain;
            Vec8<T>::st(dst, o);
        } else {
            Elem<T>::st(dst, acc[0] * p.gain);
        }
    }
}


// Register-blocked FIR for the hot case: up == down == 1 (the 4x4 low-pass after every transposed / before every strided
// convolution, and their gradients), channel-minor, 8 channels per lane.  Each lane produces a TY x TX patch of output
// pixels from a (TY + fh - 1) x (TX + fw - 1) input window held in registers: for the 4x4 filter that is 35 16-B loads per 8
// outputs (4.4 per output instead of 16; an 8-wide patch measured slower), every load 256-B-contiguous across the 16 lanes that share a pixel.
#define FIR_TX 4
#define FIR_TY 2
#define FIR_MAXF 8
template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_fir_kernel(UpfirdnArgs p, int xblocks, int yblocks)
{
    __shared__ float sf[FIR_MAXF * FIR_MAXF];
    const int ntaps = p.fw * p.fh;
    for (int t = threadIdx.x; t < ntaps; t += blockDim.x) {
        int ky = t / p.fw, kx = t - ky * p.fw;
        int fy = p.flip ? ky : p.fh - 1 - ky, fx = p.flip ? kx : p.fw - 1 - kx;
        sf[t] = p.f[fy * p.fsy + fx * p.fsx] * p.gain;
    }
    __syncthreads();
    const T* px = (const T*)p.x; T* py = (T*)p.y;
    const int cvecs = p.C >> 3;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < p.total; idx += step) {
        int64_t r = idx;
        const int c = (int)(r % cvecs) << 3; r /= cvecs;
        const int xb = (int)(r % xblocks); r /= xblocks;
        const int yb = (int)(r % yblocks); const int n = (int)(r / yblocks);
        const int ox0 = xb * FIR_TX, oy0 = yb * FIR_TY;
        float acc[FIR_TY][FIR_TX][8];
#pragma unroll
        for (int a = 0; a < FIR_TY; a++)
#pragma unroll
            for (int b = 0; b < FIR_TX; b++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[a][b][e] = 0.f;
        const T* xin = px + n * p.isn + c;
        for (int wy = 0; wy < FIR_TY + p.fh - 1; wy++) {           // input window rows
            const int iy = oy0 + wy - p.pady0;
            if ((unsigned)iy >= (unsigned)p.inH) continue;
            for (int wx = 0; wx < FIR_TX + p.fw - 1; wx++) {
                const int ix = ox0 + wx - p.padx0;
                if ((unsigned)ix >= (unsigned)p.inW) continue;
                float v[8];
                Vec8<T>::ld(xin + iy * p.isy + ix * p.isx, v);
#pragma unroll
                for (int a = 0; a < FIR_TY; a++) {
                    const int ky = wy - a;
                    if (ky < 0 || ky >= p.fh) continue;
#pragma unroll
                    for (int b = 0; b < FIR_TX; b++) {
                        const int kx = wx - b;
                        if (kx < 0 || kx >= p.fw) continue;
                        const float fv = sf[ky * p.fw + kx];
#pragma unroll
                        for (int e = 0; e < 8; e++) acc[a][b][e] += v[e] * fv;
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < FIR_TY; a++) {
            const int oy = oy0 + a;
            if (oy >= p.outH) continue;
#pragma unroll
            for (int b = 0; b < FIR_TX; b++) {
                const int ox = ox0 + b;
                if (ox >= p.outW) continue;
                Vec8<T>::st(py + n * p.osn + c + oy * p.osy + ox * p.osx, acc[a][b]);
            }
        }
    }
}

// The register-blocked kernel with the filter size known at compile time (4 x 4: the [1, 3, 3, 1] low-pass of every up / down
// layer): the window loops unroll, the taps live in registers and the tap-range tests fold away.  The bounds tests stay branches
// around each load, which also keeps the loads in program order (bounded live set).
template <class T, int FH, int FW>
__global__ __launch_bounds__(256) void upfirdn2d_fir_fixed_kernel(UpfirdnArgs p, int xblocks, int yblocks)
{
    float fv[FH][FW];                                   // visiting order: fv[ky][kx] multiplies the window pixel (.. + ky, .. + kx)
#pragma unroll
    for (int ky = 0; ky < FH; ky++)
#pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            const int fy = p.flip ? ky : FH - 1 - ky, fx = p.flip ? kx : FW - 1 - kx;
            fv[ky][kx] = p.f[fy * p.fsy + fx * p.fsx] * p.gain;
        }
    const T* px = (const T*)p.x; T* py = (T*)p.y;
    const int cvecs = p.C >> 3;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < p.total; idx += step) {
        int64_t r = idx;
        const int c = (int)(r % cvecs) << 3; r /= cvecs;
        const int xb = (int)(r % xblocks); r /= xblocks;
        const int yb = (int)(r % yblocks); const int n = (int)(r / yblocks);
        const int ox0 = xb * FIR_TX, oy0 = yb * FIR_TY;
        float acc[FIR_TY][FIR_TX][8];
#pragma unroll
        for (int a = 0; a < FIR_TY; a++)
#pragma unroll
            for (int b = 0; b < FIR_TX; b++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[a][b][e] = 0.f;
        const T* xin = px + n * p.isn + c;
#pragma unroll
        for (int wy = 0; wy < FIR_TY + FH - 1; wy++) {
            const int iy = oy0 + wy - p.pady0;
            if ((unsigned)iy >= (unsigned)p.inH) continue;
#pragma unroll
            for (int wx = 0; wx < FIR_TX + FW - 1; wx++) {
                const int ix = ox0 + wx - p.padx0;
                if ((unsigned)ix >= (unsigned)p.inW) continue;
                float v[8];
                Vec8<T>::ld(xin + iy * p.isy + ix * p.isx, v);
#pragma unroll
                for (int a = 0; a < FIR_TY; a++) {
                    const int ky = wy - a;
                    if (ky < 0 || ky >= FH) continue;
#pragma unroll
                    for (int b = 0; b < FIR_TX; b++) {
                        const int kx = wx - b;
                        if (kx < 0 || kx >= FW) continue;
#pragma unroll
                        for (int e = 0; e < 8; e++) acc[a][b][e] += v[e] * fv[ky][kx];
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < FIR_TY; a++) {
            const int oy = oy0 + a;
            if (oy >= p.outH) continue;
#pragma unroll
            for (int b = 0; b < FIR_TX; b++) {
                const int ox = ox0 + b;
                if (ox >= p.outW) continue;
                Vec8<T>::st(py + n * p.osn + c + oy * p.osy + ox * p.osx, acc[a][b]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-core FIR for the hot case (up == down == 1, 4 x 4 taps, 16-bit channel-minor tensors, C % 64 == 0, taps exactly
// representable in the tensor dtype).  The register-blocked kernels above are VALU-bound (16 fp32 FMAs + unpacking per output
// element: 2.5 TB/s); here a row of 16 output pixels x 16 channels is ONE accumulator tile,
//     D[c][ox] += sum_ix  In[iy = oy + ky][ix][c] * T_ky[ix][ox],      T_ky[ix][ox] = f[ky][ix - ox]  (Toeplitz, 32 x 16)
// i.e. four v_mfma_f32_16x16x32 per tile (one per filter row), 0.4 % of the chip's matrix rate, and the VALU only converts
// and stores.  The input window of an 8 x 32 output tile x 64 channels (11 rows x 40 pixels) is staged once by LDS-DMA in
// whole 128-B lines; the A operand (channel x 32 consecutive pixels) is read with the transposing ds_read_b64_tr_b16.
// Zero-padding = out-of-range DMA offsets; LDS pixels 40..47 of each row are only ever multiplied by zero taps and are
// cleared once so that they stay finite.  `gain` is applied in fp32 on the accumulator.
typedef __attribute__((address_space(3))) void* fir_lds_ptr;
typedef __attribute__((address_space(3))) short4_t* fir_lds_s4_ptr;
#define SBG_FIR_OOB 0x80000000u

template <class T> struct FirMfma;
template <> struct FirMfma<bf16_s> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ short bits(float v) { return (short)f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float from_bits(unsigned short b) { return bf16_bits_to_f32(b); }
};
template <> struct FirMfma<f16_s> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ short bits(float v) { return (short)f32_to_f16_bits(v); }
    static __device__ __forceinline__ float from_bits(unsigned short b) { return f16_bits_to_f32(b); }
};

template <class T, int RPW>      // RPW = output rows per wave: 2 (8 x 32 tile, 68 KB of LDS, 2 workgroups per CU) or 1 (4 x 32 tile, 43 KB, 3 per CU)
__global__ __launch_bounds__(256, RPW == 1 ? 3 : (RPW == 2 ? 2 : 1)) void upfirdn2d_fir_mfma_kernel(UpfirdnArgs p, unsigned x_bytes, int tiles_x, int tiles_y, int cblocks)
{
    constexpr int TY = 4 * RPW, TX = 32, FH = 4, FW = 4;
    constexpr int WY = TY + FH - 1;                    // 11 window rows
    constexpr int WXL = 48;                            // LDS row pitch in pixels: the K window of the second 16-pixel segment ends at 16 + 32
    constexpr int PPR = 5;                             // DMA pieces (8 pixels) per window row: pixels 0..39 (35 needed)
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int cb0 = b % cblocks; b /= cblocks;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int ox0 = tx * TX, oy0 = ty * TY;

    // clear pixels 40..47 of every window row (8 x 128 B each): 11 rows x 64 chunks of 16 B
    for (int i = tid; i < WY * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        *reinterpret_cast<float4_t*>(fsm + ((r * WXL + 40) * 128) + c * 16) = float4_t{0.f, 0.f, 0.f, 0.f};
    }
    // stage the window: piece = 8 pixels x 128 B, lane -> (pixel row = lane / 8, LDS chunk = lane % 8); the 16-B chunk pairs of LDS
    // row R are XOR-swizzled by (R >> 1) & 3 on the source side and in the transposing reads (as in conv_wgrad.hip)
    {
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
        const int drow = lane >> 3, dchunk = lane & 7;
        for (int piece = wave; piece < WY * PPR; piece += 4) {
            const int r = piece / PPR, j = piece - r * PPR;
            const int R = r * WXL + j * 8 + drow;
            const int sc = (((dchunk >> 1) ^ ((R >> 1) & 3)) << 1) | (dchunk & 1);
            const int iy = oy0 - p.pady0 + r, ix = ox0 - p.padx0 + j * 8 + drow;
            const unsigned okm = 0u - (unsigned)(((unsigned)iy < (unsigned)p.inH) & ((unsigned)ix < (unsigned)p.inW) & (j * 8 + drow < TX + FW - 1));
            const unsigned real = (unsigned)(n * (int)p.isn + iy * (int)p.isy + ix * (int)p.isx + cb0 * 64 + sc * 8) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (fir_lds_ptr)(fsm + (r * WXL + j * 8) * 128), 16, (real & okm) | (SBG_FIR_OOB & ~okm), 0, 0, 0);
        }
    }
    // Toeplitz operand, constant per lane: B lane (n = output pixel = lane & 15, k-group g = lane >> 4) element j multiplies the
    // input pixel kpix(g, j) = j < 4 ? 4g + j : 16 + 4g + (j - 4)  -- the k order of the transposing reads below
    const int fi = lane & 15, fg = lane >> 4, fq = fi >> 2, fp = fi & 3;
    short8_t bt[FH];
#pragma unroll
    for (int ky = 0; ky < FH; ky++) {
        const int fy = p.flip ? ky : FH - 1 - ky;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int kpix = j < 4 ? 4 * fg + j : 16 + 4 * fg + (j - 4);
            const int kx = kpix - fi;
            float v = 0.f;
            if (kx >= 0 && kx < FW) v = p.f[fy * p.fsy + (p.flip ? kx : FW - 1 - kx) * p.fsx];
            bt[ky][j] = FirMfma<T>::bits(v);
        }
    }
    // per-lane byte offsets of the transposing A reads (rows = pixels 4g + q and 16 + 4g + q of the K window, 16 channels at `col`)
    auto frag_off = [&](int Rrel, int col) {
        const int chunk = (col >> 3) + (fp >> 1);
        const int sw = (((chunk >> 1) ^ ((Rrel >> 1) & 3)) << 1) | (chunk & 1);
        return Rrel * 128 + sw * 16 + (fp & 1) * 8;
    };
    int offA[4];
#pragma unroll
    for (int c = 0; c < 4; c++) offA[c] = frag_off(4 * fg + fq, c * 16);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float4_t acc[RPW][2][4];                             // [output row of this wave][16-pixel segment][16-channel block]
#pragma unroll
    for (int a = 0; a < RPW; a++)
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[a][sg][c] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rl = 0; rl < RPW + FH - 1; rl++) {          // the window rows behind this wave's output rows
        const unsigned char* rowp = fsm + ((RPW * wave + rl) * WXL) * 128;
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const unsigned char* q = rowp + sg * 16 * 128 + offA[c];
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fir_lds_s4_ptr)q);
                const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fir_lds_s4_ptr)(q + 16 * 128));
                const short8_t fa = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int a = 0; a < RPW; a++) {
                    const int ky = rl - a;
                    if (ky < 0 || ky >= FH) continue;
                    acc[a][sg][c] = FirMfma<T>::run(fa, bt[ky], acc[a][sg][c]);
                }
            }
    }
    // D tile: lane holds channels cb*16 + 4 fg + {0..3} of output pixel seg*16 + (lane & 15).  v_permlane16_swap_b32 exchanges the
    // odd 16-lane rows of one channel block with the even rows of the next, after which lane group fg holds EIGHT consecutive
    // channels -- block c + (fg & 1), offset 8 (fg >> 1) -- and stores 16 B instead of 2 x 8 B (the store path is issue-bound).
    T* yb = (T*)p.y + n * p.osn + cb0 * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
    // fused tail, applied in fp32 before the exchange: this lane's channels are cb0*64 + 16 c + 4 fg + {0..3}
    float4_t t_scale[4], t_bias[4], t_post[4];
    const float t_alpha = p.tail ? p.alpha : 1.f, t_gain = p.tail ? p.act_gain : 1.f, t_cl = (p.tail && p.clamp >= 0.f) ? p.clamp : __builtin_inff();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int ch = cb0 * 64 + c * 16 + 4 * fg;
        t_scale[c] = float4_t{p.gain, p.gain, p.gain, p.gain};
        t_bias[c] = float4_t{0.f, 0.f, 0.f, 0.f};
        t_post[c] = float4_t{1.f, 1.f, 1.f, 1.f};
        if (p.tail && p.oscale) t_scale[c] *= *reinterpret_cast<const float4_t*>(p.oscale + (int64_t)n * p.C + ch);
        if (p.tail && p.bias)   t_bias[c] = *reinterpret_cast<const float4_t*>(p.bias + ch);
        if (p.tail == 1 && p.post) t_post[c] = *reinterpret_cast<const float4_t*>(p.post + (int64_t)n * p.C + ch);
    }
#pragma unroll
    for (int a = 0; a < RPW; a++) {
        const int oy = oy0 + RPW * wave + a;
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            const int ox = ox0 + sg * 16 + fi;
            const bool ok = oy < p.outH && ox < p.outW;
            T* dst = yb + oy * p.osy + ox * p.osx;
            const float nz = (p.tail && p.noise && ok) ? p.noise[n * p.noise_sn + (int64_t)oy * p.outW + ox] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float4_t v = acc[a][sg][c] * t_scale[c] + (t_bias[c] + nz);
#pragma unroll
                for (int e = 0; e < 4; e++) { float u = v[e]; u = (u > 0.f) ? u : u * t_alpha; v[e] = __builtin_amdgcn_fmed3f(u * t_gain, -t_cl, t_cl) * t_post[c][e]; }
                acc[a][sg][c] = v;
            }
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const float4_t va = acc[a][sg][c], vb = acc[a][sg][c + 1];
                const unsigned a0 = (unsigned)(unsigned short)FirMfma<T>::bits(va[0]) | ((unsigned)(unsigned short)FirMfma<T>::bits(va[1]) << 16);
                const unsigned a1 = (unsigned)(unsigned short)FirMfma<T>::bits(va[2]) | ((unsigned)(unsigned short)FirMfma<T>::bits(va[3]) << 16);
                const unsigned b0 = (unsigned)(unsigned short)FirMfma<T>::bits(vb[0]) | ((unsigned)(unsigned short)FirMfma<T>::bits(vb[1]) << 16);
                const unsigned b1 = (unsigned)(unsigned short)FirMfma<T>::bits(vb[2]) | ((unsigned)(unsigned short)FirMfma<T>::bits(vb[3]) << 16);
                const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);      // all lanes take part (no divergence above)
                const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned uint4_t;
                const uint4_t o = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) *reinterpret_cast<uint4_t*>(dst + c * 16) = o;
            }
        }
    }
}

// Sliding-window variant of the matrix-core FIR: a workgroup owns a strip of 32 output columns x 64 channels and WALKS DOWN it, four
// output rows per step.  The window rows live in a 12-row ring in LDS (row r of the strip's window in slot r % 12); a step needs the seven
// rows [4s, 4s + 7), of which only the last four are new, and those are fetched by LDS-DMA one step ahead while the matrix cores work on the
// current rows.  The tile kernel above re-reads three of every eleven window rows (and, before the column mask, 40 of 35 columns): 1.72x the
// input bytes; here the vertical overlap is read once per strip segment and the horizontal one is 35 / 32, and the loads of step s + 1, the
// MFMAs of step s and the stores of step s - 1 are in flight together (two workgroups per CU: 72 KB of LDS each).
// Order inside a step: issue the prefetch -> compute from LDS -> wait for the prefetch (the stores of the previous step, issued a whole step
// ago, have drained by then) -> barrier (every wave is done with the rows the NEXT prefetch overwrites, and sees the new rows) -> store.
// BT (backward tail): see UpfirdnArgs::dact_y -- the transposed low-pass in the backward of `bias_act -> low-pass` (D's conv0 -> the filter in front of the
// strided conv1) hands its result straight to the activation gradient, so the gradient w.r.t. the activation's output is never written or re-read.
template <class T, bool BT>
__global__ __launch_bounds__(256, 2) void upfirdn2d_fir_slide_kernel(UpfirdnArgs p, unsigned x_bytes, int tiles_x, int cblocks, int ysegs, int seg_rows)
{
    constexpr int TX = 32, FH = 4, FW = 4, SR = 4;      // SR = output rows per step (one per wave)
    constexpr int RING = 12, WXL = 48, PPR = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int cb0 = b % cblocks; b /= cblocks;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ys = b % ysegs; const int n = b / ysegs;
    const int ox0 = tx * TX;
    const int oy_begin = ys * seg_rows;
    const int oy_end = (oy_begin + seg_rows < p.outH) ? oy_begin + seg_rows : p.outH;
    const int nsteps = (oy_end - oy_begin + SR - 1) / SR;
    if (nsteps <= 0) return;

    for (int i = tid; i < RING * 64; i += 256) {        // pixels 40..47 of every ring row only ever meet zero taps: keep them finite
        const int r = i >> 6, c = i & 63;
        *reinterpret_cast<float4_t*>(fsm + ((r * WXL + 40) * 128) + c * 16) = float4_t{0.f, 0.f, 0.f, 0.f};
    }
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    const int drow = lane >> 3, dchunk = lane & 7;
    // window row w (0 = the first row behind output row oy_begin) -> ring slot w % RING; rows [w0, w1) are dealt to the waves piece by piece
    auto fetch_rows = [&](int w0, int w1) {
        for (int piece = wave + (w0 * PPR); piece < w1 * PPR; piece += 4) {
            const int w = piece / PPR, j = piece - w * PPR;
            const int slot = w % RING;
            const int R = slot * WXL + j * 8 + drow;
            const int sc = (((dchunk >> 1) ^ ((R >> 1) & 3)) << 1) | (dchunk & 1);
            const int iy = oy_begin - p.pady0 + w, ix = ox0 - p.padx0 + j * 8 + drow;
            const unsigned okm = 0u - (unsigned)(((unsigned)iy < (unsigned)p.inH) & ((unsigned)ix < (unsigned)p.inW) & (j * 8 + drow < TX + FW - 1));
            const unsigned real = (unsigned)(n * (int)p.isn + iy * (int)p.isy + ix * (int)p.isx + cb0 * 64 + sc * 8) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (fir_lds_ptr)(fsm + (slot * WXL + j * 8) * 128), 16, (real & okm) | (SBG_FIR_OOB & ~okm), 0, 0, 0);
        }
    };
    fetch_rows(0, SR + FH - 1);

    const int fi = lane & 15, fg = lane >> 4, fq = fi >> 2, fp = fi & 3;
    const unsigned lds_base = (unsigned)(uintptr_t)((fir_lds_ptr)fsm);
    short8_t bt[FH];
    {   // the sixteen taps first (uniform addresses, all in flight together), then the per-lane band matrix by selection: written as a load
        // under the lane's own condition this was 32 dependent load -> wait round trips at the head of every workgroup
        float taps[FH][FW];
#pragma unroll
        for (int ky = 0; ky < FH; ky++)
#pragma unroll
            for (int kx = 0; kx < FW; kx++) taps[ky][kx] = p.f[(p.flip ? ky : FH - 1 - ky) * p.fsy + (p.flip ? kx : FW - 1 - kx) * p.fsx];
#pragma unroll
        for (int ky = 0; ky < FH; ky++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int kpix = j < 4 ? 4 * fg + j : 16 + 4 * fg + (j - 4);
                const int kx = kpix - fi;
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < FW; q++) v = (kx == q) ? taps[ky][q] : v;
                bt[ky][j] = FirMfma<T>::bits(v);
            }
    }
    // the swizzle of an LDS row depends on (R >> 1) & 3 with R = slot * 48 + pixel: 48 is a multiple of 8, so the offsets are slot-independent
    int offA[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int Rrel = 4 * fg + fq, chunk = ((c * 16) >> 3) + (fp >> 1);
        const int sw = (((chunk >> 1) ^ ((Rrel >> 1) & 3)) << 1) | (chunk & 1);
        offA[c] = Rrel * 128 + sw * 16 + (fp & 1) * 8;
    }
    T* yb = (T*)p.y + n * p.osn + cb0 * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
    float4_t t_scale[4], t_bias[4];
    const float t_alpha = p.tail ? p.alpha : 1.f, t_gain = p.tail ? p.act_gain : 1.f, t_cl = (p.tail && p.clamp >= 0.f) ? p.clamp : __builtin_inff();
#pragma unroll
    for (int c = 0; c < 4; c++) { t_scale[c] = float4_t{p.gain, p.gain, p.gain, p.gain}; t_bias[c] = float4_t{0.f, 0.f, 0.f, 0.f}; }
    if (p.tail && p.oscale) {           // (one branch per tensor: its four loads are in flight together)
#pragma unroll
        for (int c = 0; c < 4; c++) t_scale[c] *= *reinterpret_cast<const float4_t*>(p.oscale + (int64_t)n * p.C + cb0 * 64 + c * 16 + 4 * fg);
    }
    if (p.tail && p.bias) {
#pragma unroll
        for (int c = 0; c < 4; c++) t_bias[c] = *reinterpret_cast<const float4_t*>(p.bias + cb0 * 64 + c * 16 + 4 * fg);
    }
    float4_t t_post[4];
#pragma unroll
    for (int c = 0; c < 4; c++) t_post[c] = float4_t{1.f, 1.f, 1.f, 1.f};
    if (p.tail == 1 && p.post) {
#pragma unroll
        for (int c = 0; c < 4; c++) t_post[c] = *reinterpret_cast<const float4_t*>(p.post + (int64_t)n * p.C + cb0 * 64 + c * 16 + 4 * fg);
    }
    typedef __attribute__((ext_vector_type(4))) unsigned uint4_t;
    const T* ysv = BT ? (const T*)p.dact_y + n * p.osn + cb0 * 64 + (fg & 1) * 16 + (fg >> 1) * 8 : nullptr;      // the store layout of this lane (see below)
    const float d_gpos = p.dact_gpos, d_gneg = p.dact_gneg, d_rail = p.dact_rail;
    float4_t dbacc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) dbacc[c] = float4_t{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < nsteps; s++) {
        uint4_t ysave[2][2];
        if (BT) {       // the saved activations of this step's outputs, in the (coalesced, 16 B per lane) layout the stores use; issued ahead of the prefetch
            const int oy_ = oy_begin + SR * s + wave;
#pragma unroll
            for (int sg = 0; sg < 2; sg++) {
                const int ox_ = ox0 + sg * 16 + fi;
                const bool ok_ = oy_ < oy_end && ox_ < p.outW;
                const T* src = ysv + (ok_ ? oy_ * p.osy + ox_ * p.osx : 0);
#pragma unroll
                for (int h = 0; h < 2; h++) ysave[sg][h] = *reinterpret_cast<const uint4_t*>(src + h * 32);
            }
        }
        if (s + 1 < nsteps) fetch_rows(SR * (s + 1) + FH - 1, SR * (s + 2) + FH - 1);       // the four new rows of the next window
        float4_t acc[2][4];
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[sg][c] = float4_t{0.f, 0.f, 0.f, 0.f};
        // Reads by inline assembly (lds_asm.h): behind the LDS-DMA prefetch just issued, builtin reads get a compiler-inserted vmcnt(0) and the
        // prefetch would be waited for before the first MFMA of the step.  Eight groups (filter row ky, pixel half sg) of eight reads, the next
        // group in flight while the current one multiplies.
        unsigned rowa[FH];
#pragma unroll
        for (int ky = 0; ky < FH; ky++) rowa[ky] = lds_base + (unsigned)((((SR * s + wave + ky) % RING) * WXL) * 128);
        short4_t lo[2][4], hi[2][4];
        auto issue_grp = [&](auto gt) {
            constexpr int g = decltype(gt)::value, ky = g >> 1, sg = g & 1;
            sbg_static_for<4>([&](auto ct) {
                constexpr int c = decltype(ct)::value;
                lds_tr_issue<sg * 16 * 128>(lo[g & 1][c], rowa[ky] + (unsigned)offA[c]);
                lds_tr_issue<sg * 16 * 128 + 16 * 128>(hi[g & 1][c], rowa[ky] + (unsigned)offA[c]);
            });
        };
        issue_grp(std::integral_constant<int, 0>{});
        sbg_static_for<8>([&](auto gt) {
            constexpr int g = decltype(gt)::value, ky = g >> 1, sg = g & 1;
            if constexpr (g + 1 < 8) issue_grp(std::integral_constant<int, g + 1>{});
            lds_wait<(g + 1 < 8 ? 8 : 0)>(lo[g & 1], hi[g & 1]);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const short8_t fa = {lo[g & 1][c][0], lo[g & 1][c][1], lo[g & 1][c][2], lo[g & 1][c][3], hi[g & 1][c][0], hi[g & 1][c][1], hi[g & 1][c][2], hi[g & 1][c][3]};
                acc[sg][c] = FirMfma<T>::run(fa, bt[ky], acc[sg][c]);
            }
        });
        const int oy = oy_begin + SR * s + wave;
        // fused tail in fp32, then pack to 16-bit pairs
        unsigned pk[2][4][2];
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            const int ox = ox0 + sg * 16 + fi;
            const bool okp = oy < oy_end && ox < p.outW;
            if constexpr (BT) {
                // undo the store permutation on the saved activations: swapping is its own inverse, so the pair (c, c + 1) comes back from the same instruction
                unsigned yk[4][2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const auto q0 = __builtin_amdgcn_permlane16_swap(ysave[sg][h][0], ysave[sg][h][2], false, false);
                    const auto q1 = __builtin_amdgcn_permlane16_swap(ysave[sg][h][1], ysave[sg][h][3], false, false);
                    yk[2 * h][0] = q0[0]; yk[2 * h + 1][0] = q0[1]; yk[2 * h][1] = q1[0]; yk[2 * h + 1][1] = q1[1];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float4_t v = acc[sg][c] * t_scale[c];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float yv = FirMfma<T>::from_bits((unsigned short)(yk[c][e >> 1] >> ((e & 1) * 16)));
                        const float slope = (fabsf(yv) < d_rail) ? (yv > 0.f ? d_gpos : d_gneg) : 0.f;
                        v[e] *= slope;
                        dbacc[c][e] += okp ? v[e] : 0.f;
                    }
                    pk[sg][c][0] = (unsigned)(unsigned short)FirMfma<T>::bits(v[0]) | ((unsigned)(unsigned short)FirMfma<T>::bits(v[1]) << 16);
                    pk[sg][c][1] = (unsigned)(unsigned short)FirMfma<T>::bits(v[2]) | ((unsigned)(unsigned short)FirMfma<T>::bits(v[3]) << 16);
                }
            } else {
            const float nz = (p.tail && p.noise && okp) ? p.noise[n * p.noise_sn + (int64_t)oy * p.outW + ox] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float4_t v = acc[sg][c] * t_scale[c] + (t_bias[c] + nz);
#pragma unroll
                for (int e = 0; e < 4; e++) { float u = v[e]; u = (u > 0.f) ? u : u * t_alpha; v[e] = __builtin_amdgcn_fmed3f(u * t_gain, -t_cl, t_cl) * t_post[c][e]; }
                pk[sg][c][0] = (unsigned)(unsigned short)FirMfma<T>::bits(v[0]) | ((unsigned)(unsigned short)FirMfma<T>::bits(v[1]) << 16);
                pk[sg][c][1] = (unsigned)(unsigned short)FirMfma<T>::bits(v[2]) | ((unsigned)(unsigned short)FirMfma<T>::bits(v[3]) << 16);
            }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the prefetch has landed (and the previous step's stores drained)
        __syncthreads();
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            const int ox = ox0 + sg * 16 + fi;
            const bool ok = oy < oy_end && ox < p.outW;
            T* dst = yb + oy * p.osy + ox * p.osx;
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[sg][c][0], pk[sg][c + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[sg][c][1], pk[sg][c + 1][1], false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned uint4_t;
                const uint4_t o = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) *reinterpret_cast<uint4_t*>(dst + c * 16) = o;
            }
        }
    }
    if constexpr (BT) {
        // per-channel sums of this workgroup (fixed order): over the 16 pixel lanes of a row by DPP rotates, then over the four waves through LDS
        __syncthreads();                                 // the ring is no longer read
        float* red = reinterpret_cast<float*>(fsm);      // [4 waves][64 channels]
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a = dbacc[c][e];
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xf, 0xf, false));    // row_ror:8
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x124, 0xf, 0xf, false));    // row_ror:4
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x122, 0xf, 0xf, false));    // row_ror:2
                a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x121, 0xf, 0xf, false));    // row_ror:1
                if (fi == 0) red[wave * 64 + c * 16 + 4 * fg + e] = a;
            }
        __syncthreads();
        if (tid < 64) p.dact_part[(int64_t)blockIdx.x * 64 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
}

// strip / segment geometry of the sliding-window kernel (shared by the launcher and by sbg_upfirdn2d_dact_rows)
static bool fir_slide_geometry(const UpfirdnArgs& a, int& tiles_x, int& cblocks, int& ysegs, int& seg_rows, int64_t& nblk)
{
    tiles_x = (a.outW + 31) / 32; cblocks = a.C / 64;
    const int64_t strips = (int64_t)a.N * tiles_x * cblocks;
    // vertical segments: enough workgroups for two per CU on every CU, but at least 16 output rows per segment (3 overlap rows are re-read per segment)
    ysegs = 1;
    while (strips * ysegs < 1024 && a.outH / (ysegs * 2) >= 16) ysegs *= 2;
    seg_rows = (((a.outH + ysegs - 1) / ysegs) + 3) & ~3;
    ysegs = (a.outH + seg_rows - 1) / seg_rows;
    nblk = strips * ysegs;
    return nblk <= INT32_MAX && nblk > 0;
}

template <class T>
static bool launch_fir_slide(const UpfirdnArgs& a, hipStream_t stream)
{
    constexpr int lds = 12 * 48 * 128;
    const int64_t x_bytes = 2 * ((int64_t)(a.N - 1) * a.isn + (int64_t)(a.inH - 1) * a.isy + (int64_t)(a.inW - 1) * a.isx + a.C);
    if (x_bytes >= (int64_t)SBG_FIR_OOB || a.isn < 0 || a.isy < 0 || a.isx < 0) return false;
    int tiles_x, cblocks, ysegs, seg_rows; int64_t nblk;
    if (!fir_slide_geometry(a, tiles_x, cblocks, ysegs, seg_rows, nblk)) return false;
    if (a.tail == 2) {
        auto kern = upfirdn2d_fir_slide_kernel<T, true>;
        if (!SBG_RAISE_LDS_ONCE(kern, lds)) return false;
        SBG_LAUNCH_OR(return false, kern, dim3((unsigned)nblk), dim3(256), lds, stream, a, (unsigned)x_bytes, tiles_x, cblocks, ysegs, seg_rows);
    } else {
        auto kern = upfirdn2d_fir_slide_kernel<T, false>;
        if (!SBG_RAISE_LDS_ONCE(kern, lds)) return false;
        SBG_LAUNCH_OR(return false, kern, dim3((unsigned)nblk), dim3(256), lds, stream, a, (unsigned)x_bytes, tiles_x, cblocks, ysegs, seg_rows);
    }
    return true;
}

template <class T, int RPW>
static bool launch_fir_mfma_rpw(const UpfirdnArgs& a, hipStream_t stream)
{
    constexpr int TY = 4 * RPW, lds = (TY + 3) * 48 * 128;
    const int64_t x_bytes = 2 * ((int64_t)(a.N - 1) * a.isn + (int64_t)(a.inH - 1) * a.isy + (int64_t)(a.inW - 1) * a.isx + a.C);
    if (x_bytes >= (int64_t)SBG_FIR_OOB || a.isn < 0 || a.isy < 0 || a.isx < 0) return false;
    const int tiles_x = (a.outW + 31) / 32, tiles_y = (a.outH + TY - 1) / TY, cblocks = a.C / 64;
    const int64_t nblk = (int64_t)a.N * tiles_y * tiles_x * cblocks;
    if (nblk > INT32_MAX || nblk <= 0) return false;
    auto kern = upfirdn2d_fir_mfma_kernel<T, RPW>;
    if (!SBG_RAISE_LDS_ONCE(kern, lds)) return false;
    SBG_LAUNCH_OR(return false, kern, dim3((unsigned)nblk), dim3(256), lds, stream, a, (unsigned)x_bytes, tiles_x, tiles_y, cblocks);
    return true;
}

template <class T>
static bool launch_fir_mfma(const UpfirdnArgs& a, hipStream_t stream)
{
    static const char* e = sbg_env("SBG_FIR_RPW");        // experiment switch: tile kernels (1 = 4 x 32 tiles, 2 = 8 x 32, 4 = 16 x 32) instead of the sliding window
    if (a.tail == 2) return a.outH >= 16 && launch_fir_slide<T>(a, stream);      // the backward tail exists in the sliding-window kernel only
    if (!e && a.outH >= 16) return launch_fir_slide<T>(a, stream);
    if (e && atoi(e) == 1) return launch_fir_mfma_rpw<T, 1>(a, stream);
    if (e && atoi(e) == 4) return launch_fir_mfma_rpw<T, 4>(a, stream);
    return launch_fir_mfma_rpw<T, 2>(a, stream);
}

template <class T> static bool try_fir_mfma(const UpfirdnArgs& a, hipStream_t stream) { return launch_fir_mfma<T>(a, stream); }
template <> bool try_fir_mfma<float>(const UpfirdnArgs&, hipStream_t) { return false; }

template <class T>
static int launch_upfirdn(const UpfirdnArgs& a0, bool vec8, bool exact16, hipStream_t stream)
{
    UpfirdnArgs a = a0;
    const double es = sizeof(T) == 4 ? 4 : 2;
    SbgProfScope prof(stream, SBG_K_UPFIRDN2D, 0.0,
                      es * ((double)a.N * a.C * a.inH * a.inW + (double)a.N * a.C * a.outH * a.outW),
                      {a.N, a.C, a.inH, a.inW, a.outH, a.outW, a.upx * 16 + a.downx});
    const bool mfma_ok = vec8 && exact16 && sizeof(T) == 2 && a.upx == 1 && a.upy == 1 && a.downx == 1 && a.downy == 1 && a.fw == 4 && a.fh == 4 && (a.C % 64) == 0
                         && a.outW >= 16 && a.outH >= 8 && sbg_env("SBG_FIR_NO_MFMA") == nullptr;
    // A few columns beyond a multiple of the 32-column strips (the 2 res + 1 wide outputs of the discriminator's low-pass in front of a strided
    // convolution: 257 = 8 strips + 1) would cost a whole strip of matrix-core work (9 strips for 257 columns: 11 % idle, 33 columns: 48 %): the
    // strips take the multiple of 32 and the register-blocked kernel the remaining columns, as a launch over the shifted sub-rectangle.
    const int rem = a.outW % 32;
    static const char* no_edge = sbg_env("SBG_FIR_NO_EDGE");
    if (mfma_ok && !a.tail && !no_edge && rem >= 1 && rem <= 4 && a.outW > 32) {
        UpfirdnArgs m = a, e = a;
        m.outW = a.outW - rem;
        e.outW = rem; e.padx0 = a.padx0 - m.outW; e.y = (void*)((T*)a.y + (int64_t)m.outW * a.osx);
        if (try_fir_mfma<T>(m, stream)) {
            const int xblocks = (e.outW + FIR_TX - 1) / FIR_TX, yblocks = (e.outH + FIR_TY - 1) / FIR_TY;
            e.total = (int64_t)e.N * yblocks * xblocks * (e.C >> 3);
            SBG_LAUNCH((upfirdn2d_fir_fixed_kernel<T, 4, 4>), dim3(sbg_stream_grid(e.total, 256)), dim3(256), 0, stream, e, xblocks, yblocks);
            SBG_HIP_LAUNCH_CHECK();
            return SBG_OK;
        }
    }
    if (mfma_ok && try_fir_mfma<T>(a, stream)) {
        // matrix-core FIR
    } else if (a.tail) {
        return sbg_fail(SBG_ERR_UNSUPPORTED, "upfirdn2d: fused tail requested but the matrix-core FIR path does not take this launch");
    } else if (vec8 && a.upx == 1 && a.upy == 1 && a.downx == 1 && a.downy == 1 && a.fw <= FIR_MAXF && a.fh <= FIR_MAXF && a.fw * a.fh > 1) {
        const int xblocks = (a.outW + FIR_TX - 1) / FIR_TX, yblocks = (a.outH + FIR_TY - 1) / FIR_TY;
        a.total = (int64_t)a.N * yblocks * xblocks * (a.C >> 3);
        if (a.fw == 4 && a.fh == 4)
            SBG_LAUNCH((upfirdn2d_fir_fixed_kernel<T, 4, 4>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a, xblocks, yblocks);
        else
            SBG_LAUNCH((upfirdn2d_fir_kernel<T>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a, xblocks, yblocks);
    } else if (vec8) {
        a.total = (int64_t)a.N * a.outH * a.outW * (a.C >> 3);
        SBG_LAUNCH((upfirdn2d_kernel<T, 8>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    } else {
        a.total = (int64_t)a.N * a.outH * a.outW * a.C;
        SBG_LAUNCH((upfirdn2d_kernel<T, 1>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    }
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

extern "C" int sbg_upfirdn2d_tail_supported(const sbg_upfirdn2d_params* q)
{
    if (!q) return 0;
    const int C = q->inSize[2];
    return (q->dtype == SBG_BF16 || q->dtype == SBG_F16) && q->filter_exact16 && q->upx == 1 && q->upy == 1 && q->downx == 1 && q->downy == 1
           && q->filterSize[0] == 4 && q->filterSize[1] == 4 && (C % 64) == 0 && q->outSize[0] >= 16 && q->outSize[1] >= 8
           && q->inStride[2] == 1 && q->outStride[2] == 1 && sbg_env("SBG_FIR_NO_MFMA") == nullptr;
}

// Rows of `dact_partial` (= workgroups of the sliding-window launch) for a launch with a backward tail, or -1 when that kernel does not take it.
extern "C" int64_t sbg_upfirdn2d_dact_rows(const sbg_upfirdn2d_params* q)
{
    if (!q || !sbg_upfirdn2d_tail_supported(q) || q->outSize[1] < 16 || sbg_env("SBG_FIR_RPW") != nullptr) return -1;
    if (q->inStride[0] < 0 || q->inStride[1] < 0 || q->inStride[3] < 0) return -1;
    const int64_t x_bytes = 2 * ((int64_t)(q->inSize[3] - 1) * q->inStride[3] + (int64_t)(q->inSize[1] - 1) * q->inStride[1] + (int64_t)(q->inSize[0] - 1) * q->inStride[0] + q->inSize[2]);
    if (x_bytes >= (int64_t)SBG_FIR_OOB) return -1;
    UpfirdnArgs a = {};
    a.outW = q->outSize[0]; a.outH = q->outSize[1]; a.C = q->inSize[2]; a.N = q->inSize[3];
    int tiles_x, cblocks, ysegs, seg_rows; int64_t nblk;
    return fir_slide_geometry(a, tiles_x, cblocks, ysegs, seg_rows, nblk) ? nblk : -1;
}

extern "C" int sbg_upfirdn2d(const sbg_upfirdn2d_params* q, sbg_stream_t stream)
{
    SBG_CHECK(q != nullptr && q->x != nullptr && q->f != nullptr && q->y != nullptr, "upfirdn2d: null pointer");
    SBG_CHECK(q->dtype == SBG_F32 || q->dtype == SBG_F16 || q->dtype == SBG_BF16, "upfirdn2d: unsupported dtype %d", q->dtype);
    SBG_CHECK(q->filterSize[0] >= 1 && q->filterSize[1] >= 1, "upfirdn2d: f must be at least 1x1");
    SBG_CHECK(q->upx >= 1 && q->upy >= 1, "upfirdn2d: upsampling factor must be at least 1");
    SBG_CHECK(q->downx >= 1 && q->downy >= 1, "upfirdn2d: downsampling factor must be at least 1");
    SBG_CHECK(q->outSize[0] >= 1 && q->outSize[1] >= 1, "upfirdn2d: output must be at least 1x1");
    SBG_CHECK(q->outSize[2] == q->inSize[2] && q->outSize[3] == q->inSize[3], "upfirdn2d: channel/batch size mismatch");
    const int64_t in_numel  = (int64_t)q->inSize[0] * q->inSize[1] * q->inSize[2] * q->inSize[3];
    const int64_t out_numel = (int64_t)q->outSize[0] * q->outSize[1] * q->outSize[2] * q->outSize[3];
    SBG_CHECK(in_numel <= INT32_MAX, "upfirdn2d: x is too large");
    SBG_CHECK(out_numel <= INT32_MAX, "upfirdn2d: output is too large");
    if (out_numel == 0) return SBG_OK;

    UpfirdnArgs a;
    a.x = q->x; a.f = q->f; a.y = q->y;
    a.upx = q->upx; a.upy = q->upy; a.downx = q->downx; a.downy = q->downy; a.padx0 = q->padx0; a.pady0 = q->pady0;
    a.flip = q->flip ? 1 : 0; a.gain = q->gain;
    a.inW = q->inSize[0]; a.inH = q->inSize[1]; a.C = q->inSize[2]; a.N = q->inSize[3];
    a.isx = q->inStride[0]; a.isy = q->inStride[1]; a.isc = q->inStride[2]; a.isn = q->inStride[3];
    a.fw = q->filterSize[0]; a.fh = q->filterSize[1]; a.fsx = q->filterStride[0]; a.fsy = q->filterStride[1];
    a.outW = q->outSize[0]; a.outH = q->outSize[1];
    a.osx = q->outStride[0]; a.osy = q->outStride[1]; a.osc = q->outStride[2]; a.osn = q->outStride[3];
    a.total = 0;
    a.oscale = q->oscale; a.noise = q->noise; a.noise_sn = q->noise_stride_n; a.bias = q->bias;
    a.tail = (q->act != 0) ? 1 : 0;
    a.alpha = q->act == SBG_ACT_LRELU ? q->alpha : (q->act == SBG_ACT_RELU ? 0.f : 1.f); a.act_gain = q->act_gain; a.clamp = q->clamp;
    a.dact_y = nullptr; a.dact_part = nullptr; a.dact_gpos = a.dact_gneg = 1.f; a.dact_rail = __builtin_inff();
    a.post = q->post_scale;
    SBG_CHECK(!q->post_scale || (q->act != 0 && sbg_aligned16(q->post_scale)), "upfirdn2d: post_scale belongs to the forward tail (act != 0) and must be 16-byte aligned");
    if (q->dact_y) {        // backward tail: slope of clamp(act(.) * gain) at the saved output (same tests as sbg_modconv_bwd)
        SBG_CHECK(q->act == 0, "upfirdn2d: forward and backward tails exclude each other");
        SBG_CHECK(q->dact_partial != nullptr, "upfirdn2d: the backward tail needs dact_partial (sbg_upfirdn2d_dact_rows() x 64 floats)");
        SBG_CHECK(q->dact_act == SBG_ACT_LINEAR || q->dact_act == SBG_ACT_RELU || q->dact_act == SBG_ACT_LRELU, "upfirdn2d: backward-tail activation must be linear, relu or lrelu");
        SBG_CHECK(q->dact_gain > 0.f && sbg_aligned16(q->dact_y), "upfirdn2d: backward tail: gain must be positive, dact_y 16-byte aligned");
        a.tail = 2; a.dact_y = q->dact_y; a.dact_part = q->dact_partial;
        a.dact_gpos = q->dact_gain;
        a.dact_gneg = q->dact_act == SBG_ACT_LRELU ? q->dact_gain * q->dact_alpha : (q->dact_act == SBG_ACT_RELU ? 0.f : q->dact_gain);
        a.dact_rail = q->dact_clamp >= 0.f ? q->dact_clamp : __builtin_inff();
    }

    // 8-channel vector path: channel-minor on both sides, every pixel start 16-B aligned.
    const int es = sbg_dtype_size(q->dtype);
    auto mult8 = [&](int64_t s) { return ((s * es) % 16) == 0; };
    bool vec8 = a.isc == 1 && a.osc == 1 && (a.C % 8) == 0 && sbg_aligned16(a.x) && sbg_aligned16(a.y) &&
                mult8(a.isx) && mult8(a.isy) && mult8(a.isn) && mult8(a.osx) && mult8(a.osy) && mult8(a.osn) &&
                (es == 2 || ((a.isx | a.isy | a.isn | a.osx | a.osy | a.osn) % 4) == 0);
    hipStream_t s = (hipStream_t)stream;
    const bool exact16 = q->filter_exact16 != 0;
    if (a.tail == 2) {
        SBG_CHECK(vec8 && sbg_upfirdn2d_dact_rows(q) > 0, "upfirdn2d: the backward tail needs the sliding-window matrix-core FIR (sbg_upfirdn2d_dact_rows)");
    } else if (a.tail) {
        SBG_CHECK(q->act == SBG_ACT_LINEAR || q->act == SBG_ACT_RELU || q->act == SBG_ACT_LRELU, "upfirdn2d: fused activation must be linear, relu or lrelu");
        SBG_CHECK(vec8 && sbg_upfirdn2d_tail_supported(q), "upfirdn2d: the fused tail needs the matrix-core FIR path (sbg_upfirdn2d_tail_supported)");
        SBG_CHECK((!q->oscale || sbg_aligned16(q->oscale)) && (!q->bias || sbg_aligned16(q->bias)), "upfirdn2d: oscale / bias must be 16-byte aligned");
    }
    if (q->dtype == SBG_F32) return launch_upfirdn<float>(a, vec8, false, s);
    if (q->dtype == SBG_F16) return launch_upfirdn<f16_s>(a, vec8, exact16, s);
    return launch_upfirdn<bf16_s>(a, vec8, exact16, s);
}
