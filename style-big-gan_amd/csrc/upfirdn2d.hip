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

namespace {

struct UpfirdnArgs {
    const void* x; const float* f; void* y;
    int upx, upy, downx, downy, padx0, pady0, flip; float gain;
    int inW, inH, C, N; int64_t isx, isy, isc, isn;
    int fw, fh, fsx, fsy;
    int outW, outH; int64_t osx, osy, osc, osn;
    int64_t total;    // number of lane work items
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

template <class T>
static int launch_upfirdn(const UpfirdnArgs& a0, bool vec8, hipStream_t stream)
{
    UpfirdnArgs a = a0;
    const double es = sizeof(T) == 4 ? 4 : 2;
    SbgProfScope prof(stream, SBG_K_UPFIRDN2D, 0.0,
                      es * ((double)a.N * a.C * a.inH * a.inW + (double)a.N * a.C * a.outH * a.outW),
                      {a.N, a.C, a.inH, a.inW, a.outH, a.outW, a.upx * 16 + a.downx});
    if (vec8 && a.upx == 1 && a.upy == 1 && a.downx == 1 && a.downy == 1 && a.fw <= FIR_MAXF && a.fh <= FIR_MAXF && a.fw * a.fh > 1) {
        const int xblocks = (a.outW + FIR_TX - 1) / FIR_TX, yblocks = (a.outH + FIR_TY - 1) / FIR_TY;
        a.total = (int64_t)a.N * yblocks * xblocks * (a.C >> 3);
        if (a.fw == 4 && a.fh == 4)
            hipLaunchKernelGGL((upfirdn2d_fir_fixed_kernel<T, 4, 4>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a, xblocks, yblocks);
        else
            hipLaunchKernelGGL((upfirdn2d_fir_kernel<T>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a, xblocks, yblocks);
    } else if (vec8) {
        a.total = (int64_t)a.N * a.outH * a.outW * (a.C >> 3);
        hipLaunchKernelGGL((upfirdn2d_kernel<T, 8>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    } else {
        a.total = (int64_t)a.N * a.outH * a.outW * a.C;
        hipLaunchKernelGGL((upfirdn2d_kernel<T, 1>), dim3(sbg_stream_grid(a.total, 256)), dim3(256), 0, stream, a);
    }
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

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

    // 8-channel vector path: channel-minor on both sides, every pixel start 16-B aligned.
    const int es = sbg_dtype_size(q->dtype);
    auto mult8 = [&](int64_t s) { return ((s * es) % 16) == 0; };
    bool vec8 = a.isc == 1 && a.osc == 1 && (a.C % 8) == 0 && sbg_aligned16(a.x) && sbg_aligned16(a.y) &&
                mult8(a.isx) && mult8(a.isy) && mult8(a.isn) && mult8(a.osx) && mult8(a.osy) && mult8(a.osn) &&
                (es == 2 || ((a.isx | a.isy | a.isn | a.osx | a.osy | a.osn) % 4) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (q->dtype == SBG_F32) return launch_upfirdn<float>(a, vec8, s);
    if (q->dtype == SBG_F16) return launch_upfirdn<f16_s>(a, vec8, s);
    return launch_upfirdn<bf16_s>(a, vec8, s);
}
