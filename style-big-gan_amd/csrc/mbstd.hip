// mbstd.hip -- the discriminator's minibatch-standard-deviation layer (train_parts/discriminators.py:313-328 of the reference) as one kernel per
// direction (two small launches forward, one backward).  The reference composes it from ~10 tensor ops (reshape, mean, subtract, square, mean, sqrt, mean, repeat, cat) whose backward is
// ~14 more; the tensors are tiny ([N, 512, 4, 4]), so the layer is pure launch latency -- once per discriminator pass, ~13 passes per step.
//   x: fp32 [N, C, H, W] dense (NCHW), N = G * M (sample n = g * M + m belongs to group m), C = F * c.
//   y: fp32 [N, C + F, H, W]: y[:, :C] = x;  y[g*M + m, C + f, :, :] = stat[m, f] = mean_{cc,h,w} sqrt(var_g(x[g*M + m, f*c + cc, h, w]) + 1e-8)
// Backward (first order): dx = dy[:, :C] + dstat[m, f] * (x - mean_g) / (G * c*H*W * sqrt(var + 1e-8)),  dstat[m, f] = sum_{g,h,w} dy[g*M+m, C+f, h, w].
#include "sbg_common.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Forward, pass 1: workgroup (m, f, chunk) handles 256 consecutive positions of the group's c * H * W (one per lane): mean / variance over the
// G samples, copy of x into y, partial sum of sqrt(var + 1e-8) -> part[(f * M + m) * nchunk + chunk].  (A first version ran ONE workgroup per
// (m, f) -- with the benchmark's single group that is one CU walking 8192 positions x 32 samples: 6 ms per step slower than the tensor ops.)
__global__ __launch_bounds__(256) void mbstd_fwd_kernel(const float* x, float* y, float* part, int G, int M, int F, int c, int HW, int nchunk)
{
    __shared__ float red[4];
    const int chunk = blockIdx.x % nchunk, mf = blockIdx.x / nchunk;
    const int m = mf % M, f = mf / M;
    const int C = F * c, CO = C + F;
    const int npos = c * HW;
    const int pos = chunk * 256 + threadIdx.x;
    float s = 0.f;
    if (pos < npos) {
        const int64_t xo = (int64_t)(f * c) * HW + pos;
        float mean = 0.f;
        for (int g = 0; g < G; g++) mean += x[(int64_t)(g * M + m) * C * HW + xo];
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; g++) {
            const float v = x[(int64_t)(g * M + m) * C * HW + xo];
            y[(int64_t)(g * M + m) * CO * HW + xo] = v;
            var += (v - mean) * (v - mean);
        }
        s = sqrtf(var / (float)G + 1e-8f);
    }
    const float tot = block_sum256(s, red);
    if (threadIdx.x == 0) part[(int64_t)mf * nchunk + chunk] = tot;
}

// Forward, pass 2: one workgroup per (m, f): fixed-order sum of the partials, broadcast into the extra channel of the group's samples
__global__ __launch_bounds__(256) void mbstd_stat_kernel(const float* part, float* y, int G, int M, int F, int c, int HW, int nchunk)
{
    __shared__ float red[4];
    const int m = blockIdx.x % M, f = blockIdx.x / M;
    const int C = F * c, CO = C + F;
    float s = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += 256) s += part[(int64_t)blockIdx.x * nchunk + k];
    const float stat = block_sum256(s, red) / (float)(c * HW);
    for (int i = threadIdx.x; i < G * HW; i += 256) {
        const int g = i / HW, p = i - g * HW;
        y[(int64_t)(g * M + m) * CO * HW + (int64_t)(C + f) * HW + p] = stat;
    }
}

// Backward: workgroup (m, f, chunk); every workgroup re-derives dstat from the G * H*W values of the extra channel (a few hundred loads)
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const float* x, const float* dy, float* dx, int G, int M, int F, int c, int HW, int nchunk)
{
    __shared__ float red[4];
    const int chunk = blockIdx.x % nchunk, mf = blockIdx.x / nchunk;
    const int m = mf % M, f = mf / M;
    const int C = F * c, CO = C + F;
    const int npos = c * HW;
    float part = 0.f;
    for (int i = threadIdx.x; i < G * HW; i += 256) {
        const int g = i / HW, p = i - g * HW;
        part += dy[(int64_t)(g * M + m) * CO * HW + (int64_t)(C + f) * HW + p];
    }
    const float dstat = block_sum256(part, red) / ((float)npos * (float)G);
    const int pos = chunk * 256 + threadIdx.x;
    if (pos >= npos) return;
    const int64_t xo = (int64_t)(f * c) * HW + pos;
    float mean = 0.f;
    for (int g = 0; g < G; g++) mean += x[(int64_t)(g * M + m) * C * HW + xo];
    mean /= (float)G;
    float var = 0.f;
    for (int g = 0; g < G; g++) { const float v = x[(int64_t)(g * M + m) * C * HW + xo]; var += (v - mean) * (v - mean); }
    const float k = dstat / sqrtf(var / (float)G + 1e-8f);
    for (int g = 0; g < G; g++) {
        const int64_t n = g * M + m;
        dx[n * C * HW + xo] = dy[n * CO * HW + xo] + k * (x[n * C * HW + xo] - mean);
    }
}

} // namespace

extern "C" int64_t sbg_mbstd_workspace(int N, int C, int HW, int G, int F)
{
    if (N < 1 || G < 1 || F < 1 || C % F) return -1;
    const int64_t nchunk = ((int64_t)(C / F) * HW + 255) / 256;
    return (int64_t)(N / G) * F * nchunk * (int64_t)sizeof(float);
}

extern "C" int sbg_mbstd_fwd(const float* x, float* y, void* workspace, int N, int C, int HW, int G, int F, sbg_stream_t stream)
{
    SBG_CHECK(x && y && workspace, "mbstd_fwd: null pointer");
    SBG_CHECK(N >= 1 && G >= 1 && N % G == 0 && F >= 1 && C % F == 0 && HW >= 1, "mbstd_fwd: bad sizes N=%d G=%d C=%d F=%d", N, G, C, F);
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = (int)(((int64_t)(C / F) * HW + 255) / 256);
    SBG_LAUNCH(mbstd_fwd_kernel, dim3((unsigned)((N / G) * F * nchunk)), dim3(256), 0, s, x, y, (float*)workspace, G, N / G, F, C / F, HW, nchunk);
    SBG_LAUNCH(mbstd_stat_kernel, dim3((unsigned)((N / G) * F)), dim3(256), 0, s, (const float*)workspace, y, G, N / G, F, C / F, HW, nchunk);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

extern "C" int sbg_mbstd_bwd(const float* x, const float* dy, float* dx, int N, int C, int HW, int G, int F, sbg_stream_t stream)
{
    SBG_CHECK(x && dy && dx, "mbstd_bwd: null pointer");
    SBG_CHECK(N >= 1 && G >= 1 && N % G == 0 && F >= 1 && C % F == 0 && HW >= 1, "mbstd_bwd: bad sizes N=%d G=%d C=%d F=%d", N, G, C, F);
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = (int)(((int64_t)(C / F) * HW + 255) / 256);
    SBG_LAUNCH(mbstd_bwd_kernel, dim3((unsigned)((N / G) * F * nchunk)), dim3(256), 0, s, x, dy, dx, G, N / G, F, C / F, HW, nchunk);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}
