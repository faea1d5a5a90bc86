// grouped_gemm.hip -- many small fp32 products in ONE launch.
//
// A synthesis network evaluates ~20 per-layer affine maps `styles_l = w[:, slot_l] @ W_l^T * gain_l + b_l` (reference
// train_parts/generators.py:333 `self.affine(w)` in every SynthesisLayer / ToRGBLayer, FullyConnectedLayer :97-131) -- forward ~20 GEMM
// launches of [N, 512] x [512, C_l], backward ~40 more plus ~20 bias sums, each 5-40 us of a device that is otherwise busy with
// millisecond convolutions: ~1.5 ms and ~140 launches per training step (profiles/r03b_kernel_stats.csv, Cijk_* / reduce_kernel rows).
// Here a pass's products are described by a table (sbg_gg_problem, include/sbg_hip.h) and run as one grid: problem p owns tiles
// [tile0_p, tile0_{p+1}) of 64 x 64 outputs, C = sum_t alpha_t A_t B_t (+ bias), up to two terms (the gradient w.r.t. a `w` slot that
// feeds two layers), optional row sums of A (the bias gradient rides with the weight gradient).  fp32 FMA, ascending k, fixed order:
// results do not depend on the grid.  Any strides (transposes are strides); not a matrix-core kernel on purpose -- 0.8 GFLOP per pass.
#include "sbg_common.h"

namespace {

constexpr int GG_MAX = 16;          // problems per launch (the table travels by value in the kernel arguments)
constexpr int BM = 64, BN = 64, BK = 16;

struct GgTable { sbg_gg_problem p[GG_MAX]; int tile0[GG_MAX + 1]; int count; };

__global__ __launch_bounds__(256) void grouped_gemm_kernel(GgTable tb)
{
    __shared__ float As[BK][BM + 4], Bs[BK][BN + 4];
    int pi = 0;
    while (pi + 1 < tb.count && (int)blockIdx.x >= tb.tile0[pi + 1]) pi++;
    const sbg_gg_problem& q = tb.p[pi];
    const int tiles_n = (q.N + BN - 1) / BN;
    const int t = blockIdx.x - tb.tile0[pi];
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    float rsum = 0.f;
    const bool want_rsum = q.rowsum != nullptr && n0 == 0 && tid < BM;
    for (int term = 0; term < q.nterms; term++) {
        const float* __restrict__ A = term == 0 ? q.a0 : q.a1;
        const float* __restrict__ B = term == 0 ? q.b0 : q.b1;
        const int64_t a_rs = term == 0 ? q.a0_rs : q.a1_rs, a_cs = term == 0 ? q.a0_cs : q.a1_cs;
        const int64_t b_rs = term == 0 ? q.b0_rs : q.b1_rs, b_cs = term == 0 ? q.b0_cs : q.b1_cs;
        const int K = term == 0 ? q.K0 : q.K1;
        const float alpha = term == 0 ? q.alpha0 : q.alpha1;
        float part[4][4] = {};
        // the next K-tile's 64 x 16 elements of each operand are fetched into registers while the current one multiplies (a tile is one round trip
        // to L2: without the prefetch every one of the K / 16 iterations waited for it -- 87 us for a pass's 0.8 GFLOP)
        float ra[4], rb[4];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int e = 0; e < 4; e++) {      // consecutive threads along the operand's unit-stride axis
                const int idx = tid + 256 * e;
                int m, k;
                if (a_cs == 1) { k = idx & (BK - 1); m = idx >> 4; } else { m = idx & (BM - 1); k = idx >> 6; }
                ra[e] = (m0 + m < q.M && k0 + k < K) ? A[(int64_t)(m0 + m) * a_rs + (int64_t)(k0 + k) * a_cs] : 0.f;
                int n, kb;
                if (b_rs == 1) { kb = idx & (BK - 1); n = idx >> 4; } else { n = idx & (BN - 1); kb = idx >> 6; }
                rb[e] = (n0 + n < q.N && k0 + kb < K) ? B[(int64_t)(k0 + kb) * b_rs + (int64_t)(n0 + n) * b_cs] : 0.f;
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int idx = tid + 256 * e;
                int m, k;
                if (a_cs == 1) { k = idx & (BK - 1); m = idx >> 4; } else { m = idx & (BM - 1); k = idx >> 6; }
                As[k][m] = ra[e];
                int n, kb;
                if (b_rs == 1) { kb = idx & (BK - 1); n = idx >> 4; } else { n = idx & (BN - 1); kb = idx >> 6; }
                Bs[kb][n] = rb[e];
            }
        };
        if (K > 0) fetch(0);
        for (int k0 = 0; k0 < K; k0 += BK) {
            stash();
            __syncthreads();
            if (k0 + BK < K) fetch(k0 + BK);
            if (want_rsum && term == 0) {
#pragma unroll
                for (int k = 0; k < BK; k++) rsum += As[k][tid];
            }
#pragma unroll
            for (int k = 0; k < BK; k++) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) part[i][j] = fmaf(a[i], b[j], part[i][j]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = fmaf(alpha, part[i][j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + ty * 4 + i;
        if (m >= q.M) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + tx * 4 + j;
            if (n >= q.N) continue;
            float v = acc[i][j];
            if (q.bias) v = fmaf(q.bias[n], q.bias_scale, v);
            q.c[(int64_t)m * q.c_rs + (int64_t)n * q.c_cs] = v;
        }
    }
    if (want_rsum && m0 + tid < q.M) q.rowsum[m0 + tid] = rsum * q.rowsum_scale;
}

} // namespace

extern "C" int sbg_grouped_gemm(const sbg_gg_problem* problems, int count, sbg_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (count < 0 || (count > 0 && !problems)) return sbg_fail(SBG_ERR_INVALID, "sbg_grouped_gemm: null table");
    for (int i = 0; i < count; i++) {
        const sbg_gg_problem& q = problems[i];
        if (q.M < 0 || q.N < 0 || q.nterms < 1 || q.nterms > 2 || q.K0 < 0 || (q.nterms == 2 && q.K1 < 0) || !q.c || !q.a0 || !q.b0 || (q.nterms == 2 && (!q.a1 || !q.b1)))
            return sbg_fail(SBG_ERR_INVALID, "sbg_grouped_gemm: malformed problem");
    }
    for (int i = 0; i < count;) {
        GgTable tb;
        tb.count = 0;
        int tiles = 0;
        double flops = 0, bytes = 0;
        for (; i < count && tb.count < GG_MAX; i++) {
            const sbg_gg_problem& q = problems[i];
            const int64_t nt = (int64_t)((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN);
            if (nt == 0) continue;                            // (an empty product: nothing to write)
            if (tiles + nt > (1 << 20)) return sbg_fail(SBG_ERR_INVALID, "sbg_grouped_gemm: grid too large");
            tb.p[tb.count] = q; tb.tile0[tb.count] = tiles; tb.count++;
            tiles += (int)nt;
            const double ks = (double)q.K0 + (q.nterms == 2 ? q.K1 : 0);
            flops += 2.0 * q.M * (double)q.N * ks;
            bytes += 4.0 * (q.M * ks + ks * q.N + (double)q.M * q.N);
        }
        if (tb.count == 0) continue;
        tb.tile0[tb.count] = tiles;
        SbgProfScope prof(stream, SBG_K_GROUPED_GEMM, flops, bytes, {tb.count, tiles, 0, 0, 0, 0, 0});
        SBG_LAUNCH(grouped_gemm_kernel, dim3(tiles), dim3(256), 0, stream, tb);
        SBG_HIP_LAUNCH_CHECK();
    }
    return SBG_OK;
}
