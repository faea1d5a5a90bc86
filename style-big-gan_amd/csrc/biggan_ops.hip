// biggan_ops.hip -- BigGAN-specific kernels: spectral-norm power iteration and the non-local self-attention core.
//
//  * sn_*: one power iteration of SN.W_ (biggan/layers.py:28-50,87-99) as two streaming passes over the fp32 weight matrix
//    (column sums, then row sums) with the normalisations and sigma folded into the small tail kernels.  HBM-bound:
//    2 * rows * cols * 4 bytes.
//  * attention_fwd_kernel: out = softmax(theta phi^T) g per sample (biggan/layers.py:162-166) on the fp32-input matrix cores
//    (v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fma chain, so parity with the fp32 reference is to rounding).  One wave
//    owns 16 query rows: S = Q K^T lives in accumulator registers, the row softmax runs on them (16-lane shuffles), P goes
//    through a 16 x M LDS strip to become the A operand of P V.  Keys <= 256, so no online-softmax rescaling is needed.
#include "sbg_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------- SN
#define SN_ROWS_PER_BLOCK 32

// partial[rb][k] = sum_{o in row block rb} u[o] * W[o][k]
__global__ __launch_bounds__(256) void sn_colsum_kernel(const float* W, const float* u, float* partial, int rows, int cols)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * SN_ROWS_PER_BLOCK;
    if (k >= cols) return;
    float acc = 0.f;
    const int r1 = (r0 + SN_ROWS_PER_BLOCK < rows) ? r0 + SN_ROWS_PER_BLOCK : rows;
    for (int o = r0; o < r1; o++) acc += u[o] * W[(int64_t)o * cols + k];
    partial[(int64_t)blockIdx.y * cols + k] = acc;
}

// v_raw[k] = sum_rb partial[rb][k]  (fixed order); block 0 also leaves |v_raw|^2 partial sums for the next kernel
__global__ __launch_bounds__(256) void sn_vfinish_kernel(const float* partial, float* v_raw, float* vnorm2_parts, int cols, int nrb)
{
    __shared__ float red[256];
    const int k = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (k < cols) {
        for (int rb = 0; rb < nrb; rb++) acc += partial[(int64_t)rb * cols + k];
        v_raw[k] = acc;
    }
    red[threadIdx.x] = acc * acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) vnorm2_parts[blockIdx.x] = red[0];
}

// one workgroup per row: t[o] = sum_k v_hat[k] W[o][k], v_hat = v_raw / max(|v_raw|, eps); also writes v (normalised) from row 0
__global__ __launch_bounds__(256) void sn_rowsum_kernel(const float* W, const float* v_raw, const float* vnorm2_parts, int nparts,
                                                         float* v_out, float* t, int rows, int cols, float eps)
{
    __shared__ float red[256];
    float n2 = 0.f;
    for (int i = 0; i < nparts; i++) n2 += vnorm2_parts[i];
    const float inv = 1.f / fmaxf(sqrtf(n2), eps);
    const int o = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < cols; k += 256) {
        const float vh = v_raw[k] * inv;
        if (o == 0) v_out[k] = vh;
        acc += vh * W[(int64_t)o * cols + k];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) t[o] = red[0];
}

// u_new = t / max(|t|, eps), sigma = t . u_new   (single workgroup)
__global__ __launch_bounds__(256) void sn_ufinish_kernel(const float* t, float* u_new, float* sigma, int rows, float eps)
{
    __shared__ float red[256];
    float acc = 0.f;
    for (int o = threadIdx.x; o < rows; o += 256) acc += t[o] * t[o];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float n2 = red[0];
    const float inv = 1.f / fmaxf(sqrtf(n2), eps);
    __syncthreads();
    float dot = 0.f;
    for (int o = threadIdx.x; o < rows; o += 256) { const float un = t[o] * inv; u_new[o] = un; dot += t[o] * un; }
    red[threadIdx.x] = dot;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) sigma[0] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------- attention
#define ATT_MAX_M 256

template <int MT>   // MT = M / 16 key tiles held in accumulators
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* theta, const float* phi, const float* g, float* out,
                                                            int Q, int M, int D, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 16;            // this wave's 16 query rows
    float* P = reinterpret_cast<float*>(smem) + wave * 16 * (ATT_MAX_M + 4);   // 16 x M strip (row pitch M + 4: conflict-free column reads)
    const int pitch = ATT_MAX_M + 4;
    if (q0 >= Q) return;                                     // whole wave out of range (Q % 16 == 0)
    const float* Qp = theta + ((int64_t)n * Q + q0) * D;
    const float* Kp = phi + (int64_t)n * M * D;
    const float* Vp = g + (int64_t)n * M * DV;
    const int fr = lane & 15, fk = lane >> 4;                // A: row fr, k = fk ; B: k = fk, col fr ; C: col fr, rows 4*fk + reg

    // ---- S = Q K^T : MT accumulator tiles of 16 queries x 16 keys
    float4_t S[MT];
#pragma unroll
    for (int j = 0; j < MT; j++) S[j] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += 4) {
        const float a = Qp[(int64_t)fr * D + k0 + fk];
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const float b = Kp[(int64_t)(16 * j + fr) * D + k0 + fk];       // B[k][col] = K[col][k]
            S[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, S[j], 0, 0, 0);
        }
    }
    // ---- row softmax: row r = 4*fk + e is spread over the 16 lanes sharing fk (cols) and the MT tiles
    float mx[4], sum[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float m = S[0][e];
#pragma unroll
        for (int j = 1; j < MT; j++) m = fmaxf(m, S[j][e]);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        mx[e] = m;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MT; j++) { const float pv = expf(S[j][e] - mx[e]); S[j][e] = pv; s += pv; }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
        sum[e] = s;
    }
#pragma unroll
    for (int j = 0; j < MT; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) P[(4 * fk + e) * pitch + 16 * j + fr] = S[j][e] / sum[e];
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wave's own LDS writes are visible to its own later reads
    __builtin_amdgcn_wave_barrier();

    // ---- O = P V, 16 output columns at a time
    float* Op = out + ((int64_t)n * Q + q0) * DV;
    for (int c0 = 0; c0 < DV; c0 += 16) {
        float4_t O = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int m0 = 0; m0 < M; m0 += 4) {
            const float a = P[fr * pitch + m0 + fk];                         // A[row fr][k = m0 + fk]
            const float b = Vp[(int64_t)(m0 + fk) * DV + c0 + fr];           // B[k][col fr]
            O = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, O, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) Op[(int64_t)(4 * fk + e) * DV + c0 + fr] = O[e];
    }
}

// ---- backward, first order (no [N, Q, M] map in HBM): the standard two-pass split, everything recomputed from theta / phi / g.
// Pass A (one wave per 16 queries, like the forward): S, P = softmax(S), dP = dO V^T, delta = rowsum(dP o P), dS = P o (dP - delta),
// dtheta = dS K; also writes the row statistics lse = max + log(sum) and delta for pass B.
template <int MT>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const float* theta, const float* phi, const float* g, const float* dout,
                                                               float* dtheta, float* lse, float* delta, int Q, int M, int D, int DV)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    const int pitch = ATT_MAX_M + 4;
    float* P = reinterpret_cast<float*>(smem) + wave * 16 * pitch;
    if (q0 >= Q) return;
    const float* Qp = theta + ((int64_t)n * Q + q0) * D;
    const float* Kp = phi + (int64_t)n * M * D;
    const float* Vp = g + (int64_t)n * M * DV;
    const float* dOp = dout + ((int64_t)n * Q + q0) * DV;
    const int fr = lane & 15, fk = lane >> 4;

    float4_t S[MT], dP[MT];
#pragma unroll
    for (int j = 0; j < MT; j++) { S[j] = float4_t{0.f, 0.f, 0.f, 0.f}; dP[j] = float4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int k0 = 0; k0 < D; k0 += 4) {
        const float a = Qp[(int64_t)fr * D + k0 + fk];
#pragma unroll
        for (int j = 0; j < MT; j++) S[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Kp[(int64_t)(16 * j + fr) * D + k0 + fk], S[j], 0, 0, 0);
    }
    for (int k0 = 0; k0 < DV; k0 += 4) {                    // dP = dO V^T
        const float a = dOp[(int64_t)fr * DV + k0 + fk];
#pragma unroll
        for (int j = 0; j < MT; j++) dP[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Vp[(int64_t)(16 * j + fr) * DV + k0 + fk], dP[j], 0, 0, 0);
    }
    float mx[4], sum[4], dl[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float m = S[0][e];
#pragma unroll
        for (int j = 1; j < MT; j++) m = fmaxf(m, S[j][e]);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        mx[e] = m;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MT; j++) { const float pv = expf(S[j][e] - m); S[j][e] = pv; s += pv; }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
        sum[e] = s;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < MT; j++) { S[j][e] = S[j][e] / s; d += S[j][e] * dP[j][e]; }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) d += __shfl_xor(d, off, 64);
        dl[e] = d;
    }
    if (fr == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            lse[(int64_t)n * Q + q0 + 4 * fk + e] = mx[e] + logf(sum[e]);
            delta[(int64_t)n * Q + q0 + 4 * fk + e] = dl[e];
        }
    }
#pragma unroll
    for (int j = 0; j < MT; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) P[(4 * fk + e) * pitch + 16 * j + fr] = S[j][e] * (dP[j][e] - dl[e]);       // dS
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float* dQp = dtheta + ((int64_t)n * Q + q0) * D;
    for (int c0 = 0; c0 < D; c0 += 16) {                    // dtheta = dS K
        const bool colok = c0 + fr < D;
        float4_t O = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int m0 = 0; m0 < M; m0 += 4) {
            const float a = P[fr * pitch + m0 + fk];
            const float b = colok ? Kp[(int64_t)(m0 + fk) * D + c0 + fr] : 0.f;
            O = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, O, 0, 0, 0);
        }
        if (colok) {
#pragma unroll
            for (int e = 0; e < 4; e++) dQp[(int64_t)(4 * fk + e) * D + c0 + fr] = O[e];
        }
    }
}

// Pass B (one wave per 16 keys, walking over all query blocks): P^T = exp(K Q^T - lse), dP^T = V dO^T, dS^T = P^T o (dP^T - delta),
// dg += P^T dO, dphi += dS^T Q.  Sums over the queries stay inside one wave: fixed order, reproducible.
template <int DVT, int DT>     // DV / 16 and ceil(D / 16) accumulator tiles
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const float* theta, const float* phi, const float* g, const float* dout,
                                                                const float* lse, const float* delta, float* dphi, float* dg,
                                                                int Q, int M, int D, int DV)
{
    __shared__ float tiles[4][2][16 * 17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int m0 = (blockIdx.x * 4 + wave) * 16;
    if (m0 >= M) return;
    float* Pt = tiles[wave][0];
    float* dSt = tiles[wave][1];
    const float* Qp = theta + (int64_t)n * Q * D;
    const float* Kp = phi + ((int64_t)n * M + m0) * D;
    const float* Vp = g + ((int64_t)n * M + m0) * DV;
    const float* dOp = dout + (int64_t)n * Q * DV;
    const int fr = lane & 15, fk = lane >> 4;
    float4_t aV[DVT], aK[DT];
#pragma unroll
    for (int i = 0; i < DVT; i++) aV[i] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DT; i++) aK[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int qb = 0; qb < Q; qb += 16) {
        float4_t St = float4_t{0.f, 0.f, 0.f, 0.f}, dPt = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < D; k0 += 4)                   // S^T tile: rows = keys, cols = queries
            St = __builtin_amdgcn_mfma_f32_16x16x4f32(Kp[(int64_t)fr * D + k0 + fk], Qp[(int64_t)(qb + fr) * D + k0 + fk], St, 0, 0, 0);
        for (int k0 = 0; k0 < DV; k0 += 4)
            dPt = __builtin_amdgcn_mfma_f32_16x16x4f32(Vp[(int64_t)fr * DV + k0 + fk], dOp[(int64_t)(qb + fr) * DV + k0 + fk], dPt, 0, 0, 0);
        const float l = lse[(int64_t)n * Q + qb + fr], dl = delta[(int64_t)n * Q + qb + fr];      // statistics of this lane's query column
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float pv = expf(St[e] - l);
            Pt[(4 * fk + e) * 17 + fr] = pv;
            dSt[(4 * fk + e) * 17 + fr] = pv * (dPt[e] - dl);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // k = 4 queries per MFMA
            const float ap = Pt[fr * 17 + 4 * i + fk], as = dSt[fr * 17 + 4 * i + fk];
            const float* dor = dOp + (int64_t)(qb + 4 * i + fk) * DV;
            const float* qr = Qp + (int64_t)(qb + 4 * i + fk) * D;
#pragma unroll
            for (int c = 0; c < DVT; c++) aV[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, dor[16 * c + fr], aV[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < DT; c++) aK[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, (16 * c + fr < D) ? qr[16 * c + fr] : 0.f, aK[c], 0, 0, 0);
        }
    }
    float* dVp = dg + ((int64_t)n * M + m0) * DV;
    float* dKp = dphi + ((int64_t)n * M + m0) * D;
#pragma unroll
    for (int c = 0; c < DVT; c++)
#pragma unroll
        for (int e = 0; e < 4; e++) dVp[(int64_t)(4 * fk + e) * DV + 16 * c + fr] = aV[c][e];
#pragma unroll
    for (int c = 0; c < DT; c++)
        if (16 * c + fr < D) {
#pragma unroll
            for (int e = 0; e < 4; e++) dKp[(int64_t)(4 * fk + e) * D + 16 * c + fr] = aK[c][e];
        }
}

template <int MT>
static int launch_att(const float* theta, const float* phi, const float* g, float* out, int N, int Q, int M, int D, int DV, hipStream_t s)
{
    const int lds = 4 * 16 * (ATT_MAX_M + 4) * (int)sizeof(float);
    if (!SBG_RAISE_LDS_ONCE(attention_fwd_kernel<MT>, lds)) return sbg_fail(SBG_ERR_LAUNCH, "attention_fwd: cannot raise the dynamic LDS limit to %d bytes", lds);
    SBG_LAUNCH((attention_fwd_kernel<MT>), dim3((Q + 63) / 64, N), dim3(256), lds, s, theta, phi, g, out, Q, M, D, DV);
    return SBG_OK;
}

} // namespace

extern "C" int64_t sbg_sn_workspace(int rows, int cols)
{
    const int64_t nrb = (rows + SN_ROWS_PER_BLOCK - 1) / SN_ROWS_PER_BLOCK;
    const int64_t kb = (cols + 255) / 256;
    return (nrb * cols + cols + kb + rows + 16) * (int64_t)sizeof(float);
}

extern "C" int sbg_sn_power_iteration(const float* W, const float* u, float* v, float* u_new, float* sigma, void* workspace,
                                      int rows, int cols, float eps, sbg_stream_t stream)
{
    SBG_CHECK(W && u && v && u_new && sigma && workspace, "sn_power_iteration: null pointer");
    SBG_CHECK(rows >= 1 && cols >= 1, "sn_power_iteration: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int nrb = (rows + SN_ROWS_PER_BLOCK - 1) / SN_ROWS_PER_BLOCK, kb = (cols + 255) / 256;
    float* partial = (float*)workspace;
    float* v_raw = partial + (int64_t)nrb * cols;
    float* vn2 = v_raw + cols;
    float* t = vn2 + kb;
    SbgProfScope prof(s, SBG_K_SN_POWER, 4.0 * rows * (double)cols, 8.0 * rows * (double)cols, {rows, cols});
    SBG_LAUNCH(sn_colsum_kernel, dim3(kb, nrb), dim3(256), 0, s, W, u, partial, rows, cols);
    SBG_LAUNCH(sn_vfinish_kernel, dim3(kb), dim3(256), 0, s, partial, v_raw, vn2, cols, nrb);
    SBG_LAUNCH(sn_rowsum_kernel, dim3(rows), dim3(256), 0, s, W, v_raw, vn2, kb, v, t, rows, cols, eps);
    SBG_LAUNCH(sn_ufinish_kernel, dim3(1), dim3(256), 0, s, t, u_new, sigma, rows, eps);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

extern "C" int sbg_attention_supported(int Q, int M, int D, int DV)
{
    return (Q >= 16 && Q % 16 == 0 && M >= 16 && M % 16 == 0 && M <= ATT_MAX_M && D >= 4 && D % 4 == 0 && DV >= 16 && DV % 16 == 0) ? 1 : 0;
}

template <int MT>
static int launch_att_bwd_dq(const float* theta, const float* phi, const float* g, const float* dout, float* dtheta, float* lse, float* delta,
                             int N, int Q, int M, int D, int DV, hipStream_t s)
{
    const int lds = 4 * 16 * (ATT_MAX_M + 4) * (int)sizeof(float);
    if (!SBG_RAISE_LDS_ONCE(attention_bwd_dq_kernel<MT>, lds)) return sbg_fail(SBG_ERR_LAUNCH, "attention_bwd: cannot raise the dynamic LDS limit to %d bytes", lds);
    SBG_LAUNCH((attention_bwd_dq_kernel<MT>), dim3((Q + 63) / 64, N), dim3(256), lds, s, theta, phi, g, dout, dtheta, lse, delta, Q, M, D, DV);
    return SBG_OK;
}

template <int DVT>
static int launch_att_bwd_dkv(const float* theta, const float* phi, const float* g, const float* dout, const float* lse, const float* delta,
                              float* dphi, float* dg, int N, int Q, int M, int D, int DV, hipStream_t s)
{
    const dim3 grid((M + 63) / 64, N);
    switch ((D + 15) / 16) {
        case 1: SBG_LAUNCH((attention_bwd_dkv_kernel<DVT, 1>), grid, dim3(256), 0, s, theta, phi, g, dout, lse, delta, dphi, dg, Q, M, D, DV); break;
        case 2: SBG_LAUNCH((attention_bwd_dkv_kernel<DVT, 2>), grid, dim3(256), 0, s, theta, phi, g, dout, lse, delta, dphi, dg, Q, M, D, DV); break;
        case 4: SBG_LAUNCH((attention_bwd_dkv_kernel<DVT, 4>), grid, dim3(256), 0, s, theta, phi, g, dout, lse, delta, dphi, dg, Q, M, D, DV); break;
        default: return sbg_fail(SBG_ERR_UNSUPPORTED, "attention_bwd: D = %d needs 1, 2 or 4 column tiles of 16", D);
    }
    return SBG_OK;
}

extern "C" int sbg_attention_bwd_supported(int Q, int M, int D, int DV)
{
    const int dt = (D + 15) / 16, dvt = DV / 16;
    return (sbg_attention_supported(Q, M, D, DV) && (dt == 1 || dt == 2 || dt == 4) && (dvt == 1 || dvt == 2 || dvt == 4 || dvt == 8 || dvt == 16)) ? 1 : 0;
}

extern "C" int64_t sbg_attention_bwd_workspace(int N, int Q) { return 2 * (int64_t)N * Q * (int64_t)sizeof(float); }

// First-order gradients of out = softmax(theta phi^T) g.  workspace: sbg_attention_bwd_workspace(N, Q) bytes (row statistics).
extern "C" int sbg_attention_bwd(const float* theta, const float* phi, const float* g, const float* dout, float* dtheta, float* dphi, float* dg,
                                 void* workspace, int N, int Q, int M, int D, int DV, sbg_stream_t stream)
{
    SBG_CHECK(theta && phi && g && dout && dtheta && dphi && dg && workspace, "attention_bwd: null pointer");
    SBG_CHECK(sbg_attention_bwd_supported(Q, M, D, DV), "attention_bwd: unsupported shape Q=%d M=%d D=%d DV=%d", Q, M, D, DV);
    if (N == 0) return SBG_OK;
    hipStream_t s = (hipStream_t)stream;
    float* lse = (float*)workspace;
    float* delta = lse + (int64_t)N * Q;
    SbgProfScope prof(s, SBG_K_ATTENTION, 2.0 * N * (double)Q * M * (3.0 * D + 3.0 * DV), 4.0 * N * (2.0 * Q * D + 2.0 * M * (D + DV) + 2.0 * Q * DV), {N, Q, M, D, DV, 1});
    int rc = SBG_OK;
    switch (M / 16) {
        case 1:  rc = launch_att_bwd_dq<1>(theta, phi, g, dout, dtheta, lse, delta, N, Q, M, D, DV, s); break;
        case 2:  rc = launch_att_bwd_dq<2>(theta, phi, g, dout, dtheta, lse, delta, N, Q, M, D, DV, s); break;
        case 4:  rc = launch_att_bwd_dq<4>(theta, phi, g, dout, dtheta, lse, delta, N, Q, M, D, DV, s); break;
        case 8:  rc = launch_att_bwd_dq<8>(theta, phi, g, dout, dtheta, lse, delta, N, Q, M, D, DV, s); break;
        case 16: rc = launch_att_bwd_dq<16>(theta, phi, g, dout, dtheta, lse, delta, N, Q, M, D, DV, s); break;
        default: return sbg_fail(SBG_ERR_UNSUPPORTED, "attention_bwd: M / 16 must be 1, 2, 4, 8 or 16 (got M = %d)", M);
    }
    if (rc != SBG_OK) return rc;
    switch (DV / 16) {
        case 1:  rc = launch_att_bwd_dkv<1>(theta, phi, g, dout, lse, delta, dphi, dg, N, Q, M, D, DV, s); break;
        case 2:  rc = launch_att_bwd_dkv<2>(theta, phi, g, dout, lse, delta, dphi, dg, N, Q, M, D, DV, s); break;
        case 4:  rc = launch_att_bwd_dkv<4>(theta, phi, g, dout, lse, delta, dphi, dg, N, Q, M, D, DV, s); break;
        case 8:  rc = launch_att_bwd_dkv<8>(theta, phi, g, dout, lse, delta, dphi, dg, N, Q, M, D, DV, s); break;
        case 16: rc = launch_att_bwd_dkv<16>(theta, phi, g, dout, lse, delta, dphi, dg, N, Q, M, D, DV, s); break;
        default: return sbg_fail(SBG_ERR_UNSUPPORTED, "attention_bwd: DV / 16 must be 1, 2, 4, 8 or 16 (got DV = %d)", DV);
    }
    if (rc != SBG_OK) return rc;
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

extern "C" int sbg_attention_fwd(const float* theta, const float* phi, const float* g, float* out, int N, int Q, int M, int D, int DV,
                                 sbg_stream_t stream)
{
    SBG_CHECK(theta && phi && g && out, "attention_fwd: null pointer");
    SBG_CHECK(sbg_attention_supported(Q, M, D, DV), "attention_fwd: unsupported shape Q=%d M=%d D=%d DV=%d", Q, M, D, DV);
    if (N == 0) return SBG_OK;
    hipStream_t s = (hipStream_t)stream;
    SbgProfScope prof(s, SBG_K_ATTENTION, 2.0 * N * (double)Q * M * (D + DV), 4.0 * N * ((double)Q * D + (double)M * (D + DV) + (double)Q * DV), {N, Q, M, D, DV});
    int rc = SBG_OK;
    switch (M / 16) {
        case 1:  rc = launch_att<1>(theta, phi, g, out, N, Q, M, D, DV, s); break;
        case 2:  rc = launch_att<2>(theta, phi, g, out, N, Q, M, D, DV, s); break;
        case 4:  rc = launch_att<4>(theta, phi, g, out, N, Q, M, D, DV, s); break;
        case 8:  rc = launch_att<8>(theta, phi, g, out, N, Q, M, D, DV, s); break;
        case 16: rc = launch_att<16>(theta, phi, g, out, N, Q, M, D, DV, s); break;
        default: return sbg_fail(SBG_ERR_UNSUPPORTED, "attention_fwd: M / 16 must be 1, 2, 4, 8 or 16 (got M = %d)", M);
    }
    if (rc != SBG_OK) return rc;
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}
