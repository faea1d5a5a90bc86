/*
 * sbg_hip.h -- C ABI of libsbg_hip.so, the MI355X (gfx950) hot path of the
 * Style-Big-GAN custom-op layer.
 *
 * Every entry point is `extern "C"`, takes plain device pointers, sizes and a
 * hipStream_t (passed as void*), never blocks the host, never allocates, and
 * returns 0 on success or a non-zero sbg_status; sbg_last_error() returns a
 * thread-local human readable message for the last failing call of the calling
 * thread.  Inputs are borrowed and never written; outputs are caller-allocated.
 * Entry points are re-entrant (autograd worker threads call them concurrently).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the reference checkout).
 */
#ifndef SBG_HIP_H
#define SBG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sbg_stream_t;            /* hipStream_t */

enum sbg_status {
    SBG_OK = 0,
    SBG_ERR_INVALID = 1,               /* argument validation failed (reference: TORCH_CHECK) */
    SBG_ERR_UNSUPPORTED = 2,           /* valid but not implemented by this build            */
    SBG_ERR_LAUNCH = 3                 /* hipLaunchKernel / runtime error                     */
};

enum sbg_dtype { SBG_F32 = 0, SBG_F16 = 1, SBG_BF16 = 2 };

/* activation ids follow the reference's `cuda_idx`
 * (stylegan2ada/torch_utils/ops/bias_act.py:23-33). */
enum sbg_act {
    SBG_ACT_LINEAR = 1, SBG_ACT_RELU = 2, SBG_ACT_LRELU = 3, SBG_ACT_TANH = 4, SBG_ACT_SIGMOID = 5,
    SBG_ACT_ELU = 6, SBG_ACT_SELU = 7, SBG_ACT_SOFTPLUS = 8, SBG_ACT_SWISH = 9
};

int         sbg_version(void);
const char* sbg_last_error(void);

/* ------------------------------------------------------------------------------------------
 * bias_act: y = clamp(act(x + b[(i / stepB) % sizeB]) * gain)  and its 1st / 2nd derivative forms.
 * Replaces the pybind entry `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`
 * (stylegan2ada/torch_utils/ops/bias_act.cpp:32-90) with the fields of `bias_act_kernel_params`
 * (stylegan2ada/torch_utils/ops/bias_act.h:12-31) minus the launch-tuning ones.
 * NULL pointer == "absent" (the reference passes an empty tensor).  All tensors share x's dense
 * layout; `dtype` applies to x, b, xref, yref, dy, y.  clamp < 0 disables clamping. */
int sbg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                 void* y, int dtype, int grad, int act, float alpha, float gain, float clamp,
                 int64_t sizeX, int sizeB, int64_t stepB, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d: pad -> zero-insert upsample -> 2-D FIR -> decimate.
 * Replaces the pybind entry `upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
 * flip, gain)` (stylegan2ada/torch_utils/ops/upfirdn2d.cpp:16-94) with the fields of
 * `upfirdn2d_kernel_params` (stylegan2ada/torch_utils/ops/upfirdn2d.h:14-40).
 * Sizes/strides are [W, H, C, N] in elements like the reference's int4 fields; the filter is
 * float32 [fh, fw] with element strides; padx1/pady1 are implied by the output size. */
typedef struct sbg_upfirdn2d_params {
    const void*  x;
    const float* f;
    void*        y;
    int dtype;
    int upx, upy, downx, downy, padx0, pady0;
    int flip;
    float gain;
    int     inSize[4];      int64_t inStride[4];
    int     filterSize[2];  int     filterStride[2];      /* [W, H] */
    int     outSize[4];     int64_t outStride[4];
    /* Optional fused tail (the `fma(x, dcoefs, noise)` + `bias_act` that follow the low-pass of an up-sampling synthesis layer,
     * train_parts/generators.py:84-88,328):  y = clamp(act(fir * gain * oscale[n*C + c] + noise[n*noise_stride_n + oy*outW + ox] + bias[c]) * act_gain).
     * fp32 arrays, each may be NULL; act in {0 = no tail, linear, relu, lrelu}; clamp < 0 disables.  Only the matrix-core FIR path
     * (16-bit channel-minor tensors, up = down = 1, 4x4 exact taps, C % 64 == 0) applies it: sbg_upfirdn2d_tail_supported(). */
    const float* oscale; const float* noise; int64_t noise_stride_n; const float* bias;
    int act; float alpha, act_gain, clamp;
    int     filter_exact16;  /* hint: every tap of f is exactly representable in `dtype` (bf16 / f16), so the filter may be fed to the
                                matrix cores without rounding ([1,3,3,1]-type filters are); 0 = unknown -> fp32 vector path */
    /* Optional backward tail (excludes the forward tail above).  In the backward of `bias_act -> low-pass` -- a discriminator block's conv0 followed by
     * the filter of its down-sampling conv1, train_parts/discriminators.py:286-291 -- this launch is the transposed low-pass and its result is the
     * gradient w.r.t. the bias_act's OUTPUT `dact_y` (saved; this launch's output geometry).  With dact_y != NULL the kernel multiplies that result by
     * the slope of clamp(act(.) * dact_gain) at dact_y (bias_act.py:159-210 for the piecewise-linear activations: dact_gain above zero,
     * dact_gain * dact_alpha below for lrelu, zero where |y| >= dact_clamp; dact_clamp < 0 disables) and accumulates it per channel -- the bias
     * gradient: dact_partial[r][64] for r < sbg_upfirdn2d_dact_rows(), row r = ((n * ysegs + s) * xstrips + x) * (C / 64) + channel block, to be summed
     * over everything but the channel block by the caller.  Sliding-window matrix-core FIR only: _dact_rows() returns -1 for other launches. */
    const void* dact_y; float* dact_partial; int dact_act; float dact_alpha, dact_gain, dact_clamp;
    /* Forward tail only: the finished value is multiplied by post_scale[n*C + c] -- the style modulation `x * styles` of the layer that reads this
     * output next (generators.py:79), for passes in which nothing else reads it (inference-mode generator passes).  NULL = none. */
    const float* post_scale;
} sbg_upfirdn2d_params;
int sbg_upfirdn2d_tail_supported(const sbg_upfirdn2d_params* p);
int64_t sbg_upfirdn2d_dact_rows(const sbg_upfirdn2d_params* p);
int sbg_upfirdn2d(const sbg_upfirdn2d_params* p, sbg_stream_t stream);

/* Separable filter (1-D `f` of `taps` taps applied along both axes) in one launch -- replaces the reference's two plugin calls
 * with sqrt(gain) each for a rank-1 filter (stylegan2ada/torch_utils/ops/upfirdn2d.py:236-240).  Dense planar fp32 planes
 * x [M, IH, IW] -> y [M, OH, OW]; up, down in {1, 2} on both axes; OH = (IH*up + pady0 + pady1 - taps) / down + 1 etc. is the
 * caller's to compute (as for sbg_upfirdn2d); `gain` is the total gain.  _supported() says whether (up, down, taps) fits. */
int sbg_upfirdn2d_separable_supported(int up, int down, int taps);
int sbg_upfirdn2d_separable(const float* x, const float* f, float* y, int M, int IH, int IW, int OH, int OW, int taps,
                            int up, int down, int padx0, int pady0, int flip, float gain, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (channels-last activations).
 * Replaces the aten/cuDNN calls behind `conv2d_gradfix.conv2d / conv_transpose2d`
 * (stylegan2ada/torch_utils/ops/conv2d_gradfix.py:35-45,107-118): forward, data-gradient and the
 * four sub-pixel phases of a stride-2 transposed convolution are all expressed as
 *
 *   y[n, oy*ysh + yoh, ox*ysw + yow, co] (+)= sum_{t < ntaps} sum_{ci}
 *        x[n, oy*stride + dy[t], ox*stride + dx[t], ci] * w[wslab[t]][co][ci]
 *
 * over the launch's output grid OH x OW (out-of-range input pixels read as zero).
 * x: [N, IH, IW, Cin] channel-minor (Cin % 8 == 0, pixel stride xs_w etc. in elements),
 * w: packed [nslabs][Cout][Cin] (same dtype as x), y: channel-minor, dtype `ydtype`.
 * Optional fused epilogue (every pointer may be NULL, act 0 / SBG_ACT_LINEAR with gain 1 and clamp < 0 = identity), the
 * demodulation + noise + bias_act tail of the reference's layers (train_parts/generators.py:84-88,328; ops/bias_act.py:94-123):
 *   y = clamp( act( acc * oscale[n*Cout + co] + noise[n*noise_stride_n + pixel] + bias[co] ) * gain )
 * with fp32 oscale / noise / bias, pixel = oy*OW + ox of the launch grid, act in {linear, relu, lrelu}.
 * `accumulate` adds into the existing y (fp32 y only, no epilogue).
 * Kernel choice is internal (csrc/conv_k64.hip): stride-1 launches with the nine taps of a 3x3 window on a tile-aligned grid
 * take the persistent halo-staged kernel, everything else the gather (im2col-on-the-fly) kernel. */
#define SBG_MAX_TAPS 16
typedef struct sbg_conv_params {
    const void* x; const void* w; void* y;
    const float* oscale;
    int xdtype, ydtype;
    int N, IH, IW, Cin, Cout, OH, OW;
    int64_t xs_n, xs_h, xs_w;
    int64_t ys_n, ys_h, ys_w;
    int64_t ws_slab, ws_co;
    int stride;
    int ntaps;
    int tap_dy[SBG_MAX_TAPS], tap_dx[SBG_MAX_TAPS], tap_slab[SBG_MAX_TAPS];
    int accumulate;
    const float* bias; const float* noise; int64_t noise_stride_n;
    int act; float alpha, gain, clamp;
    /* Optional split of the reduction (taps x channels) over `ksplit` workgroups per output tile, for launches with too few
     * output tiles to fill the chip (the 4x4 / 8x8 blocks): partial fp32 results go to `workspace`
     * (>= sbg_conv2d_igemm_workspace() bytes) and a second kernel sums them in a fixed order into y (bitwise reproducible).
     * Needs a dense channel-minor y; a fused epilogue and a 16-bit output are applied by that second kernel (then without `accumulate`,
     * Cout % 8 == 0); ksplit <= 1 or workspace == NULL = off. */
    void* workspace; int ksplit;
    /* Optional phases: nphase in 2..4 makes ONE launch compute several output sub-grids of the same input -- the s x s phases of a
     * stride-s transposed convolution, whose tiles then share the input through L2 instead of streaming it from HBM once per phase.
     * Phase i uses the next ph_ntaps[i] taps of the tap list (sum <= ntaps), the output grid ph_oh[i] x ph_ow[i] (instead of OH x OW)
     * and writes at y + ph_yoff[i] elements with the common ys_* strides.  nphase <= 1 = off.  No fused epilogue / split with phases. */
    int nphase; int ph_ntaps[4], ph_oh[4], ph_ow[4]; int64_t ph_yoff[4];
} sbg_conv_params;
int64_t sbg_conv2d_igemm_workspace(const sbg_conv_params* p);
int     sbg_conv2d_igemm(const sbg_conv_params* p, sbg_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Weight gradient (replaces `aten::cudnn_convolution_backward_weight` /
 * `..._transpose_backward_weight`, conv2d_gradfix.py:140-147):
 *
 *   out[t][ca][cb] = sum_{n, py, px} a[n, py, px, ca] * b[n, py*stride + dy[t], px*stride + dx[t], cb]
 *
 * a: [N, PH, PW, Ca], b: [N, BH, BW, Cb], both channel-minor with Ca % 8 == Cb % 8 == 0, same dtype;
 * out: fp32 [ntaps][Ca][Cb].  The pixel sum is split over `nsplit` workgroups that write fp32
 * partial slabs into `workspace` (>= sbg_conv2d_wgrad_workspace() bytes) and a second kernel reduces
 * them in a fixed order (bitwise reproducible). `accumulate` adds into the existing out. */
typedef struct sbg_wgrad_params {
    const void* a; const void* b; float* out; void* workspace;
    int dtype;
    int N, PH, PW, Ca, BH, BW, Cb;
    int64_t as_n, as_h, as_w;
    int64_t bs_n, bs_h, bs_w;
    int stride;
    int ntaps;
    int tap_dy[SBG_MAX_TAPS], tap_dx[SBG_MAX_TAPS];
    int accumulate;
} sbg_wgrad_params;
int64_t sbg_conv2d_wgrad_workspace(const sbg_wgrad_params* p);
int     sbg_conv2d_wgrad(const sbg_wgrad_params* p, sbg_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Per-sample channel scaling with optional per-pixel addend (the modulation `x * styles`, the demodulation
 * + noise `fma(x, dcoefs, noise)` of train_parts/generators.py:79-88 and stylegan2ada/torch_utils/ops/fma.py:15):
 *   y[n,c,p] = x[n,c,p] * a[n*C + c] (+ z[n*z_stride_n + p])         a, z: fp32; p = pixel index in [0, HW)
 * layout: 0 = planar [N][C][HW], 1 = channel-minor [N][HW][C]; x and y dense in that layout. */
int sbg_scale_nc(const void* x, const float* a, const float* z, void* y, int dtype, int layout,
                 int N, int C, int64_t HW, int64_t z_stride_n, sbg_stream_t stream);
/* Same with a per-sample, per-channel addend (the normalise-and-modulate step of biggan/layers.py `ccbn` / `bn` / `fused_bn`
 * :173-187,306-325 once the statistics are known):  y[n,c,p] = x[n,c,p] * a[n*C + c] + b[n*C + c]. */
int sbg_scale_shift_nc(const void* x, const float* a, const float* b, void* y, int dtype, int layout,
                       int N, int C, int64_t HW, sbg_stream_t stream);

/* Per-sample, per-channel dot product over pixels (the gradient of the scaling above w.r.t. `a`, and -- with
 * v == NULL -- the bias gradient `dx.sum([2, 3])` of bias_act.py:172-173):
 *   partial[s][n*C + c] = sum_{p in split s} u[n,c,p] * (v ? v[n,c,p] : 1)        partial: fp32 [nsplit][N][C]
 * The caller sums `partial` over s (fixed order => reproducible).  sbg_dot_hw_splits() returns nsplit. */
int sbg_dot_hw_splits(int layout, int N, int C, int64_t HW);
int sbg_dot_hw(const void* u, const void* v, float* partial, int dtype, int layout,
               int N, int C, int64_t HW, sbg_stream_t stream);
/* r[0][n*C + c] = sum over the H*W plane of x, r[1][n*C + c] = sum of x^2 -- both moments of a PLANAR tensor [N, C, H*W] in one pass (fp32
 * accumulation, fixed order).  The batch statistics of BigGAN's normalisation layers (biggan/layers.py:188-205 `manual_bn`,
 * sync_batchnorm/batchnorm.py:71-79: `input_sum`, `input_ssum`).  r: fp32 [2, N*C]. */
int sbg_moments_hw(const void* x, float* r, int dtype, int N, int C, int64_t HW, sbg_stream_t stream);

/* Many small fp32 matrix products in one launch (csrc/grouped_gemm.hip): the per-layer affine maps of a synthesis network
 * (train_parts/generators.py:333 `styles = self.affine(w)` in every SynthesisLayer / ToRGBLayer; FullyConnectedLayer.forward :117-131:
 * `torch.addmm(b.unsqueeze(0), x, w.t())` with the gains folded in) and their gradients -- ~20 + ~60 separate GEMM / reduction launches per
 * pass in the reference.  Problem p:  C[m][n] = sum_{t < nterms} alpha_t * sum_k A_t[m][k] B_t[k][n]  (+ bias[n] * bias_scale),
 * every operand addressed by element strides (row stride, column stride), so transposes and slices of larger tensors cost nothing;
 * `rowsum` (optional): rowsum[m] = rowsum_scale * sum_k A_0[m][k]  (the bias gradient next to the weight gradient).
 * fp32, fixed summation order (ascending k, term 0 then term 1).  Outputs of different problems must not overlap. */
typedef struct sbg_gg_problem {
    const float *a0, *b0, *a1, *b1;         /* term 1 is ignored when nterms == 1 */
    float* c; const float* bias; float* rowsum;
    int64_t a0_rs, a0_cs, b0_rs, b0_cs, a1_rs, a1_cs, b1_rs, b1_cs, c_rs, c_cs;
    int M, N, K0, K1, nterms;
    float alpha0, alpha1, bias_scale, rowsum_scale;
} sbg_gg_problem;
int sbg_grouped_gemm(const sbg_gg_problem* problems, int count, sbg_stream_t stream);

/* Both gradients of y = x * a[n, c] (the style modulation in front of a convolution, train_parts/generators.py:79; autograd's
 * `dy * a` and `(dy * x).sum([2, 3])`) in ONE pass over u = dy and v = x, channel-minor tensors with C / 8 dividing 256:
 *   y[n,p,c] = u[n,p,c] * scale[n*C + c]     and     partial[s][n*C + c] as sbg_dot_hw(u, v). */
int sbg_dot_hw_scale_supported(int C);
int sbg_dot_hw_scale(const void* u, const void* v, const float* scale, void* y, float* partial, int dtype,
                     int N, int C, int64_t HW, sbg_stream_t stream);


/* Backward head of the fused modulated-convolution layer (train_parts/generators.py:79-88 + :328, i.e. modulated_conv2d's
 * `fma(x, dcoefs, noise)` followed by bias_act): with y = clamp(act(c * dcoef[n,o] + noise[n,p] + bias[o]) * gain) saved,
 * one pass over (dy, y) writes
 *   d2[n,o,p]            = d1 * dcoef[n,o],  d1 = dy * gain * (y > 0 ? 1 : alpha) * [|y| < clamp]   (bias_act.py:159-210)
 *   partial[0][s][n][o]  = sum_p d1                         (bias gradient)
 *   partial[1][s][n][o]  = sum_p d1 * (pre - noise - bias)  (dcoef[n,o] * demodulation gradient; pre is recovered from y)
 *   dnoise[n,p]          = sum_o d1                         (optional, NULL to skip)
 * Channel-minor [N][HW][C] tensors, C / 8 a power of two <= 64; partial: fp32 [2][nsplit][N][C], nsplit = sbg_dot_hw_splits(1, N, C, HW).
 * act in {linear, relu, lrelu}; clamp < 0 disables. */
int sbg_modconv_bwd_supported(int C);
int sbg_modconv_bwd(const void* dy, const void* y, const float* dcoef, const float* noise, const float* bias,
                    void* d2, float* partial, float* dnoise, int dtype, int N, int C, int64_t HW, int64_t noise_stride_n,
                    int act, float alpha, float gain, float clamp, sbg_stream_t stream);
/* Same, for a y that feeds a style-modulated convolution and nothing else (a synthesis block's conv0 output into its conv1, generators.py:462-463):
 * `dy` is then the gradient w.r.t. y * prescale[n, c] (what that convolution's data gradient delivers); the pass also takes
 * partial3[s][n*C + c] = sum_p dy * y (the gradient of prescale) and continues with dy * prescale -- autograd's `dy * s`, `(dy * x).sum([2, 3])` and the
 * backward head above in one pass over (dy, y). */
int sbg_modconv_bwd_prescaled(const void* dy, const void* y, const float* prescale, const float* dcoef, const float* noise, const float* bias,
                              void* d2, float* partial, float* partial3, float* dnoise, int dtype, int N, int C, int64_t HW,
                              int64_t noise_stride_n, int act, float alpha, float gain, float clamp, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Spectral norm, one power iteration with one singular vector (biggan/layers.py `power_iteration` :28-50, `SN.W_` :87-99):
 *   v = normalize(u W), u' = normalize(v W^T), sigma = (v W^T) . u'        W: fp32 [rows, cols] row-major, u: fp32 [rows]
 * Writes v [cols], u_new [rows] (both normalised with F.normalize's max(norm, eps)) and sigma [1]; `workspace` holds
 * sbg_sn_workspace(rows, cols) bytes.  The caller decides whether u_new replaces the stored u (training) -- nothing is
 * updated in place.  d sigma / d W = outer(u_new, v) is applied by the host autograd Function. */
int64_t sbg_sn_workspace(int rows, int cols);
int sbg_sn_power_iteration(const float* W, const float* u, float* v, float* u_new, float* sigma, void* workspace,
                           int rows, int cols, float eps, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Minibatch standard deviation (train_parts/discriminators.py:313-328): x fp32 [N, C, H, W] dense, N = G * M (sample g*M + m is in group m),
 * C = F * c.  y fp32 [N, C + F, H, W]: y[:, :C] = x and y[g*M + m, C + f] = mean_{cc,h,w} sqrt(var_g x[g*M + m, f*c + cc, h, w] + 1e-8).
 * sbg_mbstd_bwd: first-order dx from (x, dy).  One launch each (the reference: ~10 + ~14 tensor ops on a 1 MB tensor). */
int64_t sbg_mbstd_workspace(int N, int C, int HW, int G, int F);      /* bytes of partial sums sbg_mbstd_fwd needs */
int sbg_mbstd_fwd(const float* x, float* y, void* workspace, int N, int C, int HW, int G, int F, sbg_stream_t stream);
int sbg_mbstd_bwd(const float* x, const float* dy, float* dx, int N, int C, int HW, int G, int F, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 operands for the bf16 matrix cores: hi / mid / lo bf16 split of x, the parts concatenated along one axis in one pass.
 *   x: fp32 [outer, C, inner] dense;  y: bf16 [outer, nseg * C, inner];  y[o, s*C + c, i] = part_{order[s]}(x[o, c, i]),
 *   part_0 = bf16(x), part_1 = bf16(x - part_0), part_2 = bf16(x - part_0 - part_1);  nseg <= 8, order[s] in {0, 1, 2}.
 * (No counterpart in the reference: its fp32 layers go to cuDNN / oneDNN; here they run as six bf16 MFMA products with fp32 accumulation.) */
int sbg_split_bf16_cat(const float* x, void* y, int64_t outer, int64_t C, int64_t inner, int nseg, const int* order, sbg_stream_t stream);
/* the same split of a strided 4-D fp32 view (extents `shape`, element strides `xstrides`), written DENSELY in the view's dimension order with
 * dimension `cat_dim` holding the nseg parts side by side: y[.., s * shape[cat_dim] + d_cat, ..] = part order[s] of x[d].  Replaces the
 * reference's w.permute(..).contiguous() of an fp32 conv weight ahead of the launch (torch_utils/ops/conv2d_gradfix.py:94-146 hands cuDNN
 * the strided tensor). */
int sbg_split_bf16_cat_nd(const float* x, const int64_t* shape, const int64_t* xstrides, int cat_dim, void* y, int nseg, const int* order, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Non-local self-attention core (biggan/layers.py `Attention.forward` :162-166): for every sample n
 *   out[n, q, :] = softmax_m( theta[n, q, :] . phi[n, m, :] ) @ g[n, m, :]
 * theta: fp32 [N, Q, D], phi: fp32 [N, M, D], g: fp32 [N, M, DV], out: fp32 [N, Q, DV], all row-major dense.
 * Exact-fp32 matrix-core kernel (v_mfma_f32_16x16x4_f32); the [Q, M] attention map never touches HBM.
 * Supported: Q % 16 == 0, M % 16 == 0, M <= 256, D % 4 == 0, DV % 16 == 0 (sbg_attention_supported). */
int sbg_attention_supported(int Q, int M, int D, int DV);
int sbg_attention_fwd(const float* theta, const float* phi, const float* g, float* out, int N, int Q, int M, int D, int DV,
                      sbg_stream_t stream);
/* First-order gradients of the same expression (what autograd derives from the reference's bmm / softmax / bmm, layers.py:162-166):
 *   dg = P^T dout,  dS = P o (dout g^T - rowsum(dout o out)),  dtheta = dS phi,  dphi = dS^T theta,   P = softmax(theta phi^T)
 * recomputed from theta / phi / g in two passes (per query tile, per key tile); the [Q, M] map never touches HBM and every sum over
 * queries stays inside one wave (fixed order).  workspace: sbg_attention_bwd_workspace(N, Q) bytes of row statistics.
 * Supported: the forward's shapes with D <= 64 and DV / 16 in {1, 2, 4, 8, 16} (sbg_attention_bwd_supported). */
int sbg_attention_bwd_supported(int Q, int M, int D, int DV);
int64_t sbg_attention_bwd_workspace(int N, int Q);
int sbg_attention_bwd(const float* theta, const float* phi, const float* g, const float* dout, float* dtheta, float* dphi, float* dg,
                      void* workspace, int N, int Q, int M, int D, int DV, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 master weights <-> packed convolution operands.  Replaces the per-call framework chain `w = self.weight * weight_gain`,
 * `w.to(x.dtype)` (train_parts/generators.py:176-179, discriminators.py:115-118) + layout change, and its autograd mirror.
 *   sbg_pack_weight:  out[t][a][b] = cast(w[a*sA + b*sB + kh*sKH + kw*sKW] * gain), t = kh*KW + kw, b zero-padded to Bp; `out` is the
 *                     `w` operand of sbg_conv2d_igemm ([slab][Cout][Cin]); optional w2[a][b] = sum_t (w*gain)^2 (fp32 [A][B], the
 *                     demodulation's sum over taps, generators.py:71-76).
 *   sbg_unpack_wgrad: dw[a*sA + b*sB + kh*sKH + kw*sKW] = gain * dwp[t*tap_stride + a*row_stride + b]  (+ 2 gain^2 w[...] dw2[a][b]
 *                     when dw2 != NULL; dwp may then be NULL): sbg_conv2d_wgrad's fp32 result in the parameter's own layout. */
int sbg_pack_weight(const float* w, void* out, int out_dtype, int A, int B, int KH, int KW,
                    int64_t sA, int64_t sB, int64_t sKH, int64_t sKW, int Bp, float gain, float* w2, sbg_stream_t stream);
int sbg_unpack_wgrad(const float* dwp, int64_t dwp_tap_stride, int64_t dwp_row_stride, float* dw, const float* w, const float* dw2,
                     int A, int B, int KH, int KW, int64_t sA, int64_t sB, int64_t sKH, int64_t sKW, float gain, sbg_stream_t stream);

/* Demodulation coefficients of a modulated convolution from the tap-summed squared weights w2 [O][I] (sbg_pack_weight's by-product):
 *   dcoefs[n, o] = rsqrt(sum_i styles[n, i]^2 * w2[o, i] + eps)        (train_parts/generators.py:71-76: `(w * s).square().sum([2,3,4]) + 1e-8).rsqrt()`
 * and the first-order gradient: g = d loss / d dcoefs -> dstyles [N][I], dw2 [O][I] (either may be NULL).  All fp32, dense. */
int sbg_demod_coefs(const float* styles, const float* w2, float* dcoefs, int N, int O, int I, float eps, sbg_stream_t stream);
int sbg_demod_coefs_bwd(const float* g, const float* dcoefs, const float* styles, const float* w2, float* dstyles, float* dw2,
                        int N, int O, int I, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ToRGB layer (train_parts/generators.py:344-348: modulated 1x1 convolution without demodulation to <= 4 channels + linear bias_act
 * with clamp) as streaming kernels over channel-minor 16-bit x [N, HW, C]:
 *   fwd: y[n, o, p] = clamp(sum_c x[n, p, c] * wmod[n, o, c] + bias[o])      y fp32 planar [N, O, HW]; wmod = w[o, c] * styles[n, c], fp32
 *   bwd: d1 = dy masked by the clamp (from the saved y);  dx[n, p, c] = sum_o d1 * wmod  (may be NULL);
 *        partial[n][blk][o * C + c] = per-workgroup sums of d1[o, p] * x[p, c], then O sums of d1 -- the caller adds the
 *        sbg_torgb_bwd_blocks(N, C, HW) slabs in a fixed order (reproducible) to get d wmod [N, O, C] and d bias.
 * C = 8 * 2^k <= 512, O <= 4 (sbg_torgb_supported).  clamp < 0 disables clamping. */
int sbg_torgb_supported(int C, int O);
int sbg_torgb_bwd_blocks(int N, int C, int64_t HW);
int sbg_torgb_fwd(const void* x, const float* wmod, const float* bias, float* y, int dtype, int N, int C, int O, int64_t HW,
                  float clamp, sbg_stream_t stream);
int sbg_torgb_bwd(const void* x, const float* wmod, const float* dy, const float* y, void* dx, float* partial, int dtype,
                  int N, int C, int O, int64_t HW, float clamp, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fromRGB layer of the discriminator (train_parts/discriminators.py:270-277 -> Conv2dLayer :115-124: 1x1 convolution from <= 4 image
 * channels + bias + activation * gain + clamp) as streaming kernels.  img fp32 planar [N, Ci, HW]; w fp32 [Co][Ci] (weight * weight_gain);
 * y 16-bit channel-minor [N, HW, Co]; act in {linear, relu, lrelu}.
 *   fwd: y = clamp(act(sum_c img * w + bias) * gain)
 *   bwd: d1 = dy * d(clamp(act(.) * gain)) from the saved y;  partial[n][blk][co * Ci + c] = per-workgroup sums of d1 * img, then Co sums of
 *        d1 (the caller adds the sbg_fromrgb_bwd_blocks(N, HW) slabs in a fixed order);  dimg[n, c, p] = sum_co d1 * w (NULL = not wanted). */
int sbg_fromrgb_supported(int Ci, int Co, int act);
int sbg_fromrgb_bwd_blocks(int N, int64_t HW);
int sbg_fromrgb_fwd(const float* img, const float* w, const float* bias, void* y, int dtype, int N, int Ci, int Co, int64_t HW,
                    int act, float alpha, float gain, float clamp, sbg_stream_t stream);
int sbg_fromrgb_bwd(const float* img, const float* w, const void* dy, const void* y, float* dimg, float* partial, int dtype,
                    int N, int Ci, int Co, int64_t HW, int act, float alpha, float gain, float clamp, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ADA augmentation pipe, device ops (train_parts/augmentations.py:121-433).
 *
 * grid_sample: bilinear, zero padding, align_corners = False -- the one mode of the reference's
 * `grid_sample_gradfix.grid_sample(input, grid)` (stylegan2ada/torch_utils/ops/grid_sample_gradfix.py:24-27,45), whose
 * backward is `aten::grid_sampler_2d_backward(grad_output, input, grid, 0, 0, False)` (:63-64).
 * fp32 planar tensors with element strides.  Sampling positions come from `grid` (fp32 [N, OH, OW, 2] dense, normalised
 * coordinates) or, when `grid` is NULL, from `theta` (fp32 [N, 2, 3]): the positions `F.affine_grid(theta, [N, C, OH, OW],
 * align_corners=False)` would produce (augmentations.py:299) are generated inside the kernel.
 *   sbg_grid_sample2d:      y[n, c, oy, ox]  = sum over the 4 neighbours of x weighted bilinearly (x, y required)
 *   sbg_grid_sample2d_bwd:  dx += scatter of dy (dx must be zeroed by the caller; fp32 atomics); if dgrid != NULL also
 *                           dgrid[n, oy, ox, 2] (needs x).  dy shares y's strides, dx shares x's. */
typedef struct sbg_grid_sample_params {
    const void* x; const float* grid; const float* theta; const void* dy;
    void* y; void* dx; float* dgrid;
    int N, C, IH, IW, OH, OW;
    int64_t xs_n, xs_c, xs_h, xs_w;
    int64_t ys_n, ys_c, ys_h, ys_w;
    const float* theta_host;            /* optional HOST copy of theta: lets _bwd pick the deterministic gather kernel */
} sbg_grid_sample_params;
int sbg_grid_sample2d(const sbg_grid_sample_params* p, sbg_stream_t stream);
int sbg_grid_sample2d_bwd(const sbg_grid_sample_params* p, sbg_stream_t stream);
/* 1 when _bwd will take the gather kernel for these parameters: affine positions (theta + theta_host, no grid, no dgrid, C <= 4)
 * whose per-pixel candidate box is small.  That kernel OVERWRITES every element of dx (no zero fill needed, no atomics, bitwise
 * reproducible); otherwise _bwd accumulates into a caller-zeroed dx with fp32 atomics. */
int sbg_grid_sample2d_bwd_overwrites(const sbg_grid_sample_params* p);

/* Per-sample colour transform of dense planar fp32 RGB images x [N, 3, HW]: y[n, c, :] = sum_c' M[n, c, c'] x[n, c', :] + M[n, c, 3]
 * with M fp32 [N, 3, 4] -- `C[:, :3, :3] @ images + C[:, :3, 3:]` (augmentations.py:352-354).  Its data gradient is the same call
 * with the transposed 3x3 block and a zero fourth column. */
int sbg_color_transform(const float* x, const float* M, float* y, int N, int64_t HW, sbg_stream_t stream);

/* Per-sample 1-D correlation of M dense fp32 planes [M, H, W] along W (axis 0) or H (axis 1):
 *   y[m, .., o] = sum_t x[m, .., o + t - pad] * taps[m / planes_per_filter][flip ? T-1-t : t]      (zeros outside)
 * = one of the two grouped convolutions of the image-space filter, `conv2d(images, Hz_prime.unsqueeze(2 or 3),
 * groups=batch*channels)` (augmentations.py:388-389); with flip = 1 and pad = T - 1 - pad_fwd it is their data gradient.
 * Output extent along the axis: L + 2*pad - T + 1. */
int sbg_filter1d_batch(const float* x, const float* taps, float* y, int M, int H, int W, int T, int axis, int pad,
                       int planes_per_filter, int flip, sbg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * In-process launch timing (measurement only; bench.py's roofline figures come from here).
 * While enabled, every kernel launch of this library is bracketed by two hipEvents recorded on the launch stream
 * and logged with its algorithmic flops / bytes.  sbg_prof_fetch() synchronises the logged events, writes up to `max`
 * records (oldest first) with their measured duration, clears the log and returns the number written
 * (or the number pending when out == NULL). */
enum sbg_kernel_kind {
    SBG_K_BIAS_ACT = 1, SBG_K_UPFIRDN2D = 2, SBG_K_CONV_IGEMM = 3, SBG_K_CONV_WGRAD = 4, SBG_K_WGRAD_REDUCE = 5,
    SBG_K_SCALE_NC = 6, SBG_K_DOT_HW = 7, SBG_K_SN_POWER = 9, SBG_K_ATTENTION = 10, SBG_K_GRID_SAMPLE = 11, SBG_K_FILTER1D = 12, SBG_K_COLOR = 13, SBG_K_WEIGHT_PREP = 14, SBG_K_TORGB = 15, SBG_K_FROMRGB = 16,
    SBG_K_GROUPED_GEMM = 17
};
typedef struct sbg_prof_record {
    int    kind;            /* enum sbg_kernel_kind */
    int    dims[7];         /* kernel specific shape key (see each kernel's source) */
    double flops;           /* algorithmic floating point operations of the launch (2 x MACs) */
    double bytes;           /* algorithmic HBM bytes of the launch */
    float  ms;              /* measured duration */
    int    pad;
} sbg_prof_record;
/* Experiment word (diagnosis, not part of the contract): kernel variants under A/B test select on its bits; initial value from the
 * environment variable SBG_EXPERIMENT; returns the previous value.  0 = the shipped configuration. */
int sbg_experiment_set(int value);
int sbg_prof_enable(int on);
int sbg_prof_fetch(sbg_prof_record* out, int max);

#ifdef __cplusplus
}
#endif
#endif /* SBG_HIP_H */
