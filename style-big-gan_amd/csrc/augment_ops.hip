// augment_ops.hip -- device ops of the ADA augmentation pipe:
//   * sbg_grid_sample2d / sbg_grid_sample2d_bwd: bilinear sampling, zero padding, align_corners = False -- the only mode the
//     reference's grid_sample_gradfix supports (stylegan2ada/torch_utils/ops/grid_sample_gradfix.py:12-15,45,63-64) -- with the
//     sampling positions either read from a grid tensor or generated in the kernel from a per-sample 2x3 affine matrix
//     (= affine_grid + grid_sample of train_parts/augmentations.py:299-300 without materialising the [N,H,W,2] grid);
//   * sbg_filter1d_batch: per-sample 1-D correlation along W or H -- the two grouped convolutions of the image-space filter
//     (train_parts/augmentations.py:388-389), also their data gradient.
// All three are HBM-bound streaming kernels over planar fp32 images with 1 or 3 channels: one lane per output pixel, lanes of
// a wavefront along W (coalesced 256-B rows), channel loop inside the lane so index math and weights are computed once.
#include "sbg_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct GridArgs {
    const float* x; const float* grid; const float* theta; const float* dy;
    float* y; float* dx; float* dgrid;
    int N, C, IH, IW, OH, OW;
    int64_t xs_n, xs_c, xs_h, xs_w;      // input (and dx) strides, elements
    int64_t ys_n, ys_c, ys_h, ys_w;      // output (and dy) strides
};

// sampling position of output pixel (n, oy, ox) in input pixel units (align_corners = False)
static __device__ __forceinline__ void sample_pos(const GridArgs& p, int n, int oy, int ox, float& ix, float& iy)
{
    float gx, gy;
    if (p.grid) {
        const float* g = p.grid + (((int64_t)n * p.OH + oy) * p.OW + ox) * 2;
        gx = g[0]; gy = g[1];
    } else {
        // affine_grid: base coordinates (2j + 1) / W - 1, then [gx, gy] = theta[n] @ [bx, by, 1]
        const float* t = p.theta + (int64_t)n * 6;
        const float bx = (2.0f * ox + 1.0f) / (float)p.OW - 1.0f;
        const float by = (2.0f * oy + 1.0f) / (float)p.OH - 1.0f;
        gx = t[0] * bx + t[1] * by + t[2];
        gy = t[3] * bx + t[4] * by + t[5];
    }
    ix = ((gx + 1.0f) * (float)p.IW - 1.0f) * 0.5f;
    iy = ((gy + 1.0f) * (float)p.IH - 1.0f) * 0.5f;
}

__global__ void __launch_bounds__(256) grid_sample_fwd_kernel(GridArgs p)
{
    const int64_t total = (int64_t)p.N * p.OH * p.OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.OW);
        const int oy = (int)((i / p.OW) % p.OH);
        const int n  = (int)(i / ((int64_t)p.OW * p.OH));
        float ix, iy;
        sample_pos(p, n, oy, ox, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        const float tx = ix - fx, ty = iy - fy;
        // positions far outside (or NaN) contribute nothing; the clamp keeps the int conversion defined
        const bool sane = (ix > -2.0f) && (ix < (float)p.IW + 1.0f) && (iy > -2.0f) && (iy < (float)p.IH + 1.0f);
        const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
        const bool vx0 = x0 >= 0 && x0 < p.IW, vx1 = x0 + 1 >= 0 && x0 + 1 < p.IW;
        const bool vy0 = y0 >= 0 && y0 < p.IH, vy1 = y0 + 1 >= 0 && y0 + 1 < p.IH;
        const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
        const float* xb = p.x + (int64_t)n * p.xs_n + (int64_t)y0 * p.xs_h + (int64_t)x0 * p.xs_w;
        float* yb = p.y + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        for (int c = 0; c < p.C; c++) {
            const float* xc = xb + (int64_t)c * p.xs_c;
            float v = 0.0f;
            if (vy0 && vx0) v += w00 * xc[0];
            if (vy0 && vx1) v += w01 * xc[p.xs_w];
            if (vy1 && vx0) v += w10 * xc[p.xs_h];
            if (vy1 && vx1) v += w11 * xc[p.xs_h + p.xs_w];
            yb[(int64_t)c * p.ys_c] = v;
        }
    }
}

// dx += scatter of dy through the same bilinear weights (dx zeroed by the caller); optional dgrid[n,oy,ox,2].
__global__ void __launch_bounds__(256) grid_sample_bwd_kernel(GridArgs p)
{
    const int64_t total = (int64_t)p.N * p.OH * p.OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.OW);
        const int oy = (int)((i / p.OW) % p.OH);
        const int n  = (int)(i / ((int64_t)p.OW * p.OH));
        float ix, iy;
        sample_pos(p, n, oy, ox, ix, iy);
        const float fx = floorf(ix), fy = floorf(iy);
        const float tx = ix - fx, ty = iy - fy;
        const bool sane = (ix > -2.0f) && (ix < (float)p.IW + 1.0f) && (iy > -2.0f) && (iy < (float)p.IH + 1.0f);
        const int x0 = sane ? (int)fx : -4, y0 = sane ? (int)fy : -4;
        const bool vx0 = x0 >= 0 && x0 < p.IW, vx1 = x0 + 1 >= 0 && x0 + 1 < p.IW;
        const bool vy0 = y0 >= 0 && y0 < p.IH, vy1 = y0 + 1 >= 0 && y0 + 1 < p.IH;
        const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
        const int64_t off = (int64_t)n * p.xs_n + (int64_t)y0 * p.xs_h + (int64_t)x0 * p.xs_w;
        const float* dyb = p.dy + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        float gix = 0.0f, giy = 0.0f;
        for (int c = 0; c < p.C; c++) {
            const float g = dyb[(int64_t)c * p.ys_c];
            const int64_t oc = off + (int64_t)c * p.xs_c;
            if (p.dx) {
                if (vy0 && vx0) unsafeAtomicAdd(p.dx + oc, w00 * g);
                if (vy0 && vx1) unsafeAtomicAdd(p.dx + oc + p.xs_w, w01 * g);
                if (vy1 && vx0) unsafeAtomicAdd(p.dx + oc + p.xs_h, w10 * g);
                if (vy1 && vx1) unsafeAtomicAdd(p.dx + oc + p.xs_h + p.xs_w, w11 * g);
            }
            if (p.dgrid) {
                const float* xc = p.x + oc;
                const float v00 = (vy0 && vx0) ? xc[0] : 0.0f, v01 = (vy0 && vx1) ? xc[p.xs_w] : 0.0f;
                const float v10 = (vy1 && vx0) ? xc[p.xs_h] : 0.0f, v11 = (vy1 && vx1) ? xc[p.xs_h + p.xs_w] : 0.0f;
                gix += g * ((v01 - v00) * (1 - ty) + (v11 - v10) * ty);
                giy += g * ((v10 - v00) * (1 - tx) + (v11 - v01) * tx);
            }
        }
        if (p.dgrid) {
            float* dg = p.dgrid + (((int64_t)n * p.OH + oy) * p.OW + ox) * 2;
            dg[0] = gix * (float)p.IW * 0.5f;
            dg[1] = giy * (float)p.IH * 0.5f;
        }
    }
}


// Deterministic backward for affine sampling positions: one lane per INPUT pixel gathers from the output pixels whose bilinear
// footprint covers it.  Output pixel o samples at P(o) = A o + b (affine), so the candidates of input pixel i are the integer
// points of the parallelogram A^-1 ((i - b) + (-1, 1)^2); its bounding box is walked and every candidate re-evaluates P(o) with
// the forward kernel's own expression, so the weights agree with the forward pass to rounding.  No atomics, no zero fill, and
// neighbouring lanes read overlapping dy windows (L1 / L2 hits).  ex / ey: half extents of the bounding box, computed by the
// launcher from the host copy of theta (which also bounds the loop: the launcher refuses boxes larger than GATHER_MAX_BOX).
#define GATHER_MAX_BOX 24
#define GATHER_MAX_C 4
__global__ void __launch_bounds__(256) grid_sample_bwd_gather_kernel(GridArgs p)
{
    const int64_t total = (int64_t)p.N * p.IH * p.IW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % p.IW);
        const int iy = (int)((i / p.IW) % p.IH);
        const int n  = (int)(i / ((int64_t)p.IW * p.IH));
        const float* t = p.theta + (int64_t)n * 6;
        const float a00 = t[0] * (float)p.IW / (float)p.OW, a01 = t[1] * (float)p.IW / (float)p.OH;
        const float a10 = t[3] * (float)p.IH / (float)p.OW, a11 = t[4] * (float)p.IH / (float)p.OH;
        float b0, b1;
        sample_pos(p, n, 0, 0, b0, b1);
        const float det = a00 * a11 - a01 * a10;
        const float r = 1.0f / det;
        const float i00 = a11 * r, i01 = -a01 * r, i10 = -a10 * r, i11 = a00 * r;
        const float dx_ = (float)ix - b0, dy_ = (float)iy - b1;
        const float ocx = i00 * dx_ + i01 * dy_, ocy = i10 * dx_ + i11 * dy_;
        const float ex = fabsf(i00) + fabsf(i01) + 0.01f, ey = fabsf(i10) + fabsf(i11) + 0.01f;
        float acc[GATHER_MAX_C];
#pragma unroll
        for (int c = 0; c < GATHER_MAX_C; c++) acc[c] = 0.0f;
        // (NaN / inf from a singular theta fail every comparison below and leave empty loops)
        int ox0 = (ocx - ex > -1.0f) ? (int)floorf(fminf(ocx - ex, (float)p.OW)) : -1;
        int oy0 = (ocy - ey > -1.0f) ? (int)floorf(fminf(ocy - ey, (float)p.OH)) : -1;
        int ox1 = (ocx + ex < (float)p.OW) ? (int)ceilf(fmaxf(ocx + ex, -1.0f)) : p.OW;
        int oy1 = (ocy + ey < (float)p.OH) ? (int)ceilf(fmaxf(ocy + ey, -1.0f)) : p.OH;
        ox0 = max(ox0, 0); oy0 = max(oy0, 0); ox1 = min(ox1, p.OW - 1); oy1 = min(oy1, p.OH - 1);
        ox1 = min(ox1, ox0 + GATHER_MAX_BOX); oy1 = min(oy1, oy0 + GATHER_MAX_BOX);        // hard bound; the launcher guarantees it is never hit
        for (int oy = oy0; oy <= oy1; oy++) {
            for (int ox = ox0; ox <= ox1; ox++) {
                float px, py;
                sample_pos(p, n, oy, ox, px, py);
                const float wx = 1.0f - fabsf(px - (float)ix), wy = 1.0f - fabsf(py - (float)iy);
                if (wx > 0.0f && wy > 0.0f) {
                    const float w = wx * wy;
                    const float* d = p.dy + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
#pragma unroll
                    for (int c = 0; c < GATHER_MAX_C; c++)
                        if (c < p.C) acc[c] += w * d[(int64_t)c * p.ys_c];
                }
            }
        }
        float* o = p.dx + (int64_t)n * p.xs_n + (int64_t)iy * p.xs_h + (int64_t)ix * p.xs_w;
#pragma unroll
        for (int c = 0; c < GATHER_MAX_C; c++)
            if (c < p.C) o[(int64_t)c * p.xs_c] = acc[c];
    }
}

// largest candidate box (in output pixels, per axis) over the batch, from the host copy of theta; 0 = do not use the gather kernel
static int gather_box(const float* th, int N, int IH, int IW, int OH, int OW)
{
    float worst = 0.0f;
    for (int n = 0; n < N; n++) {
        const float* t = th + (size_t)n * 6;
        const float a00 = t[0] * IW / OW, a01 = t[1] * IW / OH, a10 = t[3] * IH / OW, a11 = t[4] * IH / OH;
        const float det = a00 * a11 - a01 * a10;
        if (!(fabsf(det) > 1e-12f)) return 0;
        const float ex = (fabsf(a11) + fabsf(a01)) / fabsf(det), ey = (fabsf(a10) + fabsf(a00)) / fabsf(det);
        if (!(ex < 1e6f && ey < 1e6f)) return 0;
        worst = fmaxf(worst, fmaxf(ex, ey));
    }
    const int box = 2 * (int)ceilf(worst + 0.01f) + 2;
    return box <= GATHER_MAX_BOX ? box : 0;
}

// y[n, c, i] = sum_c' M[n, c, c'] * x[n, c', i] + M[n, c, 3]: the per-sample 3x4 colour transform (augmentations.py:352-354) as
// one streaming pass (the reference's batched [3x3] @ [3, HW] matmul maps badly onto GEMM tiles), 4 pixels per lane.
__global__ void __launch_bounds__(256) color_transform_kernel(const float* __restrict__ x, const float* __restrict__ M, float* __restrict__ y,
                                                              int N, int64_t HW, int64_t quads_per_image)
{
    const int64_t total = (int64_t)N * quads_per_image;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / quads_per_image);
        const int64_t q = (i - (int64_t)n * quads_per_image) * 4;
        const float* m = M + (int64_t)n * 12;
        const float* xb = x + (int64_t)n * 3 * HW + q;
        float* yb = y + (int64_t)n * 3 * HW + q;
        if (q + 4 <= HW && ((HW & 3) == 0)) {
            const float4_t r = *reinterpret_cast<const float4_t*>(xb), g = *reinterpret_cast<const float4_t*>(xb + HW),
                           b = *reinterpret_cast<const float4_t*>(xb + 2 * HW);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float4_t o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = m[c * 4 + 0] * r[j] + m[c * 4 + 1] * g[j] + m[c * 4 + 2] * b[j] + m[c * 4 + 3];
                *reinterpret_cast<float4_t*>(yb + c * HW) = o;
            }
        } else {
            for (int j = 0; j < 4 && q + j < HW; j++) {
                const float r = xb[j], g = xb[HW + j], b = xb[2 * HW + j];
                for (int c = 0; c < 3; c++) yb[c * HW + j] = m[c * 4 + 0] * r + m[c * 4 + 1] * g + m[c * 4 + 2] * b + m[c * 4 + 3];
            }
        }
    }
}

struct FiltArgs {
    const float* x; const float* taps; float* y;
    int M, H, W, OH, OW, T, axis, pad, planes_per_filter, flip;
};

// y[m, oy, ox] = sum_t x[m, oy, ox + t - pad] * taps[m / ppf][t]   (axis 0: along W; axis 1: along H), zeros outside.
__global__ void __launch_bounds__(256) filter1d_batch_kernel(FiltArgs p)
{
    const int64_t total = (int64_t)p.M * p.OH * p.OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.OW);
        const int oy = (int)((i / p.OW) % p.OH);
        const int m  = (int)(i / ((int64_t)p.OW * p.OH));
        const float* w = p.taps + (int64_t)(m / p.planes_per_filter) * p.T;
        const float* xb = p.x + (int64_t)m * p.H * p.W;
        float acc = 0.0f;
        if (p.axis == 0) {
            const float* row = xb + (int64_t)oy * p.W;
            for (int t = 0; t < p.T; t++) {
                const int xx = ox + t - p.pad;
                const float wt = w[p.flip ? p.T - 1 - t : t];
                if (xx >= 0 && xx < p.W) acc += row[xx] * wt;
            }
        } else {
            for (int t = 0; t < p.T; t++) {
                const int yy = oy + t - p.pad;
                const float wt = w[p.flip ? p.T - 1 - t : t];
                if (yy >= 0 && yy < p.H) acc += xb[(int64_t)yy * p.W + ox] * wt;
            }
        }
        p.y[i] = acc;
    }
}

static int fill_grid_args(GridArgs& a, const sbg_grid_sample_params* p)
{
    SBG_CHECK(p != nullptr, "grid_sample: null params");
    SBG_CHECK((p->grid != nullptr) != (p->theta != nullptr), "grid_sample: exactly one of grid / theta must be given");
    SBG_CHECK(p->N >= 0 && p->C >= 1 && p->IH >= 1 && p->IW >= 1 && p->OH >= 1 && p->OW >= 1, "grid_sample: bad sizes");
    SBG_CHECK((int64_t)p->N * p->C * p->IH * p->IW <= INT32_MAX && (int64_t)p->N * p->C * p->OH * p->OW <= INT32_MAX,
              "grid_sample: tensors are limited to INT_MAX elements");
    a.x = (const float*)p->x; a.grid = p->grid; a.theta = p->theta; a.dy = (const float*)p->dy;
    a.y = (float*)p->y; a.dx = (float*)p->dx; a.dgrid = p->dgrid;
    a.N = p->N; a.C = p->C; a.IH = p->IH; a.IW = p->IW; a.OH = p->OH; a.OW = p->OW;
    a.xs_n = p->xs_n; a.xs_c = p->xs_c; a.xs_h = p->xs_h; a.xs_w = p->xs_w;
    a.ys_n = p->ys_n; a.ys_c = p->ys_c; a.ys_h = p->ys_h; a.ys_w = p->ys_w;
    return 0;
}

}  // namespace

extern "C" int sbg_grid_sample2d(const sbg_grid_sample_params* p, sbg_stream_t stream_)
{
    GridArgs a;
    if (int rc = fill_grid_args(a, p)) return rc;
    SBG_CHECK(a.x && a.y, "grid_sample: x and y are required");
    if (a.N == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t total = (int64_t)a.N * a.OH * a.OW;
    const double bytes = 4.0 * ((double)a.N * a.C * a.IH * a.IW + (double)total * a.C + (a.grid ? 2.0 * total : 0.0));
    SbgProfScope prof(stream, SBG_K_GRID_SAMPLE, 0.0, bytes, {a.N, a.C, a.IH, a.IW, a.OH, a.OW, 0});
    SBG_LAUNCH(grid_sample_fwd_kernel, dim3(sbg_stream_grid(total, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_grid_sample2d_bwd_overwrites(const sbg_grid_sample_params* p)
{
    if (!p || !p->theta || !p->theta_host || p->grid || p->dgrid || !p->dx || p->C > GATHER_MAX_C || p->N <= 0) return 0;
    return gather_box(p->theta_host, p->N, p->IH, p->IW, p->OH, p->OW) > 0 ? 1 : 0;
}

extern "C" int sbg_grid_sample2d_bwd(const sbg_grid_sample_params* p, sbg_stream_t stream_)
{
    GridArgs a;
    if (int rc = fill_grid_args(a, p)) return rc;
    SBG_CHECK(a.dy && (a.dx || a.dgrid), "grid_sample_bwd: dy and at least one of dx / dgrid are required");
    SBG_CHECK(!a.dgrid || a.x, "grid_sample_bwd: dgrid needs x");
    if (a.N == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t total = (int64_t)a.N * a.OH * a.OW;
    const double bytes = 4.0 * ((double)total * a.C + (a.dx ? 2.0 : 1.0) * (double)a.N * a.C * a.IH * a.IW + (a.grid ? 2.0 * total : 0.0));
    if (sbg_grid_sample2d_bwd_overwrites(p)) {
        SbgProfScope prof(stream, SBG_K_GRID_SAMPLE, 0.0, bytes, {a.N, a.C, a.IH, a.IW, a.OH, a.OW, 2});
        const int64_t in_total = (int64_t)a.N * a.IH * a.IW;
        SBG_LAUNCH(grid_sample_bwd_gather_kernel, dim3((unsigned)((in_total + 255) / 256)), dim3(256), 0, stream, a);
        SBG_HIP_LAUNCH_CHECK();
        return 0;
    }
    SbgProfScope prof(stream, SBG_K_GRID_SAMPLE, 0.0, bytes, {a.N, a.C, a.IH, a.IW, a.OH, a.OW, 1});
    SBG_LAUNCH(grid_sample_bwd_kernel, dim3(sbg_stream_grid(total, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_filter1d_batch(const float* x, const float* taps, float* y, int M, int H, int W, int T, int axis, int pad,
                                  int planes_per_filter, int flip, sbg_stream_t stream_)
{
    SBG_CHECK(x && taps && y, "filter1d_batch: null pointer");
    SBG_CHECK(M >= 0 && H >= 1 && W >= 1 && T >= 1 && pad >= 0 && planes_per_filter >= 1 && (axis == 0 || axis == 1), "filter1d_batch: bad sizes");
    FiltArgs a;
    a.x = x; a.taps = taps; a.y = y; a.M = M; a.H = H; a.W = W; a.T = T; a.axis = axis; a.pad = pad;
    a.planes_per_filter = planes_per_filter; a.flip = flip;
    a.OH = axis == 1 ? H + 2 * pad - T + 1 : H;
    a.OW = axis == 0 ? W + 2 * pad - T + 1 : W;
    SBG_CHECK(a.OH >= 1 && a.OW >= 1, "filter1d_batch: filter longer than the padded image");
    SBG_CHECK((int64_t)M * H * W <= INT32_MAX && (int64_t)M * a.OH * a.OW <= INT32_MAX, "filter1d_batch: tensors are limited to INT_MAX elements");
    if (M == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t total = (int64_t)M * a.OH * a.OW;
    SbgProfScope prof(stream, SBG_K_FILTER1D, 0.0, 4.0 * ((double)M * H * W + (double)total), {M, H, W, T, axis, pad, 0});
    SBG_LAUNCH(filter1d_batch_kernel, dim3(sbg_stream_grid(total, 256)), dim3(256), 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int sbg_color_transform(const float* x, const float* M, float* y, int N, int64_t HW, sbg_stream_t stream_)
{
    SBG_CHECK(x && M && y, "color_transform: null pointer");
    SBG_CHECK(N >= 0 && HW >= 1 && (int64_t)N * 3 * HW <= INT32_MAX, "color_transform: bad sizes");
    if (N == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t quads = (HW + 3) / 4;
    SbgProfScope prof(stream, SBG_K_COLOR, 0.0, 4.0 * 6.0 * (double)N * HW, {N, 3, (int)HW, 0, 0, 0, 0});
    SBG_LAUNCH(color_transform_kernel, dim3(sbg_stream_grid((int64_t)N * quads, 256)), dim3(256), 0, stream, x, M, y, N, HW, quads);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
