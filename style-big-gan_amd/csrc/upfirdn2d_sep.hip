// upfirdn2d_sep.hip -- separable upfirdn2d (one 1-D filter along both axes) in ONE launch.
//
// The reference runs a separable filter as two plugin calls, a row pass then a column pass with sqrt(gain) each
// (stylegan2ada/torch_utils/ops/upfirdn2d.py:236-240), materialising the intermediate image in HBM.  Its hot users are the
// augmentation pipe's 2x up- / down-sampling with the 12-tap sym6 low-pass on planar fp32 RGB batches of ~500x500..1000x1000
// pixels (train_parts/augmentations.py:292,303) and their gradients.  Here a workgroup stages the input window of a 32x64
// output tile in LDS, runs the row pass LDS -> LDS and the column pass LDS -> registers -> HBM: the image is read once and
// written once (HBM-bound; algorithmic bytes = (numel_in + numel_out) * 4).
//
// Planar dense fp32 [M, H, W] planes; up and down in {1, 2} (template parameters, so the polyphase tap strides are constants);
// same index convention as the generic kernel (upfirdn2d.hip): output o reads the zero-inserted input at u = o*down - pad0 + k
// with tap k = 0..T-1 in visiting order, sf[k] = f[flip ? k : T-1-k].
#include "sbg_common.h"

namespace {

constexpr int TOY = 32, TOX = 64;           // output tile
constexpr int WIN_H = 80, WIN_W = 160;      // input window capacity (covers down = 2 with up to 16 taps)
constexpr int MAX_T = 32;

struct SepArgs {
    const float* x; const float* f; float* y;
    int M, IH, IW, OH, OW, T, padx0, pady0, flip;
    float gain;     // total gain (sqrt per pass)
    int tiles_x, tiles_y;
};

static __device__ __forceinline__ int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static __device__ __forceinline__ int posmod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

template <int UP, int DOWN>
__global__ void __launch_bounds__(256) upfirdn2d_sep_kernel(SepArgs p)
{
    __shared__ float sf[MAX_T];
    __shared__ float win[WIN_H * WIN_W];
    __shared__ float tmp[WIN_H * TOX];
    const int tid = threadIdx.x;
    if (tid < p.T) sf[tid] = p.f[p.flip ? tid : p.T - 1 - tid] * sqrtf(p.gain);

    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y; const int m = b / p.tiles_y;
    const int ox0 = tx * TOX, oy0 = ty * TOY;
    const int nox = min(TOX, p.OW - ox0), noy = min(TOY, p.OH - oy0);

    // window in input pixels: u runs over [o0*DOWN - pad0, (o0 + n - 1)*DOWN - pad0 + T - 1], input index = u / UP where divisible
    const int ux0 = ox0 * DOWN - p.padx0, ux1 = (ox0 + nox - 1) * DOWN - p.padx0 + p.T - 1;
    const int uy0 = oy0 * DOWN - p.pady0, uy1 = (oy0 + noy - 1) * DOWN - p.pady0 + p.T - 1;
    const int ix0 = floordiv(ux0 + UP - 1, UP), ix1 = floordiv(ux1, UP);     // ceil(ux0 / UP) .. floor(ux1 / UP)
    const int iy0 = floordiv(uy0 + UP - 1, UP), iy1 = floordiv(uy1, UP);
    const int ww = ix1 - ix0 + 1, wh = iy1 - iy0 + 1;                        // <= WIN_W, WIN_H (checked by the launcher)

    const float* xp = p.x + (int64_t)m * p.IH * p.IW;
    for (int i = tid; i < wh * ww; i += 256) {
        const int r = i / ww, c = i - r * ww;
        const int iy = iy0 + r, ix = ix0 + c;
        win[r * WIN_W + c] = (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) ? xp[(int64_t)iy * p.IW + ix] : 0.0f;
    }
    __syncthreads();

    // row pass: tmp[r][ox] = sum_k sf[k] * Z[r][ux + k],  Z = zero-inserted window row
    for (int i = tid; i < wh * TOX; i += 256) {
        const int r = i / TOX, oxl = i - r * TOX;
        float acc = 0.0f;
        if (oxl < nox) {
            const int base = (ox0 + oxl) * DOWN - p.padx0;
            const float* row = win + r * WIN_W;
            for (int k = posmod(-base, UP); k < p.T; k += UP)
                acc += sf[k] * row[(base + k) / UP - ix0];          // exact division: base + k is a multiple of UP
        }
        tmp[r * TOX + oxl] = acc;
    }
    __syncthreads();

    // column pass
    float* yp = p.y + (int64_t)m * p.OH * p.OW;
    for (int i = tid; i < TOY * TOX; i += 256) {
        const int oyl = i / TOX, oxl = i - oyl * TOX;
        if (oyl >= noy || oxl >= nox) continue;
        const int base = (oy0 + oyl) * DOWN - p.pady0;
        float acc = 0.0f;
        for (int k = posmod(-base, UP); k < p.T; k += UP)
            acc += sf[k] * tmp[((base + k) / UP - iy0) * TOX + oxl];
        yp[(int64_t)(oy0 + oyl) * p.OW + ox0 + oxl] = acc;
    }
}

static bool window_fits(int up, int down, int T)
{
    // widest window: a full tile; + 2 for the floor / ceil slack
    const int ww = ((TOX - 1) * down + T - 1) / up + 2, wh = ((TOY - 1) * down + T - 1) / up + 2;
    return ww <= WIN_W && wh <= WIN_H && T <= MAX_T;
}

}  // namespace

extern "C" int sbg_upfirdn2d_separable_supported(int up, int down, int taps)
{
    return (up == 1 || up == 2) && (down == 1 || down == 2) && taps >= 1 && window_fits(up, down, taps) ? 1 : 0;
}

extern "C" int sbg_upfirdn2d_separable(const float* x, const float* f, float* y, int M, int IH, int IW, int OH, int OW, int taps,
                                       int up, int down, int padx0, int pady0, int flip, float gain, sbg_stream_t stream_)
{
    SBG_CHECK(x && f && y, "upfirdn2d_separable: null pointer");
    SBG_CHECK(M >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "upfirdn2d_separable: bad sizes");
    if (!sbg_upfirdn2d_separable_supported(up, down, taps))
        return sbg_fail(SBG_ERR_UNSUPPORTED, "upfirdn2d_separable: up / down must be 1 or 2 and the filter at most %d taps", MAX_T);
    SBG_CHECK((int64_t)M * IH * IW <= INT32_MAX && (int64_t)M * OH * OW <= INT32_MAX, "upfirdn2d_separable: tensors are limited to INT_MAX elements");
    if (M == 0) return 0;
    SepArgs a;
    a.x = x; a.f = f; a.y = y; a.M = M; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.T = taps;
    a.padx0 = padx0; a.pady0 = pady0; a.flip = flip; a.gain = gain;
    a.tiles_x = (OW + TOX - 1) / TOX; a.tiles_y = (OH + TOY - 1) / TOY;
    const int64_t blocks = (int64_t)M * a.tiles_x * a.tiles_y;
    SBG_CHECK(blocks <= INT32_MAX, "upfirdn2d_separable: too many tiles");
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_UPFIRDN2D, 0.0, 4.0 * ((double)M * IH * IW + (double)M * OH * OW), {M, 1, IH, IW, OH, OW, 100 + up * 10 + down});
    dim3 grid((unsigned)blocks), block(256);
    if (up == 1 && down == 1)      hipLaunchKernelGGL((upfirdn2d_sep_kernel<1, 1>), grid, block, 0, stream, a);
    else if (up == 2 && down == 1) hipLaunchKernelGGL((upfirdn2d_sep_kernel<2, 1>), grid, block, 0, stream, a);
    else if (up == 1 && down == 2) hipLaunchKernelGGL((upfirdn2d_sep_kernel<1, 2>), grid, block, 0, stream, a);
    else                           hipLaunchKernelGGL((upfirdn2d_sep_kernel<2, 2>), grid, block, 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
