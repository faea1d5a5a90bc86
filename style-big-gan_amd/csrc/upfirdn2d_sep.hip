// upfirdn2d_sep.hip -- separable upfirdn2d (one 1-D filter along both axes) in ONE launch.
//
// The reference runs a separable filter as two plugin calls, a row pass then a column pass with sqrt(gain) each
// (stylegan2ada/torch_utils/ops/upfirdn2d.py:236-240), materialising the intermediate image in HBM.  Its hot users are the
// augmentation pipe's 2x up- / down-sampling with the 12-tap sym6 low-pass on planar fp32 RGB batches of ~500x500..1000x1000
// pixels (train_parts/augmentations.py:292,303) and their gradients.  Here a workgroup stages the input window of a 32x64
// (16x64 when decimating) output tile in LDS, runs the row pass LDS -> LDS and the column pass LDS -> registers -> HBM: the image is read once and
// written once (HBM-bound; algorithmic bytes = (numel_in + numel_out) * 4).
//
// Planar dense fp32 [M, H, W] planes; up and down in {1, 2} (template parameters, so the polyphase tap strides are constants);
// same index convention as the generic kernel (upfirdn2d.hip): output o reads the zero-inserted input at u = o*down - pad0 + k
// with tap k = 0..T-1 in visiting order, sf[k] = f[flip ? k : T-1-k].
#include "sbg_common.h"

namespace {

constexpr int TOX = 64;                     // output tile width (one 256-B row of fp32 per wavefront store)
constexpr int MAX_T = 16;                   // taps (the window buffers below are sized for it)
template <int DOWN> struct Tile { static constexpr int TOY = DOWN == 2 ? 16 : 32; };
template <int UP, int DOWN> struct Win {    // input window of a full tile, +2 for the floor / ceil slack
    static constexpr int W = ((TOX - 1) * DOWN + MAX_T - 1) / UP + 2;
    static constexpr int H = ((Tile<DOWN>::TOY - 1) * DOWN + MAX_T - 1) / UP + 2;
};

struct SepArgs {
    const float* x; const float* f; float* y;
    int M, IH, IW, OH, OW, T, padx0, pady0, flip;
    float gain;     // total gain (sqrt per pass)
    int tiles_x, tiles_y;
};

static __device__ __forceinline__ int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static __device__ __forceinline__ int posmod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// LDS per workgroup: <2,1> 10 KB, <1,1> 34 KB, <2,2> 22 KB, <1,2> 39 KB -> 4..16 workgroups per CU
template <int UP, int DOWN>
__global__ void __launch_bounds__(256) upfirdn2d_sep_kernel(SepArgs p)
{
    constexpr int TOY = Tile<DOWN>::TOY, WW = Win<UP, DOWN>::W, WH = Win<UP, DOWN>::H;
    constexpr int NK = (MAX_T + UP - 1) / UP;       // taps one output touches (per phase)
    __shared__ float sf[MAX_T + UP];                // zero-extended, so the unrolled tap loops need no bound check
    __shared__ float win[WH * WW];
    __shared__ float tmp[WH * TOX];
    const int tid = threadIdx.x;
    if (tid < MAX_T + UP) sf[tid] = tid < p.T ? p.f[p.flip ? tid : p.T - 1 - tid] * sqrtf(p.gain) : 0.0f;

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
    const int ww = ix1 - ix0 + 1, wh = iy1 - iy0 + 1;                        // <= WW, WH by construction (T <= MAX_T)

    const float* xp = p.x + (int64_t)m * p.IH * p.IW;
    for (int i = tid; i < WH * WW; i += 256) {                               // the whole buffer: slack rows / columns read as zeros
        const int r = i / WW, c = i - r * WW;
        const int iy = iy0 + r, ix = ix0 + c;
        win[i] = (r < wh && c < ww && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) ? xp[(int64_t)iy * p.IW + ix] : 0.0f;
    }
    __syncthreads();

    // row pass: tmp[r][ox] = sum_k sf[k] * Z[r][ux + k],  Z = zero-inserted window row; lane = output column
    {
        const int oxl = tid & (TOX - 1);
        const int base = (ox0 + oxl) * DOWN - p.padx0;
        const int k0 = posmod(-base, UP);
        const int i0 = (base + k0) / UP - ix0;          // exact division: base + k0 is a multiple of UP; i0 + j < WW + slack
        float fk[NK];
#pragma unroll
        for (int j = 0; j < NK; j++) fk[j] = sf[k0 + j * UP];
        for (int r = tid / TOX; r < WH; r += 256 / TOX) {      // all rows: the column pass's zero-weight taps may touch the slack rows
            const float* row = win + r * WW + i0;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NK; j++) acc += fk[j] * row[min(j, WW - 1 - i0)];
            tmp[r * TOX + oxl] = acc;
        }
    }
    __syncthreads();

    // column pass
    float* yp = p.y + (int64_t)m * p.OH * p.OW;
    {
        const int oxl = tid & (TOX - 1);
        for (int oyl = tid / TOX; oyl < noy; oyl += 256 / TOX) {
            const int base = (oy0 + oyl) * DOWN - p.pady0;
            const int k0 = posmod(-base, UP);
            const int r0 = (base + k0) / UP - iy0;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NK; j++) acc += sf[k0 + j * UP] * tmp[min(r0 + j, WH - 1) * TOX + oxl];
            if (oxl < nox) yp[(int64_t)(oy0 + oyl) * p.OW + ox0 + oxl] = acc;
        }
    }
}

}  // namespace

extern "C" int sbg_upfirdn2d_separable_supported(int up, int down, int taps)
{
    return (up == 1 || up == 2) && (down == 1 || down == 2) && taps >= 1 && taps <= MAX_T ? 1 : 0;
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
    const int toy = down == 2 ? Tile<2>::TOY : Tile<1>::TOY;
    a.tiles_x = (OW + TOX - 1) / TOX; a.tiles_y = (OH + toy - 1) / toy;
    const int64_t blocks = (int64_t)M * a.tiles_x * a.tiles_y;
    SBG_CHECK(blocks <= INT32_MAX, "upfirdn2d_separable: too many tiles");
    hipStream_t stream = (hipStream_t)stream_;
    SbgProfScope prof(stream, SBG_K_UPFIRDN2D, 0.0, 4.0 * ((double)M * IH * IW + (double)M * OH * OW), {M, 1, IH, IW, OH, OW, 100 + up * 10 + down});
    dim3 grid((unsigned)blocks), block(256);
    if (up == 1 && down == 1)      SBG_LAUNCH((upfirdn2d_sep_kernel<1, 1>), grid, block, 0, stream, a);
    else if (up == 2 && down == 1) SBG_LAUNCH((upfirdn2d_sep_kernel<2, 1>), grid, block, 0, stream, a);
    else if (up == 1 && down == 2) SBG_LAUNCH((upfirdn2d_sep_kernel<1, 2>), grid, block, 0, stream, a);
    else                           SBG_LAUNCH((upfirdn2d_sep_kernel<2, 2>), grid, block, 0, stream, a);
    SBG_HIP_LAUNCH_CHECK();
    return 0;
}
