// conv_thin.hip -- convolutions with FEW channels on both sides (Cin, Cout <= 64, one of them <= 32): the 16 / 32 / 64-channel layers of the
// high-resolution blocks (configs/ffhq_sg2.yaml at 1024x1024: channel_base 16384 -> 16 channels at 1024^2, 32 at 512^2), their strided /
// transposed / 1x1 relatives and the 3-channel data gradient of fromRGB.
//
// The K-step-64 kernels of conv_k64.hip tile 64..128 output channels x 64 input channels PER TAP: at 16 -> 16 channels fifteen sixteenths of
// their MFMAs multiply padding (measured 24 TFLOP/s, 330 GB/s: 6.5 ms for a launch whose tensors stream in 0.43 ms).  Here the reduction
// axis is the packed pair (tap, channel), k = tap * Cin + ci, so a 3x3 layer with 16 channels is K = 144 -> five 32-deep MFMA steps, the
// output tile is Cout x 256 pixels, and the kernel is what the layer really is -- a streaming pass bound by HBM:
//   * weights (<= 64 x 576 values) sit in LDS for the whole workgroup, rows padded so that the 16 rows of an A fragment start in
//     16 different bank groups;
//   * the B fragment of lane (pixel fi, k-group fg) is eight consecutive channels of ONE tap of ONE input pixel = one 16-B global load
//     (the nine taps of neighbouring pixels re-read the same lines, which L1 / L2 absorb; HBM sees every line once);
//   * the next step's loads are issued before the current step's MFMAs; occupancy (8-16 waves per CU) hides the rest;
//   * the fused epilogue (demodulation, noise, bias, activation, gain, clamp) and the phase table of transposed convolutions are the
//     ones of sbg_conv_params.
// Same tap-list contract as the other convolution kernels (include/sbg_hip.h), so forward, data gradient, strided and transposed
// (phased) launches all take it.
#include "conv_common.h"

using namespace sbgconv;

namespace {

template <class MF, int TC, int TP>  // TC = 16-channel output fragments (Cout <= 16 * TC); TP = 16-pixel fragments per wave (a wave owns 16 * TP consecutive pixels)
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvArgs p, int pitch)       // pitch = LDS row pitch of the weights, in 16-bit elements
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_dy[SBG_MAX_TAPS], s_dx[SBG_MAX_TAPS], s_slab[SBG_MAX_TAPS];
    unsigned short* lw = reinterpret_cast<unsigned short*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = blockIdx.y;
    const bool phased = p.nphase > 1;
    const int tap0 = phased ? p.ph_tap0[ph] : 0, ntaps = phased ? p.ph_ntaps[ph] : p.ntaps;
    const int OH = phased ? p.ph_OH[ph] : p.OH, OW = phased ? p.ph_OW[ph] : p.OW;
    const int P = phased ? p.ph_P[ph] : p.P;
    const int64_t yoff = phased ? p.ph_yoff[ph] : 0;
    // output tile of a workgroup: (4 * TP) rows x 16 columns of ONE image -- a 2-D tile, so that the three rows of taps a wave reads are the
    // rows its neighbours in the workgroup read too (L1 hits; a 256 x 1 strip fetched every input row through three different workgroups)
    constexpr int TROWS = 4 * TP;
    const int tiles_x = (OW + 15) >> 4, tiles_y = (OH + TROWS - 1) / TROWS;
    int b_ = blockIdx.x;
    const int tx = b_ % tiles_x; b_ /= tiles_x;
    const int ty = b_ % tiles_y; const int n_img = b_ / tiles_y;
    if (n_img >= p.N) return;                               // (uniform per workgroup, before any barrier)
    (void)P;
    const int K = ntaps * p.Cin, nsteps = (K + 31) >> 5;

    // (tap tables first: per-lane indexing into the kernel-argument arrays would send them through scratch memory)
    if (tid < ntaps) { s_dy[tid] = p.tap_dy[tap0 + tid]; s_dx[tid] = p.tap_dx[tap0 + tid]; s_slab[tid] = p.tap_slab[tap0 + tid]; }
    __syncthreads();
    // ---- weights -> LDS: row co holds k = 0 .. 32 * nsteps - 1 (zeros beyond K and for co >= Cout), eight k per 16-B cell
    {
        const int cells = nsteps * 4;                       // 16-B cells per row
        for (int c = tid; c < 16 * TC * cells; c += 256) {
            const int co = c / cells, k = (c - co * cells) * 8;
            short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (co < p.Cout && k < K) {
                const int t = k / p.Cin, ci = k - t * p.Cin;
                v = *reinterpret_cast<const short8_t*>(p.w + (int64_t)s_slab[t] * p.ws_slab + (int64_t)co * p.ws_co + ci);
            }
            *reinterpret_cast<short8_t*>(lw + co * pitch + k) = v;
        }
    }
    __syncthreads();

    const int fi = lane & 15, fg = lane >> 4;
    const int cin8 = p.Cin >> 3;
    // ---- this lane's pixels: fragment j = row ty * TROWS + wave * TP + j, column tx * 16 + fi
    int iy0[TP], ix0[TP];
    const int64_t xbase = (int64_t)n_img * p.xs_n;
    const int ox = tx * 16 + fi;
#pragma unroll
    for (int j = 0; j < TP; j++) {
        const int oy = ty * TROWS + wave * TP + j;
        const bool ok = oy < OH && ox < OW;
        iy0[j] = ok ? oy * p.stride : -(1 << 28);           // invalid pixels fail every range test
        ix0[j] = ox * p.stride;
    }
    float4_t acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int j = 0; j < TP; j++) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    auto load_b = [&](int ks, short8_t (&fb)[TP]) {
        const int q = 4 * ks + fg;                          // 8-channel group index along k
        const int t = q / cin8, c0 = (q - t * cin8) * 8;
        const bool kok = q * 8 < K;
        const int dy = kok ? s_dy[t] : 0, dx = kok ? s_dx[t] : 0;
#pragma unroll
        for (int j = 0; j < TP; j++) {
            const int iy = iy0[j] + dy, ix = ix0[j] + dx;
            short8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
                v = *reinterpret_cast<const short8_t*>(p.x + xbase + (int64_t)iy * p.xs_h + (int64_t)ix * p.xs_w + c0);
            fb[j] = v;
        }
    };
    auto step = [&](int ks, const short8_t (&fb)[TP]) {
        short8_t fa[TC];
#pragma unroll
        for (int i = 0; i < TC; i++) fa[i] = *reinterpret_cast<const short8_t*>(lw + (16 * i + fi) * pitch + 32 * ks + 8 * fg);
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int j = 0; j < TP; j++) acc[i][j] = Mfma<MF>::run(fa[i], fb[j], acc[i][j]);
    };
    // two fragment buffers with FIXED names (a buffer chosen by a run-time index would live in scratch memory): the loads of the next step
    // are issued before the MFMAs of the current one
    short8_t fb0[TP], fb1[TP];
    load_b(0, fb0);
    for (int ks = 0; ks < nsteps; ks += 2) {
        if (ks + 1 < nsteps) load_b(ks + 1, fb1);
        step(ks, fb0);
        if (ks + 1 < nsteps) {
            if (ks + 2 < nsteps) load_b(ks + 2, fb0);
            step(ks + 1, fb1);
        }
    }

    // ---- epilogue: lane holds output channels 16 i + 4 fg + {0..3} of pixel (fragment j, column fi)
    const bool plain = (p.act <= SBG_ACT_LINEAR) && p.gain == 1.f && p.clamp < 0.f && !p.bias && !p.noise && !p.oscale;
    const float alpha = (p.act == SBG_ACT_LRELU) ? p.alpha : (p.act == SBG_ACT_RELU ? 0.f : 1.f);
    const float cl = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    const bool vec_ok = ((p.Cout & 3) == 0) && (((p.ys_n | p.ys_h | p.ys_w | yoff) & 3) == 0) && ((((uintptr_t)p.y) & 15) == 0);
#pragma unroll
    for (int j = 0; j < TP; j++) {
        const int oy = ty * TROWS + wave * TP + j, n = n_img;
        if (oy >= OH || ox >= OW) continue;
        const int64_t ybase = yoff + (int64_t)n * p.ys_n + (int64_t)oy * p.ys_h + (int64_t)ox * p.ys_w;
        const float nz = (!plain && p.noise) ? p.noise[(int64_t)n * p.noise_sn + (int64_t)oy * OW + ox] : 0.f;
#pragma unroll
        for (int i = 0; i < TC; i++) {
            const int co = 16 * i + 4 * fg;
            if (co >= p.Cout) continue;
            float4_t v = acc[i][j];
            if (!plain) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (co + e >= p.Cout) continue;
                    float u = v[e];
                    if (p.oscale) u *= p.oscale[(int64_t)n * p.Cout + co + e];
                    u += nz + (p.bias ? p.bias[co + e] : 0.f);
                    u = (u > 0.f) ? u : u * alpha;
                    v[e] = __builtin_amdgcn_fmed3f(u * p.gain, -cl, cl);
                }
            }
            if (p.ydtype == SBG_F32) {
                float* dst = (float*)p.y + ybase + co;
                if (vec_ok) {
                    if (p.accumulate) v += *reinterpret_cast<const float4_t*>(dst);
                    *reinterpret_cast<float4_t*>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (co + e < p.Cout) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
                }
            } else {
                unsigned short* dst = (unsigned short*)p.y + ybase + co;
                if (vec_ok) {
                    short4_t o = {(short)Mfma<MF>::cvt(v[0]), (short)Mfma<MF>::cvt(v[1]), (short)Mfma<MF>::cvt(v[2]), (short)Mfma<MF>::cvt(v[3])};
                    *reinterpret_cast<short4_t*>(dst) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (co + e < p.Cout) dst[e] = Mfma<MF>::cvt(v[e]);
                }
            }
        }
    }
}

template <class MF, int TC, int TP>
static int launch_thin(const ConvArgs& a, int maxP, int pitch, int lds, hipStream_t stream)
{
    auto kern = conv_thin_kernel<MF, TC, TP>;
    if (lds > 64 * 1024 && !SBG_RAISE_LDS_ONCE(kern, lds))
        return sbg_fail(SBG_ERR_LAUNCH, "conv2d_igemm: cannot raise the dynamic LDS limit to %d bytes", lds);
    int maxOH = a.OH, maxOW = a.OW;
    if (a.nphase > 1) { maxOH = maxOW = 0; for (int i = 0; i < a.nphase; i++) { if (a.ph_OH[i] > maxOH) maxOH = a.ph_OH[i]; if (a.ph_OW[i] > maxOW) maxOW = a.ph_OW[i]; } }
    const int64_t nblk = (int64_t)a.N * ((maxOH + 4 * TP - 1) / (4 * TP)) * ((maxOW + 15) / 16);     // a phase with a smaller grid leaves its surplus workgroups idle
    if (nblk > INT32_MAX || nblk < 1) return sbg_fail(SBG_ERR_INVALID, "conv2d_igemm: grid too large");
    const dim3 grid((unsigned)nblk, (unsigned)(a.nphase > 1 ? a.nphase : 1));
    SBG_LAUNCH(kern, grid, dim3(256), lds, stream, a, pitch);
    SBG_HIP_LAUNCH_CHECK();
    return SBG_OK;
}

} // namespace

// Returns SBG_OK / an error, or -1 when the launch is not a thin one (the caller then uses the wide kernels).
int sbg_conv_thin_dispatch(ConvArgs& a, bool bf16, hipStream_t stream)
{
    static const char* off = sbg_env("SBG_CONV_NO_THIN");
    if (off) return -1;
    if ((a.Cin & 7) || a.Cin > 64 || a.Cout > 64 || a.Cout < 1) return -1;
    if (!((a.Cin <= 32) || (a.Cout <= 32))) return -1;                 // 64 x 64 stays with the wide kernels
    if (a.ksplit > 1 || a.ntaps < 1) return -1;
    if (a.xs_n < 0 || a.xs_h < 0 || a.xs_w < 0 || a.ws_slab < 0 || a.ws_co < 0) return -1;
    if ((a.xs_n & 7) || (a.xs_h & 7) || (a.xs_w & 7) || (a.ws_slab & 7) || (a.ws_co & 7)) return -1;      // 16-B loads
    if ((((uintptr_t)a.x) & 15) || (((uintptr_t)a.w) & 15)) return -1;
    int maxtaps = a.ntaps, maxP = a.P;
    if (a.nphase > 1) {
        maxtaps = 0; maxP = 0;
        for (int i = 0; i < a.nphase; i++) { if (a.ph_ntaps[i] > maxtaps) maxtaps = a.ph_ntaps[i]; if (a.ph_P[i] > maxP) maxP = a.ph_P[i]; }
    }
    if (maxP <= 0) return -1;
    const int kpad = ((maxtaps * a.Cin + 31) >> 5) << 5;
    if (kpad > 640) return -1;
    const int pitch = (((kpad >> 1) | 4)) << 1;                          // row pitch: (kpad / 2) dwords with bit 2 set -> 16 rows on 16 bank groups
    const int tc = (a.Cout + 15) >> 4;
    const int TCs = tc <= 1 ? 1 : (tc <= 2 ? 2 : 4);
    const int lds = 16 * TCs * pitch * 2;
    const double ys = a.ydtype == SBG_F32 ? 4.0 : 2.0;
    double macs = 0.0, outpix = 0.0;
    if (a.nphase > 1) { for (int i = 0; i < a.nphase; i++) { macs += (double)a.ph_P[i] * a.ph_ntaps[i]; outpix += a.ph_P[i]; } }
    else { macs = (double)a.P * a.ntaps; outpix = a.P; }
    SbgProfScope prof(stream, SBG_K_CONV_IGEMM, 2.0 * macs * a.Cout * (double)a.Cin,
                      2.0 * a.N * a.IH * a.IW * (double)a.Cin + 2.0 * a.ntaps * a.Cout * (double)a.Cin + ys * outpix * (double)a.Cout * (a.accumulate ? 2 : 1),
                      {(int)outpix, a.Cout, a.Cin, a.ntaps, a.stride, a.OH, 6000000 + TCs * 16});
    // one or two output fragments: eight pixel fragments per wave (the A fragment read, the tap decode and the bounds math are shared by twice
    // the pixels; measured: see DESIGN.md); four output fragments keep four (accumulator registers)
    static const char* tp4 = sbg_env("SBG_THIN_TP4");
    const bool wide = TCs <= 2 && tp4 && atoi(tp4) == 0 && maxP >= (1 << 16);       // measured no better than four rows per wave: off unless SBG_THIN_TP4=0
    if (bf16) {
        if (TCs == 1) return wide ? launch_thin<bf16_mfma, 1, 8>(a, maxP, pitch, lds, stream) : launch_thin<bf16_mfma, 1, 4>(a, maxP, pitch, lds, stream);
        if (TCs == 2) return wide ? launch_thin<bf16_mfma, 2, 8>(a, maxP, pitch, lds, stream) : launch_thin<bf16_mfma, 2, 4>(a, maxP, pitch, lds, stream);
        return launch_thin<bf16_mfma, 4, 4>(a, maxP, pitch, lds, stream);
    }
    if (TCs == 1) return wide ? launch_thin<f16_mfma, 1, 8>(a, maxP, pitch, lds, stream) : launch_thin<f16_mfma, 1, 4>(a, maxP, pitch, lds, stream);
    if (TCs == 2) return wide ? launch_thin<f16_mfma, 2, 8>(a, maxP, pitch, lds, stream) : launch_thin<f16_mfma, 2, 4>(a, maxP, pitch, lds, stream);
    return launch_thin<f16_mfma, 4, 4>(a, maxP, pitch, lds, stream);
}
