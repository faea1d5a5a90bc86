// conv_common.h -- types shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_k64.hip).
#pragma once
#include "sbg_common.h"

namespace sbgconv {

struct bf16_mfma { static constexpr int dtype = SBG_BF16; };
struct f16_mfma  { static constexpr int dtype = SBG_F16;  };

template <class MF> struct Mfma;
template <> struct Mfma<bf16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short cvt(float v) { return f32_to_bf16_bits(v); }
};
template <> struct Mfma<f16_mfma> {
    static __device__ __forceinline__ float4_t run(short8_t a, short8_t b, float4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short cvt(float v) { return f32_to_f16_bits(v); }
};

struct ConvArgs {
    const unsigned short* x; const unsigned short* w; void* y; const float* oscale;
    const float* bias; const float* noise; int64_t noise_sn; int act; float alpha, gain, clamp;
    int ydtype;
    int N, IH, IW, Cin, Cout, OH, OW;
    int64_t xs_n, xs_h, xs_w, ys_n, ys_h, ys_w, ws_slab, ws_co;
    int stride, ntaps;
    int tap_dy[SBG_MAX_TAPS], tap_dx[SBG_MAX_TAPS], tap_slab[SBG_MAX_TAPS];
    int accumulate;
    int P;            // N * OH * OW output pixels of this launch
    int ptiles, ctiles;
    // phases (conv_gather_ld_kernel): up to four output sub-grids of one launch (the s x s phases of a transposed convolution), each
    // with its own run of taps [ph_tap0, ph_tap0 + ph_ntaps), grid ph_OH x ph_OW, pixel count ph_P and y offset; nphase == 1: the whole launch
    int nphase, ph_rot_div;   // phase of tile r = (r % nphase + (r / nphase) / ph_rot_div) % nphase: a persistent workgroup cycles through the phases
    int ph_tap0[4], ph_ntaps[4], ph_OH[4], ph_OW[4], ph_P[4];
    int64_t ph_yoff[4];
    int ksplit;               // gather kernel: the K-steps are split over gridDim.y workgroups, each writing its own fp32 slab
    int64_t y_split_stride;   // elements between the slabs (0 when ksplit == 1)
    int lds_params;   // halo kernel: a loader wave stages each tile's epilogue parameters (noise / bias / demodulation coefficients) in LDS
    int debug;        // ablation switches, honoured only by -DSBG_K64_DEBUG builds (diagnosis; see conv_k64.hip)
};

} // namespace sbgconv

// conv_k64.hip: K-step-64 LDS-DMA kernels (gather and halo-staged).  Returns SBG_OK / an error, or -1 when the launch does not
// fit these kernels (the caller then uses the kernels of conv_igemm.hip).
int sbg_conv_k64_dispatch(sbgconv::ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, void* workspace, int ksplit, hipStream_t stream);
// conv_halo8.hip: the 3x3 / stride-1 halo-staged tile in an 8-wave structure with two accumulator sets (a finished tile drains behind the next
// one).  Returns SBG_OK / an error, or -1 when the launch stays with conv_k64.hip's conv_halo_ld_kernel.
int sbg_conv_halo8_dispatch(sbgconv::ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, hipStream_t stream);
// conv_thin.hip: few-channel convolutions (Cin, Cout <= 64, one of them <= 32) as a streaming kernel with the reduction axis packed
// over (tap, channel).  Returns SBG_OK / an error, or -1 when the launch is not a thin one.
int sbg_conv_thin_dispatch(sbgconv::ConvArgs& a, bool bf16, hipStream_t stream);
// conv_up2.hip: all four phases of a stride-2 3x3 transposed convolution from one staged input halo.  Returns SBG_OK / an error, or -1 when the
// launch is not of that form; on SBG_OK `border` describes the remaining last row / column rectangles as an ordinary phased launch.
int sbg_conv_up2_dispatch(sbgconv::ConvArgs& a, bool bf16, int64_t x_bytes, int64_t w_bytes, sbg_conv_params* border, const sbg_conv_params* q, hipStream_t stream);
