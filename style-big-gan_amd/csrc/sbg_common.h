// sbg_common.h -- shared helpers for the gfx950 kernels of libsbg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>
#include <mutex>
#include "../../include/sbg_hip.h"

// ------------------------------------------------------------------------------------------------
// Error reporting: thread-local message, integer status codes (no exceptions cross the C ABI).

std::string& sbg_err_slot();

static inline int sbg_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    sbg_err_slot() = buf;
    return code;
}

#define SBG_CHECK(cond, ...) do { if (!(cond)) return sbg_fail(SBG_ERR_INVALID, __VA_ARGS__); } while (0)
#define SBG_HIP_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return sbg_fail(SBG_ERR_LAUNCH, "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

// Launch geometry is validated on the host BEFORE anything is enqueued: a dispatch packet with a zero grid dimension, more than 1024
// work-items per workgroup or more LDS than a CU owns must never reach the queue (the runtime's own checks run after the fact, and
// a tool that intercepts the queue -- rocprofv3 counter collection -- sees the packet as it was built).
bool sbg_launch_geometry_ok(dim3 grid, dim3 block, size_t lds_bytes, const char* kernel, const char* file, int line);

#define SBG_LAUNCH_OR(on_error, kern, grid, block, lds, stream, ...) do { \
    const dim3 sbg_g_ = (grid), sbg_b_ = (block); \
    if (!sbg_launch_geometry_ok(sbg_g_, sbg_b_, (size_t)(lds), #kern, __FILE__, __LINE__)) { on_error; } \
    hipLaunchKernelGGL(kern, sbg_g_, sbg_b_, (lds), (stream), __VA_ARGS__); } while (0)
#define SBG_LAUNCH(kern, grid, block, lds, stream, ...) SBG_LAUNCH_OR(return SBG_ERR_INVALID, kern, grid, block, lds, stream, __VA_ARGS__)

// One-time, thread-safe raise of a kernel's dynamic-LDS limit (forward runs on the caller's thread, backward on autograd's workers).
// Evaluates to true when the limit is in place.
#define SBG_RAISE_LDS_ONCE(kern, bytes) ([&]() -> bool { static std::once_flag once_; static bool ok_ = false; \
    std::call_once(once_, [&] { ok_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) == hipSuccess; }); \
    return ok_; }())

// Experiment switches (SBG_* environment variables) are read once per process, never on the launch path.
const char* sbg_env(const char* name);      // cached getenv: the first call per name scans `environ`, later calls are a table lookup
int sbg_experiment();                       // experiment word (SBG_EXPERIMENT / sbg_experiment_set): kernel variants under A/B test select on its bits

// ------------------------------------------------------------------------------------------------
// 16-bit float storage <-> fp32 math.

typedef __attribute__((ext_vector_type(8))) short  short8_t;
typedef __attribute__((ext_vector_type(4))) short  short4_t;
typedef __attribute__((ext_vector_type(4))) float  float4_t;
typedef __attribute__((ext_vector_type(4))) int    int4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

struct bf16_s { unsigned short v; };   // storage-only tags used for template dispatch
struct f16_s  { unsigned short v; };

template <class T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void  st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_s> {
    static __device__ __forceinline__ float ld(const bf16_s* p) { return __uint_as_float(((unsigned)p->v) << 16); }
    static __device__ __forceinline__ void  st(bf16_s* p, float v) { __bf16 h = (__bf16)v; p->v = __builtin_bit_cast(unsigned short, h); }
};
template <> struct Elem<f16_s> {
    static __device__ __forceinline__ float ld(const f16_s* p) { _Float16 h = __builtin_bit_cast(_Float16, p->v); return (float)h; }
    static __device__ __forceinline__ void  st(f16_s* p, float v) { _Float16 h = (_Float16)v; p->v = __builtin_bit_cast(unsigned short, h); }
};

static __device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
static __device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) { __bf16 h = (__bf16)v; return __builtin_bit_cast(unsigned short, h); }
static __device__ __forceinline__ float f16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
static __device__ __forceinline__ unsigned short f32_to_f16_bits(float v) { _Float16 h = (_Float16)v; return __builtin_bit_cast(unsigned short, h); }

// Vector of 8 elements of T <-> 8 floats (16 B for 16-bit types, 32 B for fp32).
template <class T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[8]) {
        float4_t a = *reinterpret_cast<const float4_t*>(p), b = *reinterpret_cast<const float4_t*>(p + 4);
        o[0]=a[0];o[1]=a[1];o[2]=a[2];o[3]=a[3];o[4]=b[0];o[5]=b[1];o[6]=b[2];o[7]=b[3];
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
        float4_t a = {v[0],v[1],v[2],v[3]}, b = {v[4],v[5],v[6],v[7]};
        *reinterpret_cast<float4_t*>(p) = a; *reinterpret_cast<float4_t*>(p + 4) = b;
    }
};
template <> struct Vec8<bf16_s> {
    static __device__ __forceinline__ void ld(const bf16_s* p, float (&o)[8]) {
        short8_t a = *reinterpret_cast<const short8_t*>(p);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = bf16_bits_to_f32((unsigned short)a[i]);
    }
    static __device__ __forceinline__ void st(bf16_s* p, const float (&v)[8]) {
        short8_t a;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = (short)f32_to_bf16_bits(v[i]);
        *reinterpret_cast<short8_t*>(p) = a;
    }
};
template <> struct Vec8<f16_s> {
    static __device__ __forceinline__ void ld(const f16_s* p, float (&o)[8]) {
        short8_t a = *reinterpret_cast<const short8_t*>(p);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = f16_bits_to_f32((unsigned short)a[i]);
    }
    static __device__ __forceinline__ void st(f16_s* p, const float (&v)[8]) {
        short8_t a;
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = (short)f32_to_f16_bits(v[i]);
        *reinterpret_cast<short8_t*>(p) = a;
    }
};

static inline int sbg_dtype_size(int dtype) { return dtype == SBG_F32 ? 4 : 2; }
static inline bool sbg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Grid size for streaming kernels: enough workgroups to fill 256 CUs x 8, grid-stride the rest.
static inline unsigned sbg_stream_grid(int64_t work_items, int block)
{
    int64_t need = (work_items + block - 1) / block;
    if (need < 1) need = 1;
    const int64_t cap = 256 * 8;
    return (unsigned)(need < cap ? need : cap);
}

// ------------------------------------------------------------------------------------------------
// Launch timing (see sbg_prof_* in include/sbg_hip.h).  Usage: { SbgProfScope prof(stream, kind, flops, bytes, dims); launch; }
bool sbg_prof_on();
int  sbg_prof_open(hipStream_t s, int kind, double flops, double bytes, const int* dims, int ndims);
void sbg_prof_close(hipStream_t s, int slot);

struct SbgProfScope {
    hipStream_t s; int slot;
    SbgProfScope(hipStream_t stream, int kind, double flops, double bytes, std::initializer_list<int> dims) : s(stream), slot(-1) {
        if (sbg_prof_on()) slot = sbg_prof_open(stream, kind, flops, bytes, dims.begin(), (int)dims.size());
    }
    ~SbgProfScope() { if (slot >= 0) sbg_prof_close(s, slot); }
};
