// LDS reads issued as inline assembly, with hand-counted waits.
//
// Why: a wave that fills LDS with `buffer_load ... lds` (LDS-DMA) and then reads LDS through the builtins gets an `s_waitcnt vmcnt(0)` from the
// compiler in front of the first read -- it cannot prove that the read does not alias the DMA destination, so the wave waits for the loads it
// has just issued and a prefetch never overlaps the compute it was meant to hide behind.  The kernels that prefetch from their compute waves
// (weight-gradient rows kernel, sliding-window FIR) therefore order stages themselves (counted `s_waitcnt vmcnt(N)` + barrier) and read with
// the instructions below, which the compiler neither schedules across each other nor counts.  Rules for users: LDS returns in issue order, so
// `lds_wait<N>` = "all but the last N reads have landed"; no scalar memory reads may sit between an issue and its wait (lgkmcnt counts
// them too and they return out of order) -- load kernel arguments before the loop; every register handed to lds_tr_issue must pass through an
// lds_wait before its first use (the "+v" operand is what orders the use behind the s_waitcnt).
#pragma once
#include <utility>

typedef __attribute__((ext_vector_type(4))) short sbg_short4_t;

template <int OFF>
static __device__ __forceinline__ void lds_tr_issue(sbg_short4_t& d, unsigned addr)      // ds_read_b64_tr_b16 d, addr offset:OFF
{
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
static __device__ __forceinline__ void lds_wait(sbg_short4_t& a, sbg_short4_t& b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
static __device__ __forceinline__ void lds_wait(sbg_short4_t (&a)[4], sbg_short4_t (&b)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

template <class F, int... Is>
static __device__ __forceinline__ void sbg_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
static __device__ __forceinline__ void sbg_static_for(F&& f) { sbg_static_for_impl(std::make_integer_sequence<int, N>{}, f); }      // compile-time unrolled loop: f(integral_constant<int, i>)
