// nm_devmem.hpp -- inter-workgroup hand-off primitives for work items of ONE launch (gfx950: 8 XCDs with private L2s, a
// vector L1 per CU that other CUs' stores never refresh). Used by the octave-tail kernel (nm_tail.hip) and the detection
// body it shares with the per-octave launch (nm_detect_dev.hpp). The forms follow the CDNA4 guide's recipe for exchanging
// data inside a launch:
//   producer: every handed-off byte is stored WRITE-THROUGH (agent-scope relaxed atomic store = global_store ... sc1), every
//             storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds to an agent-scope counter;
//   consumer: ONE lane polls that counter with relaxed agent-scope loads (sc1: served past the L1) and s_sleep, then ONE
//             agent-scope acquire fence (buffer_inv sc1: drops this CU's L1 lines), s_waitcnt vmcnt(0), a barrier, and only then
//             plain loads of the payload by every wave.
// Every shared word is accessed through a GLOBAL (address_space(1)) pointer, never a flat one, and never by a plain store.
#pragma once
#include <hip/hip_runtime.h>

namespace nmdev {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;

__device__ __forceinline__ gu32 *g32(const void *p) { return (gu32 *)(unsigned long long)p; }
__device__ __forceinline__ gu64 *g64(const void *p) { return (gu64 *)(unsigned long long)p; }
__device__ __forceinline__ gi32 *gi(const void *p) { return (gi32 *)(unsigned long long)p; }

__device__ __forceinline__ void store_f32_agent(float *p, float v)
{
    __hip_atomic_store(g32(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_i32_agent(int *p, int v)
{
    __hip_atomic_store(gi(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_f2_agent(float2 *p, float a, float b)      // p 8-byte aligned
{
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(g64(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte write-through store (p 16-byte aligned): ONE global_store_dwordx4 sc1. Narrow sc1 stores are one fabric write per
// lane (a dword costs ~6x the time per byte of a dwordx4), so bulk payload goes out 16 bytes per lane. The compiler does not
// count an asm store in vmcnt; publish_add's explicit s_waitcnt vmcnt(0) does (the hardware counter sees it).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_f4_agent(float4 *p, float4 v)
{
    const v4f_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"((unsigned long long)p), "v"(d) : "memory");
}
__device__ __forceinline__ int load_i32_agent(const int *p)
{
    return __hip_atomic_load(gi(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int add_i32_agent(int *p, int v)
{
    return __hip_atomic_fetch_add(gi(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Producer side, called by ALL threads of the workgroup after their write-through stores: drain, barrier, one add.
__device__ __forceinline__ void publish_add(int *counter, int v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) add_i32_agent(counter, v);
}

// Consumer side, called by ALL threads: thread 0 polls until *counter >= target (bounded: ~2 s, then the launch's error
// word is set and false is returned to every thread), one acquire, barrier. s_flag: one int of LDS.
constexpr unsigned NM_SPIN_LIMIT = 1u << 23;
__device__ __forceinline__ bool wait_counters(const int *c0, int t0, const int *c1, int t1, int *error_word, int *s_flag)
{
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while ((c0 && load_i32_agent(c0) < t0) || (c1 && load_i32_agent(c1) < t1)) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > NM_SPIN_LIMIT || load_i32_agent(error_word) != 0) { ok = 0; store_i32_agent(error_word, 1); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = ok;
    }
    __syncthreads();
    const bool ok = *s_flag != 0;
    __syncthreads();                                           // s_flag may be rewritten by the next wait
    return ok;
}

}  // namespace nmdev
