// nm_pk_dev.hpp -- packed-fp32 FMA helpers (v_pk_fma_f32 with the tap in a scalar register), shared by the Gaussian kernels of
// nm_pyramid.hip and the octave-tail kernel (nm_tail.hip). Header-only, inline.
#pragma once
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));

namespace nmpk {

// The same with the tap in a SCALAR register: op_sel_hi:[1,0,1] makes both halves of the packed FMA read the low dword of
// the scalar pair, so a uniform weight needs no VGPR pair and no v_mov to build one (the kernel spent 24 % of its VALU
// instructions on register moves, most of them broadcasting the 2R+1 taps into pairs for both passes).
__device__ __forceinline__ v2f pk_fma_s(v2f a, unsigned long long wq, v2f c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "s"(wq));
    return c;
}
// first tap of a sum that starts from +0: fma(a, w, +0) with the inline constant as the addend -- the accumulators need no
// zeroing moves (24 per thread and tile)
__device__ __forceinline__ v2f pk_fma_s0(v2f a, unsigned long long wq)
{
    v2f c;
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(c) : "v"(a), "s"(wq));
    return c;
}

}  // namespace nmpk
