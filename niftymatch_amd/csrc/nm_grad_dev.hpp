// nm_grad_dev.hpp -- device-side gradient (magnitude, angle) of kernels/cudamath.cu:38-54, shared by the Gaussian kernels'
// fused epilogue (nm_pyramid.hip) and the octave-tail kernel (nm_tail.hip). Header-only, inline.
#pragma once
#include "nm_common.hpp"
#include "nm_fpspec.hpp"

namespace nmgrad {

using nmfp::fma32;

// RN(sqrt(s)) for s = 0 or 2^-96 <= s < 2^96 without the denormal-safe expansion; other inputs take the IEEE path.
__device__ __forceinline__ float sqrt_rn_fast_core(float s)
{
    const float e = __builtin_amdgcn_rsqf(__builtin_fmaxf(s, 0x1p-100f));
    float y = s * e;
    const float h = 0.5f * e;
    y = fma32(fma32(-y, y, s), h, y);
    y = fma32(fma32(-y, y, s), h, y);
    return y;
}
__device__ __forceinline__ float sqrt_rn(float s)
{
    if (__builtin_expect((s >= 0x1p-96f || s == 0.0f) && s < 0x1p96f, 1)) return sqrt_rn_fast_core(s);
    return __builtin_sqrtf(s);
}

// (magnitude, angle) of kernels/cudamath.cu:38-54 from the 4-neighbourhood; same value sequence as gradient_kernel
__device__ __forceinline__ float2 grad_of(float xm, float xp, float ym, float yp)
{
    const float dx = xp - xm, dy = yp - ym;
    const float g = 0.5f * sqrt_rn(fma32(dx, dx, dy * dy));
    float r = 0.f;
    if (g != 0.0f) {
        r = (float)((double)nmfp::atan2f_spec(dy, dx) + nmfp::TWO_PI_D);      // in [pi, 3 pi]
        if (r > nmfp::TWO_PI_F) r -= nmfp::TWO_PI_F;                          // mod_2pi_f: one subtraction suffices
    }
    return make_float2(g, r);
}

}  // namespace nmgrad
