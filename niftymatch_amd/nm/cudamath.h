// cudamath.h -- drop-in for NiftyMatch src/gpu/kernels/cudamath.h:18-87 (file name kept so client includes resolve).
#ifndef __CUDA_MATH_H__
#define __CUDA_MATH_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <math.h>

extern "C" int DivUp(int a, int b);      //!< ceil(a/b)
extern "C" int DivDown(int a, int b);    //!< a/b
extern "C" int AlignUp(int a, int b);    //!< smallest multiple of b that is >= a
extern "C" int AlignDown(int a, int b);  //!< largest multiple of b that is <= a

//! C = A - B
template <typename TYPE>
void subtract(const TYPE *A, const TYPE *B, TYPE *C, const int width, const int height, hipStream_t stream = 0);

//! result = (0.5*|grad|, angle in (0,2pi]) per interior pixel of \c source; border pixels are written as (0,0).
template <typename TYPE>
void gradient(const TYPE *source, float2 *result, const int width, const int height, hipStream_t stream = 0);

//! x modulo 2*pi by repeated subtraction / addition (reference: cudamath.h:82-87)
inline __host__ __device__ float mod_2pi_f(float x)
{
    const float two_pi = (float)(2 * M_PI);
    while (x > two_pi) x -= two_pi;
    while (x < 0.0F) x += two_pi;
    return x;
}

#endif
