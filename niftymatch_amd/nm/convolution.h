// convolution.h -- drop-in for NiftyMatch src/gpu/kernels/convolution.h:19-23 (cudaStream_t -> hipStream_t).
#ifndef __CONVOLUTION_H__
#define __CONVOLUTION_H__

#include <hip/hip_runtime_api.h>

//! Separable zero-padded correlation of \c image with the 2*kernel_radius+1 taps in device memory \c kernel:
//! \c buffer receives the row pass, \c result the column pass. Instantiated for float.
template <typename TYPE>
void convolve(TYPE *result, const TYPE *image, TYPE *buffer, const int width, const int height,
              const float *kernel, const int kernel_radius, hipStream_t stream = 0);

#endif
