// downsample.h -- drop-in for NiftyMatch src/gpu/kernels/downsample.h:21-24.
#ifndef __DOWNSAMPLE_H__
#define __DOWNSAMPLE_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

//! result[y][x] = source[2y][2x]; dimensions are passed explicitly by the caller. Instantiated for float and uchar4.
template <typename DataType>
void downsample_by_2(DataType *result, const int result_width, const int result_height, const DataType *source,
                     const int source_width, const int source_height, hipStream_t stream = 0);

#endif
