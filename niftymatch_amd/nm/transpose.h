// transpose.h -- drop-in for NiftyMatch src/gpu/kernels/transpose.h:16-21.
#ifndef __TRANSPOSE_H__
#define __TRANSPOSE_H__

#include <hip/hip_runtime_api.h>

template <typename TYPE>
void transpose(TYPE *odata, const TYPE *idata, int width, int height, hipStream_t stream = 0);

#endif
