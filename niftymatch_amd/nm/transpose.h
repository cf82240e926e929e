// transpose.h -- out-of-place matrix transpose of the `kernels` library (NiftyMatch: src/gpu/kernels/transpose.h:16-21).
// compute_sift_matches no longer needs it (the fused matcher reads A and B row-major), but client code that feeds
// compute_brute_force_distance its transposed query set still does.
#ifndef __TRANSPOSE_H__
#define __TRANSPOSE_H__

#include <hip/hip_runtime_api.h>

//! odata (height rows x width columns becomes width rows x height columns): odata[x * height + y] = idata[y * width + x].
//! Both buffers are device memory, width * height elements each, and must not overlap. Instantiated for float.
//! \param width number of columns of idata
//! \param height number of rows of idata
template <typename TYPE>
void transpose(TYPE *odata, const TYPE *idata, int width, int height, hipStream_t stream = 0);

#endif
