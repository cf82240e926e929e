// ransac.h -- drop-in for the part of NiftyMatch src/gpu/kernels/ransac.h that consumes the matcher's output directly.
// align_points (ransac.h:8-10) is provided; the RANSAC model fits (ransac_homography / _translation / _similarity,
// ransac.h:12-22) are out of scope of this path and are not declared.
#ifndef __RANSAC_H__
#define __RANSAC_H__

#include <hip/hip_runtime_api.h>

void align_points(const float *src_x, const float *src_y, const float *dst_x, const float *dst_y, float *c_src_x,
                  float *c_src_y, float *c_dst_x, float *c_dst_y, const int *matches, const int num_pts,
                  hipStream_t stream = 0);

#endif
