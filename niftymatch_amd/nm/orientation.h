// orientation.h -- drop-in for NiftyMatch src/gpu/kernels/orientation.h:19-24.
#ifndef __ORIENTATION_H__
#define __ORIENTATION_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

void detect_orientations(const float4 *key_pts, const float2 *grad, const int num_pts, const int octave_width,
                         const int octave_height, float gauss_factor, const float xper, float2 *result,
                         hipStream_t stream = 0);

#endif
