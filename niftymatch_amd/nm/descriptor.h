// descriptor.h -- drop-in for NiftyMatch src/gpu/kernels/descriptor.h:25-30.
#ifndef __DESCRIPTOR_H__
#define __DESCRIPTOR_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

void compute_sift_descriptors(const float4 *key_pts, const float2 *orients, const float2 *grad, const int num_pts,
                              const int octave_width, const int octave_height, const int num_dogs, const float xper,
                              float *desc, float *x, float *y, hipStream_t stream = 0);

#endif
