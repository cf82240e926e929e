// keypoint.h -- drop-in for NiftyMatch src/gpu/kernels/keypoint.h:25-59.
// Deviation: the reference's cudaTextureObject_t arguments are plain device planes (const float*): its fetches are
// exact texel loads (utils/cudatex2D.cu:15-19). The masked overload takes the full-resolution mask plane + its size.
#ifndef __KEYPOINT_H__
#define __KEYPOINT_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "cudatex2D.h"

void find_keypoints(const float *current, const float *down, const float *up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream = 0);

void find_keypoints(const float *current, const float *mask, const int mask_width, const int mask_height,
                    const float *down, const float *up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream = 0);

// The reference's argument lists (kernels/keypoint.h:25-59) with NmTexture in the place of cudaTextureObject_t: float
// planes of exactly width x height texels for current / down / up, the full-resolution float mask for \c mask.
void find_keypoints(NmTexture current, NmTexture down, NmTexture up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream = 0);

void find_keypoints(NmTexture current, NmTexture mask, NmTexture down, NmTexture up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream = 0);

#endif
