// ransac.h -- drop-in for NiftyMatch src/gpu/kernels/ransac.h:8-22: the consumers of the matcher's output.
// All pointers are device memory; `homography` receives 9 floats (row-major 3x3) on the device.
#ifndef __RANSAC_H__
#define __RANSAC_H__

#include <hip/hip_runtime_api.h>

//! Gather matched coordinates: row i gets (src[i], dst[matches[i]]) or (-1,-1,-1,-1) when matches[i] == -1.
void align_points(const float *src_x, const float *src_y, const float *dst_x, const float *dst_y, float *c_src_x,
                  float *c_src_y, float *c_dst_x, float *c_dst_y, const int *matches, const int num_pts,
                  hipStream_t stream = 0);

//! RANSAC over `iterations` random minimal samples of the points with src_x >= 0; returns false (and prints the count,
//! like the reference) when there are fewer than 4 such points. dst_size is unused, as in the reference.
bool ransac_homography(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int dst_size,
                       float inlier_threshold, int iterations, float *homography, hipStream_t stream = 0);

//! Same for a pure translation (1 sample per hypothesis; needs >= 2 valid points, as the reference checks).
bool ransac_translation(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int dst_size,
                        float inlier_threshold, int iterations, float *homography, hipStream_t stream = 0);

//! Same for a similarity transform (2 samples per hypothesis).
bool ransac_similarity(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int dst_size,
                       float inlier_threshold, int iterations, float *homography, hipStream_t stream = 0);

//! Extension: fix the seed of the host-side sampler (the reference always seeds from std::random_device, which is what
//! seed 0 selects here). Declared with C linkage in nm_abi.h as well.
extern "C" void nm_ransac_seed(unsigned int seed);

#endif
