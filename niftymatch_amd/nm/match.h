// match.h -- drop-in for NiftyMatch src/gpu/kernels/match.h:18-46.
#ifndef __MATCH_H__
#define __MATCH_H__

#include <hip/hip_runtime_api.h>

//! A: sift_vector_size x size_A (transposed queries), B: size_B x sift_vector_size;
//! result[j*size_A + i] = squared L2 distance. Instantiated for float, sift_vector_size = 128.
template <typename TYPE>
void compute_brute_force_distance(const TYPE *A, const int size_A, const TYPE *B, const int size_B,
                                  const int sift_vector_size, TYPE *result, hipStream_t stream = 0);

//! Row-wise best / second best of \c distance (rows x cols, row stride buffer_width) + ratio test.
template <typename TYPE>
void get_sift_matches(const TYPE *distance, const int rows, const int cols, const int buffer_width, int *result,
                      float ambiguity = 0.8f, hipStream_t stream = 0);

#endif
