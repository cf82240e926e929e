// bgra_2_gray.h -- drop-in for NiftyMatch src/gpu/kernels/bgra_2_gray.h:14-66.
#ifndef __BGRA_2_GRAY_H__
#define __BGRA_2_GRAY_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

//! output = 0.07 B + 0.72 G + 0.21 R. Instantiated for float.
template <typename OutputType>
void cuda_grayscale(const uchar4 *bgra, OutputType *output, const int width, const int height, hipStream_t stream = 0);

//! channel 0..3 = B, G, R, A. Instantiated for float.
template <typename OutputType>
void cuda_extract_channel(const uchar4 *bgra, OutputType *output, const int width, const int height, const int channel,
                          hipStream_t stream = 0);

//! channel 0..2 receive (unsigned char)input; channel 3 is set to 255. Instantiated for float.
template <typename InputType>
void cuda_put_channel(uchar4 *bgra, const InputType *input, const int width, const int height, const int channel,
                      hipStream_t stream = 0);

void cuda_set_alpha_to_const(uchar4 *bgra, const int width, const int height, const unsigned char val = 255,
                             hipStream_t stream = 0);

#endif
