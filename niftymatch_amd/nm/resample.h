// resample.h -- drop-in for NiftyMatch src/gpu/kernels/resample.h:7-38. cudaTextureObject_t -> NmTexture (cudatex2D.h).
#ifndef __KERNEL_RESAMPLE_H__
#define __KERNEL_RESAMPLE_H__

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include "cudatex2D.h"

//! Writes the (inverse, by default) projective image of the cols x rows pixel grid into x_pos / y_pos and samples the
//! uchar4 texture there.
void resample_perspective_transform(uchar4 *result, NmTexture text, const int cols, const int rows, float *x_pos,
                                    float *y_pos, const float *mat3x3, bool inverse = true, hipStream_t stream = 0);

//! result = 0 where the sampled mask is <= threshold, else sample * 255.999
void resample_mask(unsigned char *result, NmTexture text, const int cols, const int rows, const float *x_pos,
                   const float *y_pos, const float threshold = 0.5f, hipStream_t stream = 0);

//! Warps the nw x nh frame by mat3x3 and accumulates it, weight-averaged, into canvas / canvas_wts at offset (tx, ty).
//! frame, frame_mask and frame_wts all are fw x fh.
void transform_blend(uchar4 *canvas, const int cw, const int ch, NmTexture frame, const int fw, const int fh,
                     const int nw, const int nh, const float *mat3x3, const int tx, const int ty, NmTexture frame_mask,
                     float *canvas_wts, NmTexture frame_wts, hipStream_t stream = 0);

//! undistorted[i] = tex(x[i], y[i]) * 255.9999
void resample_undistort(NmTexture tex, const float *x, const float *y, const size_t cols, const size_t rows,
                        float *undistorted, hipStream_t stream = 0);

#endif
