// undistort.h -- drop-in for NiftyMatch src/gpu/kernels/undistort.h:30-35.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdlib.h>

//! Radial (k1, k2, k3) distortion map in the OpenCV convention: for each position, (u, v) is where to read the
//! distorted image. camera_matrix = fx, fy, cx, cy; both parameter arrays are DEVICE pointers. cols / rows only size
//! the arrays; results are not clamped to the image.
void cuda_undistort(const float *x, const float *y, const size_t cols, const size_t rows, const float *camera_matrix,
                    const float *distortion_coeffs, float *u, float *v, hipStream_t stream = 0);
