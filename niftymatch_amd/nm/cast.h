// cast.h -- drop-in for NiftyMatch src/gpu/kernels/cast.h:17-22.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

//! dst = (TO)src, saturated at max_val when max_val != 0. Instantiated for <float, unsigned char>.
template <typename FROM, typename TO>
void cuda_cast(const FROM *src, const size_t cols, const size_t rows, TO *dst, TO max_val = 0, hipStream_t stream = 0);
