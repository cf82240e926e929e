// siftfunctions.h -- host orchestration of the SIFT path (drop-in for NiftyMatch src/gpu/sift/siftfunctions.h:19-101).
#ifndef __SIFTFUNCTIONS_H__
#define __SIFTFUNCTIONS_H__

#include <hip/hip_runtime_api.h>

#include "cudatex2D.h"
#include "pyramidata.h"
#include "siftdata.h"
#include "siftparams.h"

//! Matches of A's descriptors in B. \c distance (A._num_items x B._num_items, device) receives the squared L2
//! matrix; passing NULL skips materialising it (extension). A->_match_indexes receives the result.
void compute_sift_matches(SiftData *A, SiftData *B, float *distance, float ambiguity = 0.8f, hipStream_t stream = 0);

void compute_dog(PyramidData &pydata, const int octave_width, const int octave_height, hipStream_t stream = 0);

void compute_gradients(PyramidData &pydata, const SiftParams &params, const int octave_width,
                       const int octave_height, hipStream_t stream = 0);

void compute_keypoints(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                       const int octave_height, hipStream_t stream = 0);

//! \c mask: full-resolution device plane mask_width x mask_height (was a cudaTextureObject_t).
void compute_keypoints_with_mask(PyramidData &pydata, SiftParams &params, const float *mask, const int mask_width,
                                 const int mask_height, const int octave, const int octave_width,
                                 const int octave_height, hipStream_t stream = 0);

//! The reference's own argument list (sift/siftfunctions.h:70-73): \c mask is the texture view of the full-resolution
//! float mask (NmTexture takes the place of cudaTextureObject_t, see cudatex2D.h).
void compute_keypoints_with_mask(PyramidData &pydata, SiftParams &params, NmTexture mask, const int octave,
                                 const int octave_width, const int octave_height, hipStream_t stream = 0);

void compute_orientations(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                          const int octave_height, hipStream_t stream = 0);

void compute_descriptors(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                         const int octave_height, SiftData &data, hipStream_t stream = 0);

#endif
