// siftfunctions.cpp -- host orchestration (reference: src/gpu/sift/siftfunctions.cu:15-181) on the C ABI / L2 launchers.
#include "../siftfunctions.h"

#include <cmath>

#include "../../../include/nm_abi.h"
#include "../cudamath.h"
#include "../descriptor.h"
#include "../exception.h"
#include "../keypoint.h"
#include "../match.h"
#include "../orientation.h"
#include "../transpose.h"

// The reference transposes A, builds the transposed matrix, transposes it back and scans rows (siftfunctions.cu:21-39).
// Here one fused MFMA pass finds the candidates and the exact pass decides; `distance` is filled only when asked for.
void compute_sift_matches(SiftData *A, SiftData *B, float *distance, float ambiguity, hipStream_t stream)
{
    const int A_size = A->_num_items;
    const int B_size = B->_num_items;
    if (A_size <= 0 || B_size <= 0) return;
    nm::device_vector<int> ws(nm_sift_match_workspace_bytes(A_size, B_size) / sizeof(int) + 1);
    nm_check(nm_sift_match_f32(A->_desc.data(), A_size, B->_desc.data(), B_size, distance, A->_match_indexes.data(),
                               ambiguity, ws.data(), stream),
             "SIFT matching failed");
    nm_check((int)hipStreamSynchronize(stream), "SIFT matching failed");   // ws is released on return
}

void compute_dog(PyramidData &pydata, const int octave_width, const int octave_height, hipStream_t stream)
{
    for (int i = 0; i < pydata._num_dogs; ++i)
        subtract<float>(pydata._octave[i + 1].data(), pydata._octave[i].data(), pydata._dog[i].data(), octave_width,
                        octave_height, stream);
}

void compute_gradients(PyramidData &pydata, const SiftParams &params, const int octave_width, const int octave_height,
                       hipStream_t stream)
{
    float2 *g = pydata._grad.data();
    const size_t offset = (size_t)octave_width * octave_height;
    for (int i = params._level_min + 1; i <= params._level_max - 2; ++i)
        gradient<float>(pydata._octave[i + 1].data(), g + i * offset, octave_width, octave_height, stream);
}

static void keypoints_impl(PyramidData &pydata, const SiftParams &params, const float *mask, int mask_w, int mask_h,
                           const int octave, const int ow, const int oh, hipStream_t stream)
{
    const float xper = std::pow(2.0, octave);
    for (int i = 1; i < pydata._num_dogs - 1; ++i) {
        // the whole full-resolution map is reset, as in the reference (siftfunctions.cu:120-121)
        nm_check(nm_fill_u32(pydata._key_pts[i - 1].data(), pydata._key_pts[i - 1].size() * 4, 0xBF800000u, stream),
                 "Keypoint map reset failed");
        if (mask)
            find_keypoints(pydata._dog[i].data(), mask, mask_w, mask_h, pydata._dog[i - 1].data(),
                           pydata._dog[i + 1].data(), ow, oh, params._peak_threshold, params._edge_threshold, xper,
                           params._sigma_0, params._num_dog_levels, i - 1, pydata._key_pts[i - 1].data(), stream);
        else
            find_keypoints(pydata._dog[i].data(), pydata._dog[i - 1].data(), pydata._dog[i + 1].data(), ow, oh,
                           params._peak_threshold, params._edge_threshold, xper, params._sigma_0,
                           params._num_dog_levels, i - 1, pydata._key_pts[i - 1].data(), stream);
    }
}

void compute_keypoints(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                       const int octave_height, hipStream_t stream)
{
    keypoints_impl(pydata, params, nullptr, 0, 0, octave, octave_width, octave_height, stream);
}

void compute_keypoints_with_mask(PyramidData &pydata, SiftParams &params, const float *mask, const int mask_width,
                                 const int mask_height, const int octave, const int octave_width,
                                 const int octave_height, hipStream_t stream)
{
    keypoints_impl(pydata, params, mask, mask_width, mask_height, octave, octave_width, octave_height, stream);
}

void compute_orientations(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                          const int octave_height, hipStream_t stream)
{
    const float xper = std::pow(2.0, octave);
    const int num_pixels_for_octave = octave_width * octave_height;
    if (stream) nm_check((int)hipStreamSynchronize(stream), "stream sync");   // collation runs on the NULL stream
    for (int i = 0; i < params._num_dog_levels; ++i) {
        pydata.gpu_collate_keypoints_for_level(i, num_pixels_for_octave);
        if (pydata._orientations[i].size() == 0) return;       // an empty level ends the octave (siftfunctions.cu:145)
        detect_orientations(pydata._collated_kpts[i].data(), pydata._grad.data(), (int)pydata._orientations[i].size(),
                            octave_width, octave_height, 1.5f, xper, pydata._orientations[i].data(), stream);
    }
}

void compute_descriptors(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                         const int octave_height, SiftData &data, hipStream_t stream)
{
    const float xper = std::pow(2.0, octave);
    for (int i = 0; i < params._num_dog_levels; ++i) {
        if (pydata._orientations[i].size() == 0) return;       // siftfunctions.cu:160
        int num_pts = (int)pydata._orientations[i].size();
        const int capacity = (int)(data._desc.size() / SIFT_VECTOR_SIZE);
        if (num_pts + data._num_items > capacity) num_pts = capacity - data._num_items;
        if (num_pts > 0) {
            compute_sift_descriptors(pydata._collated_kpts[i].data(), pydata._orientations[i].data(),
                                     pydata._grad.data(), num_pts, octave_width, octave_height,
                                     params._num_dog_levels, xper,
                                     data._desc.data() + (size_t)data._num_items * SIFT_VECTOR_SIZE,
                                     data._x.data() + data._num_items, data._y.data() + data._num_items, stream);
            data._num_items += num_pts;
        }
    }
}
