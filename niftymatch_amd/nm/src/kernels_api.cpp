// kernels_api.cpp -- C++ launchers of the `kernels` library: the reference's L2 signatures on top of the C ABI.
// Error convention: a failing launch prints file:line + the HIP error and exit()s, as helper_cuda.h's
// getLastCudaError does in the reference (e.g. kernels/convolution.cu:152,158, kernels/match.cu:134).
#include <string>

#include "../../../include/nm_abi.h"
#include "../bgra_2_gray.h"
#include "../cast.h"
#include "../convolution.h"
#include "../cudamath.h"
#include "../descriptor.h"
#include "../downsample.h"
#include "../exception.h"
#include "../keypoint.h"
#include "../match.h"
#include "../orientation.h"
#include "../ransac.h"
#include "../resample.h"
#include "../transpose.h"
#include "../undistort.h"

template <typename TYPE>
void convolve(TYPE *result, const TYPE *image, TYPE *buffer, const int width, const int height, const float *kernel,
              const int kernel_radius, hipStream_t stream)
{
    nm_check(nm_convolve_f32(result, image, buffer, width, height, kernel, kernel_radius, stream), "Convolution failed");
}
template void convolve<float>(float *, const float *, float *, const int, const int, const float *, const int, hipStream_t);

template <typename DataType>
void downsample_by_2(DataType *result, const int result_width, const int result_height, const DataType *source,
                     const int source_width, const int source_height, hipStream_t stream)
{
    nm_check(nm_downsample2_f32(result, result_width, result_height, source, source_width, source_height, stream),
             "Downsampling kernel failed");
}
template void downsample_by_2<float>(float *, const int, const int, const float *, const int, const int, hipStream_t);

template <typename TYPE>
void subtract(const TYPE *A, const TYPE *B, TYPE *C, const int width, const int height, hipStream_t stream)
{
    nm_check(nm_subtract_f32(A, B, C, width, height, stream), "Subtract launch failed");
}
template void subtract<float>(const float *, const float *, float *, const int, const int, hipStream_t);

template <typename TYPE>
void gradient(const TYPE *source, float2 *result, const int width, const int height, hipStream_t stream)
{
    nm_check(nm_gradient_f32(source, reinterpret_cast<float *>(result), width, height, stream),
             "Set gradient launch failed");
}
template void gradient<float>(const float *, float2 *, const int, const int, hipStream_t);

void find_keypoints(const float *current, const float *down, const float *up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream)
{
    nm_check(nm_find_keypoints_f32(current, down, up, width, height, peak_threshold, edge_threshold, xper, sigma_0,
                                   num_dogs, dog, reinterpret_cast<float *>(result), stream),
             "Keypoint detection launch failed");
}

void find_keypoints(const float *current, const float *mask, const int mask_width, const int mask_height,
                    const float *down, const float *up, const int width, const int height, const float peak_threshold,
                    const float edge_threshold, const float xper, const float sigma_0, const int num_dogs, const int dog,
                    float4 *result, hipStream_t stream)
{
    nm_check(nm_find_keypoints_masked_f32(current, mask, mask_width, mask_height, down, up, width, height,
                                          peak_threshold, edge_threshold, xper, sigma_0, num_dogs, dog,
                                          reinterpret_cast<float *>(result), stream),
             "Keypoint detection launch failed");
}

static const float *plane_of(const NmTexture &t, int width, int height, const char *what)
{
    if (t.format != NM_TEXEL_F32 || !t.data) RUNTIME_EXCEPTION(std::string("find_keypoints: ") + what + " must be a float texture");
    if (width > 0 && (t.width != width || t.height != height))
        RUNTIME_EXCEPTION(std::string("find_keypoints: ") + what + " does not have the octave's geometry");
    return static_cast<const float *>(t.data);
}

void find_keypoints(NmTexture current, NmTexture down, NmTexture up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream)
{
    find_keypoints(plane_of(current, width, height, "current"), plane_of(down, width, height, "down"),
                   plane_of(up, width, height, "up"), width, height, peak_threshold, edge_threshold, xper, sigma_0, num_dogs,
                   dog, result, stream);
}

void find_keypoints(NmTexture current, NmTexture mask, NmTexture down, NmTexture up, const int width, const int height,
                    const float peak_threshold, const float edge_threshold, const float xper, const float sigma_0,
                    const int num_dogs, const int dog, float4 *result, hipStream_t stream)
{
    find_keypoints(plane_of(current, width, height, "current"), plane_of(mask, 0, 0, "mask"), mask.width, mask.height,
                   plane_of(down, width, height, "down"), plane_of(up, width, height, "up"), width, height, peak_threshold,
                   edge_threshold, xper, sigma_0, num_dogs, dog, result, stream);
}

void detect_orientations(const float4 *key_pts, const float2 *grad, const int num_pts, const int octave_width,
                         const int octave_height, float gauss_factor, const float xper, float2 *result,
                         hipStream_t stream)
{
    nm_check(nm_detect_orientations(reinterpret_cast<const float *>(key_pts), reinterpret_cast<const float *>(grad),
                                    num_pts, octave_width, octave_height, gauss_factor, xper,
                                    reinterpret_cast<float *>(result), stream),
             "Orientation histogram launch failed");
}

void compute_sift_descriptors(const float4 *key_pts, const float2 *orients, const float2 *grad, const int num_pts,
                              const int octave_width, const int octave_height, const int num_dogs, const float xper,
                              float *desc, float *x, float *y, hipStream_t stream)
{
    nm_check(nm_compute_sift_descriptors(reinterpret_cast<const float *>(key_pts),
                                         reinterpret_cast<const float *>(orients),
                                         reinterpret_cast<const float *>(grad), num_pts, octave_width, octave_height,
                                         num_dogs, xper, desc, x, y, stream),
             "SIFT descriptor detection launch failed");
}

template <typename TYPE>
void transpose(TYPE *odata, const TYPE *idata, int width, int height, hipStream_t stream)
{
    nm_check(nm_transpose_f32(odata, idata, width, height, stream), "Transpose kernel failed");
}
template void transpose<float>(float *, const float *, int, int, hipStream_t);

template <typename TYPE>
void compute_brute_force_distance(const TYPE *A, const int size_A, const TYPE *B, const int size_B,
                                  const int sift_vector_size, TYPE *result, hipStream_t stream)
{
    nm_check(nm_bf_distance_f32(A, size_A, B, size_B, sift_vector_size, result, stream),
             "Brute force distance computation launch failed");
}
template void compute_brute_force_distance<float>(const float *, const int, const float *, const int, const int,
                                                  float *, hipStream_t);

template <typename TYPE>
void get_sift_matches(const TYPE *distance, const int rows, const int cols, const int buffer_width, int *result,
                      float ambiguity, hipStream_t stream)
{
    nm_check(nm_get_sift_matches_f32(distance, rows, cols, buffer_width, result, ambiguity, stream),
             "Set matches launch failed");
}
template void get_sift_matches<float>(const float *, const int, const int, const int, int *, float, hipStream_t);

template <>
void downsample_by_2<uchar4>(uchar4 *result, const int result_width, const int result_height, const uchar4 *source,
                             const int source_width, const int source_height, hipStream_t stream)
{
    nm_check(nm_downsample2_u8x4(reinterpret_cast<unsigned char *>(result), result_width, result_height,
                                 reinterpret_cast<const unsigned char *>(source), source_width, source_height, stream),
             "Downsampling kernel failed");
}

template <typename OutputType>
void cuda_grayscale(const uchar4 *bgra, OutputType *output, const int width, const int height, hipStream_t stream)
{
    nm_check(nm_grayscale_f32(reinterpret_cast<const unsigned char *>(bgra), output, width, height, stream),
             "CUDA grayscale launch failed");
}
template void cuda_grayscale<float>(const uchar4 *, float *, const int, const int, hipStream_t);

template <typename OutputType>
void cuda_extract_channel(const uchar4 *bgra, OutputType *output, const int width, const int height, const int channel,
                          hipStream_t stream)
{
    nm_check(nm_extract_channel_f32(reinterpret_cast<const unsigned char *>(bgra), output, width, height, channel, stream),
             "CUDA extract channel launch failed");
}
template void cuda_extract_channel<float>(const uchar4 *, float *, const int, const int, const int, hipStream_t);

template <typename InputType>
void cuda_put_channel(uchar4 *bgra, const InputType *input, const int width, const int height, const int channel,
                      hipStream_t stream)
{
    nm_check(nm_put_channel_f32(reinterpret_cast<unsigned char *>(bgra), input, width, height, channel, stream),
             "CUDA put channel launch failed");
}
template void cuda_put_channel<float>(uchar4 *, const float *, const int, const int, const int, hipStream_t);

void cuda_set_alpha_to_const(uchar4 *bgra, const int width, const int height, const unsigned char val, hipStream_t stream)
{
    nm_check(nm_set_alpha_to_const(reinterpret_cast<unsigned char *>(bgra), width, height, val, stream),
             "CUDA set alpha launch failed");
}

template <typename FROM, typename TO>
void cuda_cast(const FROM *src, const size_t cols, const size_t rows, TO *dst, TO max_val, hipStream_t stream)
{
    nm_check(nm_cast_f32_u8(src, cols, rows, dst, max_val, stream), "Cast kernel launch failed");
}
template void cuda_cast<float, unsigned char>(const float *, const size_t, const size_t, unsigned char *, unsigned char,
                                              hipStream_t);

void align_points(const float *src_x, const float *src_y, const float *dst_x, const float *dst_y, float *c_src_x,
                  float *c_src_y, float *c_dst_x, float *c_dst_y, const int *matches, const int num_pts,
                  hipStream_t stream)
{
    nm_check(nm_align_points(src_x, src_y, dst_x, dst_y, c_src_x, c_src_y, c_dst_x, c_dst_y, matches, num_pts, stream),
             "Align points launch failed");
}

// ---- N3/N4: undistort.h, resample.h ----
void cuda_undistort(const float *x, const float *y, const size_t cols, const size_t rows, const float *camera_matrix,
                    const float *distortion_coeffs, float *u, float *v, hipStream_t stream)
{
    nm_check(nm_undistort_map_f32(x, y, cols, rows, camera_matrix, distortion_coeffs, u, v, stream),
             "Undistort kernel launch failed");
}

void resample_undistort(NmTexture tex, const float *x, const float *y, const size_t cols, const size_t rows,
                        float *undistorted, hipStream_t stream)
{
    nm_check(nm_resample_undistort_f32(tex.data, tex.width, tex.height, tex.format, x, y, cols, rows, undistorted, stream),
             "Resample 2D image kernel launch failed for undistort");
}

void resample_mask(unsigned char *result, NmTexture text, const int cols, const int rows, const float *x_pos,
                   const float *y_pos, const float threshold, hipStream_t stream)
{
    nm_check(nm_resample_mask_u8(result, text.data, text.width, text.height, text.format, cols, rows, x_pos, y_pos,
                                 threshold, stream),
             "Resample 2D mask launch failed");
}

void resample_perspective_transform(uchar4 *result, NmTexture text, const int cols, const int rows, float *x_pos,
                                    float *y_pos, const float *mat3x3, bool inverse, hipStream_t stream)
{
    nm_check(text.format == NM_TEXEL_U8X4_NORM ? 0 : (int)hipErrorInvalidValue, "Resample 2D image needs a uchar4 texture");
    nm_check(nm_resample_perspective_u8x4(reinterpret_cast<unsigned char *>(result),
                                          static_cast<const unsigned char *>(text.data), text.width, text.height, cols,
                                          rows, x_pos, y_pos, mat3x3, inverse ? 1 : 0, stream),
             "Resample 2D image launch failed");
}

void transform_blend(uchar4 *canvas, const int cw, const int ch, NmTexture frame, const int fw, const int fh, const int nw,
                     const int nh, const float *mat3x3, const int tx, const int ty, NmTexture frame_mask,
                     float *canvas_wts, NmTexture frame_wts, hipStream_t stream)
{
    nm_check(frame.format == NM_TEXEL_U8X4_NORM ? 0 : (int)hipErrorInvalidValue, "Blend needs a uchar4 frame texture");
    nm_check(nm_transform_blend(reinterpret_cast<unsigned char *>(canvas), cw, ch,
                                static_cast<const unsigned char *>(frame.data), fw, fh, nw, nh, mat3x3, tx, ty,
                                frame_mask.data, frame_mask.format, canvas_wts, frame_wts.data, frame_wts.format, stream),
             "Blend launch failed");
}
