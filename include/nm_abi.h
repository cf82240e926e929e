/*
 * include/nm_abi.h -- C ABI of libnm_hip.so: the MI355X (gfx950) drop-in for NiftyMatch's SIFT detect/describe +
 * brute-force L2 match path.
 *
 * Every entry point replaces one host launcher of the reference (file:line under /root/reference/src/gpu/) and
 * keeps its argument order and meaning. Differences common to all of them:
 *   - plain C: raw DEVICE pointers, ints, floats; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - float2/float4 arrays are passed as float* with the same memory layout (x,y[,z,w] interleaved);
 *   - the return value is 0 on success or the hipError_t of the failing launch / API call (the reference prints
 *     and exit()s inside the launcher, helper_cuda.h getLastCudaError; the C++ wrappers in niftymatch_amd/nm/ restore
 *     that convention on top of this status);
 *   - textures are replaced by plain row-major planes (the reference's fetches are exact texel loads,
 *     utils/cudatex2D.cu:15-19);
 *   - all launches are asynchronous on `stream`; none allocates, frees or synchronises unless stated.
 * All functions are re-entrant; all state lives in the arguments.
 */
#ifndef NM_ABI_H
#define NM_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NM_API __attribute__((visibility("default")))
#else
#define NM_API
#endif

/* ---- integer helpers: the reference's only extern "C" symbols (kernels/cudamath.h:18-45, cudamath.cu:5-23) ---- */
NM_API int DivUp(int a, int b);
NM_API int DivDown(int a, int b);
NM_API int AlignUp(int a, int b);
NM_API int AlignDown(int a, int b);

/* ---- library / device ---- */
NM_API const char *nm_version(void);                 /* "niftymatch_amd <ver> gfx950" */
NM_API int nm_device_count(int *count);              /* hipGetDeviceCount */
NM_API int nm_set_device(int device);                /* replaces CudaUtils::setup_CUDA (utils/cudautils.cpp:19-28) */
NM_API const char *nm_error_string(int status);      /* hipGetErrorString */

/* Fill `count` 32-bit words at device pointer dst with `pattern` (replaces thrust::fill / device_vector value
 * initialisation, sift/siftfunctions.cu:84-85,120-121, sift/pyramidata.cu:42-46).                                */
NM_API int nm_fill_u32(void *dst, size_t count, unsigned int pattern, void *stream);

/* Profiling hook (no reference counterpart; the reference's CudaTimer brackets whole calls, utils/cudatimer.cu:3-22).
 * While a (start, stop) hipEvent_t pair is registered for `site`, the launcher records start immediately before and
 * stop immediately after that kernel (sequence) on the stream it launches on. Pass NULLs to clear. Per host thread.
 *   NM_PROF_MATCH_TOP2 : the MFMA screening kernel inside nm_sift_match_f32 / _shard_f32 / _batch[_dev]_f32: one launch per
 *                        pair under the fp32 and bf16x3 screens; under the two-stage screen ONE launch (its coarse pass)
 *                        for all pairs of the call
 *   NM_PROF_PYRAMID_O0 : the octave-0 pyramid sequence (5 fused Gaussian + DoG + gradient launches) inside
 *                        nm_sift_detect_describe[_batch] / nm_sift_octave_pyramid (all frames of a batch)       */
#define NM_PROF_MATCH_TOP2 0
#define NM_PROF_PYRAMID_O0 1
#define NM_PROF_DESCRIBE 2      /* frame_desc_kernel inside nm_sift_detect_describe[_batch] (all frames of the call) */
#define NM_PROF_ORIENT 3        /* frame_orient_kernel of the same calls */
#define NM_PROF_DETECT_O0 4     /* detect_stage_kernel of octave 0 of the same calls (per-octave launches, not the tail launch) */
#define NM_PROF_DISTANCE 5      /* distance_mfma_kernel inside nm_sift_match_f32 when `distance` is materialised on the MFMA */
#define NM_PROF_SITES 6
NM_API int nm_profile_events(int site, void *start_event, void *stop_event);
/* The same with a caller-owned list of npairs (start, stop) hipEvent_t pairs, events[2k], events[2k+1], consumed by the
 * k-th launch of the site (a batched call launches the MFMA kernel once per pair; under the two-stage screen, whose coarse
 * pass is one launch per call, the call's FIRST pair brackets that launch and the pairs of its other pairs are recorded
 * back to back, so that a list of one pair per matched pair stays valid and sums to the launch). npairs = 0 clears. */
NM_API int nm_profile_event_pairs(int site, void *const *events, int npairs);

/* Self-test (no reference counterpart): the gradient's fast correctly-rounded square root is compared with the IEEE
 * expansion for EVERY float of its domain [2^-96, 2^96) and 0; *d_mismatches (device) receives the number of differing
 * inputs. Must be 0. */
NM_API int nm_selftest_sqrt(unsigned long long *d_mismatches, void *stream);
/* Self-test (no reference counterpart): the descriptor's window weight (float)exp((nx^2 + ny^2) / 8) (kernels/descriptor.cu:108)
 * is evaluated on voting samples by a division-free form whose result is proven equal to the spec's binary64 sequence unless
 * it reports a nearby binary32 rounding boundary (then the spec sequence runs). Every float of the form's domain [0, 12.875] is
 * compared. d_out (device, 3 x unsigned long long): [0] unreported differences (must be 0), [1] inputs that report a nearby
 * boundary, [2] inputs tested. */
NM_API int nm_selftest_expw(unsigned long long *d_out, void *stream);
/* Self-test (no reference counterpart): the orientation kernel (kernels/orientation.cu:39-75, 181-192) hoists a keypoint's part of
 * each vote's arithmetic -- the window test in binary32, r2 / (2 sigma^2) with the divisor's reciprocal refined once, the 3-tap
 * mean's division by 3 as a binary32 residual correction. Each form is compared with the expression it replaces: the two
 * one-operand forms over ALL 2^32 binary32 inputs, the division on 2^32 pseudo-random pairs of its domain. d_out (device,
 * 5 x unsigned long long): [0] differing thirds (must be 0), [1] inputs the third's guard rejects (2^24 + 1), [2] differing window
 * tests (must be 0), [3] differing quotients (must be 0), [4] quotients tested. */
NM_API int nm_selftest_orient(unsigned long long *d_out, void *stream);
/* Self-test of the matrix-pipe rounding premise under the matcher's proofs (no reference counterpart; what it protects is
 * the exact scan of kernels/match.cu:83-117, which match_finalize_kernel must reproduce from MFMA-screened candidates).
 * instruction: 0 = v_mfma_f32_32x32x16_bf16 (bf16x3 screen, every norm k-slot), 1 = v_mfma_f32_32x32x16_f16 (coarse pass of
 * the default two-stage screen). Runs, on the current device: the operand / accumulator layout probe, the directed rounding
 * cases, about n_random random single instructions, and about n_chains x 1024 accumulator chains issued exactly as the
 * screens issue them (one bf16 norm k-slot instruction, then 8 f16 or 24 bf16 instructions) on adversarial row families
 * incl. fp16-subnormal operands, each against binary64. d_out: NM_SELFTEST_MFMA_OUTPUTS floats on the device:
 *   [0] layout mismatches (must be 0)
 *   [1] C = 1 + 16 x 2^-25: (D - 1) in ulp (4 = dot product formed first)      [2] C = 1 + 2^-24 (1 + 2^-6): 1 = round to nearest
 *   [3] 1 + 15 x 2^-25, C = 0: ulp above 1                                      [4] C = 2^24 + 16 x 1: D - 2^24 (16 = summed first)
 *   [5] random instructions: max |D - exact| / (2^-24 (|C| + sum |a_k b_k|))
 *   [6] the same against the model: max |D - exact| / (2^-24 |exact| + 7 x 2^-24 (pmax_lo + pmax_hi)); <= 1 = inside the
 *       model whose DOUBLE the kernels' constants assume              [7] random instructions run
 *   [8] chains: max |value - exact of the same operand images| / (sqrt na + sqrt nb)^2 -- to be held against
 *       nm_sift_match_accum_budget(screen)                             [9] the same over the fp16-subnormal families only
 *   [10] chain launches of 1024 chains each                            [11] same-half truncation probe, ulp above 1 */
#define NM_SELFTEST_MFMA_OUTPUTS 16
NM_API int nm_selftest_mfma_model(int instruction, int n_random, int n_chains, float *d_out, void *stream);
/* The same for the fp32 instruction v_mfma_f32_32x32x2_f32 (the matcher's fp32 screen; the materialised distance pass of
 * nm_sift_match_f32): about n_random single results and n_chains x 1024 accumulator chains against binary64. d_out: 8 floats:
 *   [0] max |D - exact| / (2^-24 (|C| + |a0 b0| + |a1 b1|)) (<= 2 = two roundings: the premise of both bounds)
 *   [1] results equal to fma(a1, b1, fma(a0, b0, C))    [2] results equal to the correctly rounded exact sum    [5] results tested
 *   [3] the distance pass's form (two chains of 64 k-steps + one add): max |v - exact| / (sqrt na + sqrt nb)^2, to be held
 *       against nm_sift_match_distance_budget()    [4] the fp32 screen's form (one 130-step chain), against
 *       nm_sift_match_accum_budget(0) */
NM_API int nm_selftest_mfma_f32(int n_random, int n_chains, float *d_out, void *stream);
/* DIST_C of the MFMA distance pass (HOST function): |value - exact distance of the centred rows| <= DIST_C (sqrt nx + sqrt ny)^2
 * is what its acceptance test assumes; 67 + 8 roundings' worth with 15 % of slack. */
NM_API float nm_sift_match_distance_budget(void);
/* What the finalize pass budgets for the accumulation error of a screen's MFMA chain, as a multiple of
 * (sqrt na + sqrt nb)^2 (HOST function; screen: 0 = fp32, 1 = bf16x3, 2 = two-stage coarse pass): the share of
 * screen_err_coeff that is a hardware premise rather than arithmetic. The self-test's [8] must stay below half of it. */
NM_API float nm_sift_match_accum_budget(int screen);

/* ---- host-side scale-space constants ---- */
/* PyramidData::create_kernel_for_sigma (sift/pyramidata.cu:105-123). HOST function: writes 2*radius+1 normalised
 * taps to host memory `taps` (may be NULL to query) and returns radius = ceil(4*sigma).                          */
NM_API int nm_create_kernel_for_sigma(float sigma, float *taps);

/* ---- pyramid stages ---- */
/* convolve<float> (kernels/convolution.h:19-23, convolution.cu:141-159): zero-padded separable correlation,
 * `buffer` receives the row pass, `result` the column pass of `buffer`. `kernel` = 2r+1 taps in DEVICE memory.   */
NM_API int nm_convolve_f32(float *result, const float *image, float *buffer, int width, int height,
                           const float *kernel, int kernel_radius, void *stream);
/* downsample_by_2<float> (kernels/downsample.h:21-24, downsample.cu:20-29). */
NM_API int nm_downsample2_f32(float *result, int result_width, int result_height, const float *source,
                              int source_width, int source_height, void *stream);
/* subtract<float> (kernels/cudamath.h:56-60, cudamath.cu:57-67): C = A - B. */
NM_API int nm_subtract_f32(const float *A, const float *B, float *C, int width, int height, void *stream);
/* gradient<float> (kernels/cudamath.h:71-75, cudamath.cu:72-80): result is float2 (magnitude, angle in (0,2pi]).
 * Deviation: the 1-pixel border, which the reference never writes, is written as (0,0).            */
NM_API int nm_gradient_f32(const float *source, float *result, int width, int height, void *stream);

/* The loops of compute_dog (sift/siftfunctions.cu:42-51) and compute_gradients (:53-63) as ONE launch each: arrays of n
 * device plane pointers (host arrays), n <= 8 / n <= 3. Same arithmetic as nm_subtract_f32 / nm_gradient_f32. */
NM_API int nm_subtract_batch_f32(int n, const float *const *A, const float *const *B, float *const *C, int width,
                                 int height, void *stream);
NM_API int nm_gradient_batch_f32(int n, const float *const *source, float *const *result, int width, int height,
                                 void *stream);

/* ---- keypoints ---- */
/* find_keypoints, unmasked overload (kernels/keypoint.h:25-32, keypoint.cu:240-251). current/down/up are DoG
 * planes (were cudaTextureObject_t). `result` is the dense width*height float4 map, pre-filled with -1 by the
 * caller as in the reference (sift/siftfunctions.cu:120-121).                                                   */
NM_API int nm_find_keypoints_f32(const float *current, const float *down, const float *up, int width, int height,
                                 float peak_threshold, float edge_threshold, float xper, float sigma_0,
                                 int num_dogs, int dog, float *result, void *stream);
/* find_keypoints, masked overload (kernels/keypoint.h:52-59, keypoint.cu:226-237). `mask` is the full-resolution
 * mask plane mask_width x mask_height; it is sampled exactly as the reference's bilinear border texture fetch at
 * ((x+0.5)*xper, (y+0.5)*xper) (keypoint.cu:214).                                                                */
NM_API int nm_find_keypoints_masked_f32(const float *current, const float *mask, int mask_width, int mask_height,
                                        const float *down, const float *up, int width, int height,
                                        float peak_threshold, float edge_threshold, float xper, float sigma_0,
                                        int num_dogs, int dog, float *result, void *stream);
/* PyramidData::gpu_collate_keypoints_for_level's thrust::copy_if (sift/pyramidata.cu:84-88): stable compaction
 * of the entries with w >= 0 among the first num_pixels of `dense` into `out`; the count is written to the
 * DEVICE int *d_count. workspace: nm_compact_workspace_bytes(num_pixels) bytes of device scratch.               */
NM_API size_t nm_compact_workspace_bytes(int num_pixels);
NM_API int nm_compact_keypoints(const float *dense, int num_pixels, float *out, int *d_count, void *workspace,
                                void *stream);

/* The three find_keypoints calls of compute_keypoints[_with_mask] (sift/siftfunctions.cu:65-134) in ONE launch: dog[0..4]
 * are the octave's DoG planes, level l searches dog[l+1] between dog[l] and dog[l+2]; mask may be NULL. EVERY pixel of
 * the first width*height float4 of result[0..2] is written (the accepted keypoint or (-1,-1,-1,-1)), so only the part
 * of a map beyond width*height needs the reference's reset (siftfunctions.cu:120-121).                            */
NM_API int nm_find_keypoints3_f32(const float *const dog[5], const float *mask, int mask_width, int mask_height,
                                  int width, int height, float peak_threshold, float edge_threshold, float xper,
                                  float sigma_0, int num_dogs, float *const result[3], void *stream);
/* The same launch, which also resets entries [width * height, reset_end[l]) of result[l] to -1 (reset_end NULL: nothing):
 * compute_keypoints' per-octave reset of the dense maps (thrust::fill, sift/siftfunctions.cu:120-121) without launches of its
 * own -- the detection writes every entry of the octave's region, only what a larger octave left behind it needs the reset. */
NM_API int nm_find_keypoints3_reset_f32(const float *const dog[5], const float *mask, int mask_width, int mask_height,
                                        int width, int height, float peak_threshold, float edge_threshold, float xper,
                                        float sigma_0, int num_dogs, float *const result[3], const size_t reset_end[3],
                                        void *stream);
/* The frame driver's detection of one octave on caller-provided DoG planes: fused 3-level detection straight into ordered
 * lists, no dense maps (what nm_sift_detect_describe does per octave). The orchestration rules of compute_orientations /
 * compute_descriptors are applied: an empty level ends the octave (siftfunctions.cu:145,160), at most `capacity`
 * keypoints in total (:165-169). out: capacity float4, the levels' raster-ordered lists back to back; d_counts: 3 device
 * ints = keypoints kept per level. workspace: nm_find_keypoints3_compact_workspace_bytes(width, height) bytes.        */
NM_API size_t nm_find_keypoints3_compact_workspace_bytes(int width, int height);
NM_API int nm_find_keypoints3_compact_f32(const float *const dog[5], int width, int height, float peak_threshold,
                                          float edge_threshold, float xper, float sigma_0, int num_dogs, int capacity,
                                          float *out, int *d_counts, void *workspace, void *stream);
/* Tuning (extension, no counterpart in the reference): the batched detection launches of the frame driver take unit groups of 20
 * image rows instead of 5 when the launch still has at least `min_groups` of them (default 2048; results are identical
 * either way, tests/test_gpu_frame.py). min_groups < 0 restores the default; INT_MAX disables the tall form. Returns the
 * previous value. */
NM_API int nm_sift_set_detect_tall_min(int min_groups);
/* nm_compact_keypoints for three dense maps at once (three launches instead of nine); d_counts: 3 device ints. */
NM_API size_t nm_compact3_workspace_bytes(int num_pixels);
NM_API int nm_compact_keypoints3(const float *const dense[3], int num_pixels, float *const out[3], int *d_counts,
                                 void *workspace, void *stream);

/* ---- orientation + descriptor ---- */
/* detect_orientations (kernels/orientation.h:19-24, orientation.cu:219-230). `result` float2 per keypoint,
 * pre-filled with (-1,-1) by the caller (pyramidata.cu:90).                                                     */
NM_API int nm_detect_orientations(const float *key_pts, const float *grad, int num_pts, int octave_width,
                                  int octave_height, float gauss_factor, float xper, float *result, void *stream);
/* compute_sift_descriptors (kernels/descriptor.h:25-30, descriptor.cu:243-255). */
NM_API int nm_compute_sift_descriptors(const float *key_pts, const float *orients, const float *grad, int num_pts,
                                       int octave_width, int octave_height, int num_dogs, float xper, float *desc,
                                       float *x, float *y, void *stream);

/* detect_orientations / compute_sift_descriptors for the (up to three) level lists of one octave in ONE launch each: the
 * loops of compute_orientations / compute_descriptors (sift/siftfunctions.cu:136-181). Host arrays of n_levels device
 * pointers and counts. nm_detect_orientations_levels writes both components of every result (unset = -1): no pre-fill. */
NM_API int nm_detect_orientations_levels(int n_levels, const float *const *key_pts, const int *num_pts, const float *grad,
                                         int octave_width, int octave_height, float gauss_factor, float xper,
                                         float *const *result, void *stream);
NM_API int nm_compute_sift_descriptors_levels(int n_levels, const float *const *key_pts, const float *const *orients,
                                              const int *num_pts, const float *grad, int octave_width, int octave_height,
                                              int num_dogs, float xper, float *const *desc, float *const *x,
                                              float *const *y, void *stream);
/* Device-sized forms of the two (round 5): the three level counts of the octave are read ON THE DEVICE (d_counts: what
 * nm_compact_keypoints3 left there), so the host needs no count to issue them -- the reference's client synchronises once
 * per level for it (thrust::copy_if, sift/pyramidata.cu:84-91). An empty level ends the octave (siftfunctions.cu:145,160);
 * max_pts bounds a level's count (grid size; the lists hold at least as many entries). Descriptors go to the CONTAINER's
 * arrays: slot = running count + kept keypoints of the earlier levels + index, the running count read from *d_base_in (NULL:
 * host_base) and clipped at `capacity` (siftfunctions.cu:165-169); the new running count is written to *d_items_out. h_counts
 * (3 ints) / h_items (1 int): device-accessible pointers of pinned host memory that receive the raw counts / the new running
 * count, valid once `stream` has reached the launch (or NULL). nm/lazy_count.h is the C++ layer's use of them.            */
NM_API int nm_detect_orientations_levels_dev(const float *const *key_pts, const int *d_counts, int max_pts, const float *grad,
                                             int octave_width, int octave_height, float gauss_factor, float xper,
                                             float *const *result, int *h_counts, void *stream);
NM_API int nm_compute_sift_descriptors_levels_dev(const float *const *key_pts, const float *const *orients, const int *d_counts,
                                                  int max_pts, const int *d_base_in, int host_base, int capacity,
                                                  int *d_items_out, int *h_items, const float *grad, int octave_width,
                                                  int octave_height, int num_dogs, float xper, float *desc, float *x, float *y,
                                                  void *stream);

/* ---- matcher ---- */
/* transpose<float> (kernels/transpose.h:16-21, transpose.cu:33-40): odata[x*height+y] = idata[y*width+x]. */
NM_API int nm_transpose_f32(float *odata, const float *idata, int width, int height, void *stream);
/* compute_brute_force_distance<float> (kernels/match.h:18-23, match.cu:120-135): A is the TRANSPOSED query set
 * (dim x size_A), B is size_B x dim, result[j*size_A + i] = squared L2 distance. dim must be 128.               */
NM_API int nm_bf_distance_f32(const float *A, int size_A, const float *B, int size_B, int sift_vector_size,
                              float *result, void *stream);
/* get_sift_matches<float> (kernels/match.h:40-46, match.cu:141-150): per-row best / second best + ratio test. */
NM_API int nm_get_sift_matches_f32(const float *distance, int rows, int cols, int buffer_width, int *result,
                                   float ambiguity, void *stream);

/* compute_sift_matches (sift/siftfunctions.h:19-21, siftfunctions.cu:15-40) on raw descriptor arrays:
 * A: nA x 128, B: nB x 128 row-major. Fused MFMA path; `distance` (nA x nB) is optional (NULL = do not
 * materialise; extension). result[i] in {-1, 0..nB-1}, left untouched when the second-best distance is <= 0
 * (match.cu:107-116). Match decisions are made on distances recomputed exactly in the reference's summation
 * order; rows whose best two cannot be proven from the MFMA pass are re-scanned exactly. workspace:
 * nm_sift_match_workspace_bytes(nA, nB) bytes of device scratch.
 * Domain: ANY float input gives the reference scan's answer. The MFMA screens themselves work on descriptors whose
 * squared norms are finite and below 1e37; a query row outside that, and every row of a call in which ANY candidate is
 * outside it (NaN, +-inf, huge), is matched by the exact fallback alone, i.e. by the scan of match.cu:88-116 with its
 * behaviour on such values: a NaN distance to candidate 0 stays the row's minimum for good (the row becomes -1), a NaN
 * distance to any other candidate is skipped (`current < x` is false), and min2 is exact also above 2139095040 -- the scan
 * overwrites it with the old minimum at every replacement (:97), so its initial value 2139095040.0f only bounds a row whose
 * minimum sits at candidate 0. Such calls are correct, not fast (one exact 128-D distance per pair of rows).        */
NM_API size_t nm_sift_match_workspace_bytes(int nA, int nB);
/* How nm_sift_match_f32 fills a requested `distance` (process-wide; NM_MATCH_DISTANCE=exact|mfma sets the initial value):
 *   1 (default) = on the fp32 matrix cores, as the squared-norm expansion of the rows centred on the mean row of B
 *       (compute_brute_force_distance, kernels/match.cu:14-80, is the reference this replaces: 3 flop per (i, j, k) on scalar
 *       ALUs). EVERY entry is within 1e-4 relative of the reference's chain acc = fma(t, t, acc), t = a_k - b_k: an entry is
 *       kept only if a rigorous bound on the contraction's rounding error stays below that; all others
 *       -- near-duplicates, exact copies, cancellation, non-finite values -- are recomputed by the reference's own chain and
 *       are bit-equal to it (should the list of 32 x 32 blocks holding such entries overflow, the whole matrix is). The pass
 *       uses the same `workspace`.
 *   0 = the exact VALU kernel: every entry bit-equal to the reference's chain (what nm_bf_distance_f32, the building block,
 *       always computes).
 * result[] does not depend on the mode: match decisions are made on exactly recomputed distances either way.
 * nm_sift_match_distance_listed: diagnostics of the LAST such pass on `workspace` (same nA, nB): 32 x 32 blocks of the matrix
 * that held an entry outside the bound and were re-examined (-1 if the pass took the exact kernel for lack of scratch: fewer
 * than ~80 query rows), and the capacity of the block list (more listed than that = the whole matrix came from the exact
 * kernel). Synchronises `stream`. */
NM_API int nm_sift_match_set_distance_mode(int mode);
NM_API int nm_sift_match_get_distance_mode(void);
NM_API int nm_sift_match_distance_listed(const void *workspace, int nA, int nB, int *listed, int *capacity, void *stream);
/* Which MFMA screen the fused matcher runs before its exact finalize (process-wide; results are identical):
 * 0 = fp32 (v_mfma_f32_32x32x2_f32 on the descriptors themselves), 1 = bf16x3 (v_mfma_f32_32x32x16_bf16 on operands
 * split into two bf16 pieces: ~3x faster, a few more rows take the exact fallback), 2 = two-stage (a coarse pass with one
 * fp16 product per k on v_mfma_f32_32x32x16_f16 whose error is bounded per row from the norms of the fp16 rounding
 * residuals; rows it cannot prove -- about a percent on SIFT descriptors, every row when a squared norm reaches 1e9 -- are
 * screened again by the bf16x3 kernel; ~2x faster again). Default 2; the environment variable
 * NM_MATCH_SCREEN=f32|bf16x3|f16 sets the initial value. Extension: the reference has one (exact VALU) path. */
NM_API int nm_sift_match_set_screen(int screen);
NM_API int nm_sift_match_get_screen(void);
/* How many pairs ONE launch of the screening kernel covers in a batched call of n_pairs non-empty pairs under the current screen
 * (diagnostic: what a timing harness divides a launch's duration by). Round 6: when n_pairs is a multiple of the device's XCD count
 * every pair of a launch belongs to one XCD (its workgroups alone screen it, the XCDs work on 8 pairs side by side): the two-stage
 * screen's coarse pass is one launch for all n_pairs, the single-pass screens launch once per 8 pairs; otherwise the coarse pass
 * still covers all pairs and the single-pass screens launch once per pair. NM_COARSE_PAIR_XCD=0 / NM_TOP2_PAIR_XCD=0 keep the
 * divisions of rounds 3-5. Results never depend on the division. */
NM_API int nm_sift_match_pairs_per_launch(int n_pairs);
/* n <= 16 independent matches in one call (arrays of n): the norms, finalize and fallback launches cover all pairs at
 * once (the pair is a grid dimension), only the MFMA kernel runs once per pair -- 4 + n stream operations instead of
 * 5 n. result[k] as in nm_sift_match_f32 (no distance matrices). workspace: nm_sift_match_batch_workspace_bytes bytes
 * (the per-pair bounds laid end to end, in order). */
#define NM_SIFT_MATCH_MAX_BATCH 16
NM_API size_t nm_sift_match_batch_workspace_bytes(int n, const int *nA, const int *nB);
NM_API int nm_sift_match_batch_f32(int n, const float *const *A, const int *nA, const float *const *B, const int *nB,
                                   int *const *result, float ambiguity, void *workspace, void *stream);
/* The same with the set sizes read from DEVICE memory: d_nA[k] / d_nB[k] point at the int the frame driver wrote
 * (d_num_items of nm_sift_detect_describe[_batch]; the reference keeps SiftData::_num_items on the host after one
 * synchronisation per level, sift/siftfunctions.cu:165-178, and compute_sift_matches reads it there, :15-40). A live
 * client therefore chains detect -> match on a stream with no host read-back, and a HIP graph captured over both replays
 * correctly on frames with different keypoint counts. capA / capB bound the sizes (values above them are clipped, as the
 * frame driver clips at its capacity; a size <= 0 makes the pair a no-op): every grid and the workspace are laid out for
 * them, and the work plan for the real sizes is made on the device. Results are identical to the host-sized entry called
 * with the same sizes. workspace: nm_sift_match_batch_dev_workspace_bytes(n, capA, capB) bytes.
 * nm_sift_match_fallback_count(workspace + k * (that / n), capA, capB, ...) reads pair k's fallback count. */
NM_API size_t nm_sift_match_batch_dev_workspace_bytes(int n, int capA, int capB);
NM_API int nm_sift_match_batch_dev_f32(int n, const float *const *A, const int *const *d_nA, const float *const *B,
                                       const int *const *d_nB, int capA, int capB, int *const *result, float ambiguity,
                                       void *workspace, void *stream);
/* The same call cut into its three dependent phases, for clients that pipeline several calls over streams of their own:
 * PREP (norms, split operand images, the device-side work plan: streaming, ~5 us per pair), SCREEN (the MFMA kernel, one
 * launch per pair: it fills the chip by itself) and FINISH (exact finalize, fallback, merge: latency-bound gathers that
 * leave most of the chip idle). The phases of one call must run in this order on the same workspace -- the CLIENT orders
 * them with its events -- but PREP of call k + 1 and FINISH of call k - 1 may run on a second stream beside SCREEN of call k
 * (bench.py does that: the MFMA launches stay back to back on one stream). phases = 7 is nm_sift_match_batch_dev_f32. */
#define NM_MATCH_PHASE_PREP 1
#define NM_MATCH_PHASE_SCREEN 2
#define NM_MATCH_PHASE_FINISH 4
NM_API int nm_sift_match_batch_dev_phases_f32(int phases, int n, const float *const *A, const int *const *d_nA,
                                              const float *const *B, const int *const *d_nB, int capA, int capB,
                                              int *const *result, float ambiguity, void *workspace, void *stream);
/* HOST functions (no device access; on a box without a GPU the MI355X geometry of 256 CUs / 8 XCDs is assumed): the work
 * distribution the matcher uses for (nA, nB). plan[0..9] = query blocks of 256 rows, candidate tiles of 128 rows, persistent
 * workgroups G, partial lists per query S (<= 64, what the workspace bound assumes), XCD groups X, workgroups per group,
 * tiles per chunk Tc, chunks C, query blocks per group (base, and how many groups hold one more). Units = blocks x tiles;
 * group x = workgroups {x, x + X, ...} holds a contiguous share of the query blocks and walks its units chunk by chunk,
 * query block by query block inside a chunk, so that the workgroups of one XCD stream the same candidate tiles together.
 * nm_sift_match_plan_segments lists what workgroup `wg` does, in order: rows of 5 ints (query block, first tile, number of
 * tiles, partial-list slot, 1 if the segment completes its query block); returns the number of segments. For tests and
 * capacity planning. Sets of 2^22 (4 194 304) rows or more are outside the matcher's domain (32-bit byte offsets and unit
 * indices): nm_sift_match_plan returns an error status, nm_sift_match_plan_segments -1, and every matching entry point
 * returns an error status before any plan is made. */
NM_API int nm_sift_match_plan(int nA, int nB, int plan[10]);
NM_API int nm_sift_match_plan_segments(int nA, int nB, int wg, int *segments, int max_segments);
NM_API int nm_sift_match_f32(const float *A, int nA, const float *B, int nB, float *distance, int *result,
                             float ambiguity, void *workspace, void *stream);
/* Diagnostics: number of query rows of the LAST nm_sift_match_f32 / _shard_f32 call on `workspace` (same nA, nB) that took
 * the exact full-scan fallback because the MFMA candidate pass could not prove its top-2. Synchronises the stream. */
NM_API int nm_sift_match_fallback_count(const void *workspace, int nA, int nB, int *host_count, void *stream);
/* The same for the two-stage screen: rows whose coarse (fp16) result could not be proven and that the bf16x3 pass screened
 * again. Meaningful after a call under screen 2 only (the counter is reset by such calls). Synchronises the stream. */
NM_API int nm_sift_match_second_pass_count(const void *workspace, int nA, int nB, int *host_count, void *stream);
/* Multi-GPU building blocks (no reference counterpart: the reference is single-GPU). A shard call scans the local
 * rows [0,nB_shard) of B and emits, per query row: min1 = the smallest non-NaN distance of the shard, index1 +
 * index_offset = its lowest index (-1 if the shard has no distance below +inf), min2 = the smallest of the shard's OTHER
 * non-NaN distances, +inf if there is none -- NOT clamped to the scan's initial 2139095040.0f, which only bounds a row
 * whose minimum sits at global candidate 0 (match.cu:91,97) and is applied by the merge. The shard that holds global
 * candidate 0 (index_offset == 0) reports (NaN, 0, smallest other) for a row whose distance to that candidate is NaN, as
 * the scan keeps it. The merge combines n_shards such triples per row, given shard-major (n_shards x nA), in ascending
 * shard order so that the lowest global index wins ties, and applies the clamp and the ratio test: the result equals
 * the single-GPU scan for every input. */
NM_API int nm_sift_match_shard_f32(const float *A, int nA, const float *B_shard, int nB_shard, int index_offset,
                                   float *min1, int *idx1, float *min2, void *workspace, void *stream);
NM_API int nm_sift_match_merge_f32(const float *min1, const int *idx1, const float *min2, int n_shards, int nA,
                                   int *result, float ambiguity, void *stream);

/* The whole sharded match as ONE call per rank (native counterpart of niftymatch_amd/parallel.py::match_sharded): every
 * rank holds all of A and rows [index_offset, index_offset + nB_shard) of B; the call computes the shard triples, issues
 * ONE ncclAllGather of 3 * nA int32 per rank (12 B per row: latency-bound, far below the per-link xGMI bandwidth) on
 * `stream`, and merges the n_ranks triples in ascending rank order so that the lowest global index wins ties
 * (match.cu:94-105). nccl_comm is the caller's ncclComm_t (RCCL), passed as void* to keep this header free of rccl.h;
 * ncclAllGather is resolved from the running process, libnm_hip.so does not link librccl. n_ranks = 1 needs no
 * communicator. Shards must be passed in rank order: rank g's index_offset = rows of B held by ranks < g.
 * workspace: nm_sift_match_allgather_workspace_bytes(nA, nB_shard, n_ranks) bytes. No reference counterpart (single-GPU). */
/* The merge step alone, on the buffer the all-gather produces: rank g's block is (min1[nA], idx1[nA], min2[nA]) as int32
 * bit patterns at packed + g * 3 * nA. Ranks in ascending order = ascending global candidate index (Q14). */
NM_API int nm_sift_match_merge_packed_f32(const int *packed, int n_shards, int nA, int *result, float ambiguity,
                                          void *stream);
NM_API size_t nm_sift_match_allgather_workspace_bytes(int nA, int nB_shard, int n_ranks);
NM_API int nm_sift_match_allgather_f32(const float *A, int nA, const float *B_shard, int nB_shard, int index_offset,
                                       int n_ranks, int *result, float ambiguity, void *workspace, void *nccl_comm,
                                       void *stream);

/* ---- "next" rows of SURVEY.md 8(f): the element-wise stages either side of the path ---- */
/* cuda_grayscale<float> (kernels/bgra_2_gray.h:14-18, bgra_2_gray.cu:9-31): 0.07 B + 0.72 G + 0.21 R. bgra = uchar4. */
NM_API int nm_grayscale_f32(const unsigned char *bgra, float *output, int width, int height, void *stream);
/* cuda_extract_channel<float> (bgra_2_gray.cu:36-62): channel 0..3 = B, G, R, A. */
NM_API int nm_extract_channel_f32(const unsigned char *bgra, float *output, int width, int height, int channel,
                                  void *stream);
/* cuda_put_channel<float> (bgra_2_gray.cu:65-92): channel 3 writes 255, as the reference does. */
NM_API int nm_put_channel_f32(unsigned char *bgra, const float *input, int width, int height, int channel, void *stream);
/* cuda_set_alpha_to_const (bgra_2_gray.cu:95-112). */
NM_API int nm_set_alpha_to_const(unsigned char *bgra, int width, int height, unsigned char val, void *stream);
/* cuda_cast<float, unsigned char> (kernels/cast.h:17-22, cast.cu:8-39): saturates at max_val when max_val != 0. */
NM_API int nm_cast_f32_u8(const float *src, size_t cols, size_t rows, unsigned char *dst, unsigned char max_val,
                          void *stream);
/* downsample_by_2<uchar4> (kernels/downsample.cu:32). */
NM_API int nm_downsample2_u8x4(unsigned char *result, int result_width, int result_height, const unsigned char *source,
                               int source_width, int source_height, void *stream);
/* align_points (kernels/ransac.h:8-10, ransac.cu:29-59): gathers the matched coordinates; unmatched rows get -1. */
NM_API int nm_align_points(const float *src_x, const float *src_y, const float *dst_x, const float *dst_y,
                           float *c_src_x, float *c_src_y, float *c_dst_x, float *c_dst_y, const int *matches,
                           int num_pts, void *stream);

/* ---- N3/N4 warps. The reference samples cudaTextureObject_t's made by CudaTex2D (utils/cudatex2D.cu:13-19: border
 * addressing, linear filter, unnormalised coordinates, normalised-float reads). Here a texture is a device plane
 * (pointer, width, height, texel format); the filter is done in software with the texture unit's 1/256 weights.   */
#define NM_TEX_U8N 0      /* unsigned char, read as c/255          (cudaReadModeNormalizedFloat) */
#define NM_TEX_U8X4N 1    /* uchar4, each channel read as c/255 */
#define NM_TEX_F32 2      /* float, read as is                      (cudaReadModeElementType) */
/* cuda_undistort (kernels/undistort.h:30-35, undistort.cu:7-64): camera_matrix = fx, fy, cx, cy; distortion_coeffs =
 * k1, k2, k3 (device pointers, as in the reference). */
NM_API int nm_undistort_map_f32(const float *x, const float *y, size_t cols, size_t rows, const float *camera_matrix,
                                const float *distortion_coeffs, float *u, float *v, void *stream);
/* resample_undistort (kernels/resample.h:34-38, resample.cu:100-113,234-248): tex(x+.5, y+.5) * 255.9999f */
NM_API int nm_resample_undistort_f32(const void *tex, int tex_width, int tex_height, int tex_format, const float *x,
                                     const float *y, size_t cols, size_t rows, float *undistorted, void *stream);
/* resample_mask (resample.h:12-14, resample.cu:69-82,207-216) */
NM_API int nm_resample_mask_u8(unsigned char *result, const void *tex, int tex_width, int tex_height, int tex_format,
                               int cols, int rows, const float *x_pos, const float *y_pos, float threshold, void *stream);
/* resample_perspective_transform (resample.h:7-10, resample.cu:84-98,116-204): fills x_pos / y_pos with the (inverse)
 * projective map of the pixel grid and samples the uchar4 texture there; one launch instead of two. */
NM_API int nm_resample_perspective_u8x4(unsigned char *result, const unsigned char *tex, int tex_width, int tex_height,
                                        int cols, int rows, float *x_pos, float *y_pos, const float *mat3x3, int inverse,
                                        void *stream);
/* transform_blend (resample.h:16-20, resample.cu:7-66,218-232): frame (uchar4), frame_mask and frame_wts are fw x fh. */
NM_API int nm_transform_blend(unsigned char *canvas, int cw, int ch, const unsigned char *frame, int fw, int fh, int nw,
                              int nh, const float *mat3x3, int tx, int ty, const void *frame_mask, int mask_format,
                              float *canvas_wts, const void *frame_wts, int wts_format, void *stream);

/* RANSAC hypothesis evaluation (kernels/ransac.cu:430-521 kernels + the max_element/copy of :523-694).
 * model: 0 translation (1 sample per hypothesis), 1 similarity (2), 2 homography (4). rand_list: DEVICE array of
 * iterations * samples point indices (the reference draws them on the host, ransac.cu:543-551; see the C++ wrappers).
 * Outputs (device): homographies iterations x 9 (zeros for a hypothesis with a repeated index), inliers iterations,
 * H_best 9 floats = the hypothesis at the FIRST maximum of inliers, *d_position its index (may be NULL).
 * Points with src_x < 0 (unmatched rows of align_points) are ignored. The null vector of the design matrix is found by
 * an independent one-sided Jacobi, not the reference's GSL port: see csrc/nm_ransac_math.hpp.                     */
NM_API int nm_ransac_f32(int model, const float *src_x, const float *src_y, const float *dst_x, const float *dst_y,
                         int num_pts, const int *rand_list, int iterations, float inlier_threshold,
                         float *homographies, int *inliers, float *H_best, int *d_position, void *stream);

/* Seed of the host-side sampler used by the C++ ransac_* wrappers (0 = std::random_device, the reference's behaviour). */
NM_API void nm_ransac_seed(unsigned int seed);

/* ---- per-frame driver ---- */
/* The per-octave client loop the reference leaves to its caller, run entirely on `stream` with no
 * host synchronisation and no allocation: Gaussian pyramid + DoG + gradients + extrema + ordered compaction +
 * orientations + descriptors of one grayscale fp32 frame. An arena owns every intermediate buffer
 * (replaces PyramidData, sift/pyramidata.cu:24-50) and is bound to one stream at a time.                        */
typedef struct nm_sift_arena nm_sift_arena;
NM_API int nm_sift_arena_create(int width, int height, int capacity, nm_sift_arena **arena);
NM_API void nm_sift_arena_destroy(nm_sift_arena *arena);
NM_API size_t nm_sift_arena_bytes(const nm_sift_arena *arena);
/* The reference's run-time knobs for the calls that follow on this arena. SiftParams::_peak_threshold and
 * _edge_threshold are public fields a client sets between frames (sift/siftparams.h:97-98; read per call by
 * compute_keypoints, sift/siftfunctions.cu:123-125): defaults 0 and 10. edge_threshold must be > 0. All arenas of one
 * batched call must hold the same pair. Host-side setters (no stream operation): they take effect for the calls ENQUEUED
 * afterwards; a HIP graph captured earlier keeps the values it was captured with.                                 */
NM_API int nm_sift_arena_set_params(nm_sift_arena *arena, float peak_threshold, float edge_threshold);
NM_API int nm_sift_arena_get_params(const nm_sift_arena *arena, float *peak_threshold, float *edge_threshold);
/* compute_keypoints_with_mask (sift/siftfunctions.cu:65-98, kernels/keypoint.cu:204-224): detection only where the
 * full-resolution mask, fetched with the reference's bilinear border texture at ((x+0.5) xper, (y+0.5) xper), is >= 1.
 * mask: caller-owned DEVICE plane of exactly the arena's width x height floats, read by every later call until it is
 * replaced; NULL removes it. Per arena: the frames of a batched call may have different masks or none.               */
NM_API int nm_sift_arena_set_mask(nm_sift_arena *arena, const float *mask, int mask_width, int mask_height);
/* Diagnostics of the octave-tail launch (csrc/nm_tail.hip; the octaves >= 2 of a call run as ONE persistent launch whose
 * work items draw tickets): the plan's segments (5 ints each: kind 0 = levels 1..3 / whole plane, 1 = levels 4..5, 2 = detect,
 * 3 = scan + gather, 4 = gradient planes of a whole octave; slot; items per frame; first item per frame; octave) and, when the
 * arena was created under NM_TAIL_TRACE=1 and was the first arena of the last call, 4 words per item of that launch (meta,
 * clock at ticket, at inputs ready, at done; 100 MHz). Returns the items per frame (0: this geometry takes the per-octave
 * launches; < 0: error). Synchronises the device when `out` is given. */
NM_API int nm_sift_arena_tail_trace(const nm_sift_arena *arena, unsigned long long *out, int max_items, int *segments,
                                    int max_segments);
NM_API int nm_sift_arena_tail_segments(const nm_sift_arena *arena);
/* Failure reporting of the octave-tail launch. Every wait inside it is bounded (~2 s); a wait that times out sets a sticky
 * error word, the remaining work items of THAT launch are drained without working, and the launch leaves its state clean for
 * the next call. What the caller sees: d_num_items of every frame of that call reads -1 (outputs of that call are invalid;
 * the device-sized matcher entries treat a negative size as 0), and nm_sift_arena_tail_status -- on the first arena of the
 * call, whose state words the launch used -- writes *status = 1 (0 after a complete launch; synchronises `stream`). The
 * call after a failed one is unaffected. nm_sift_arena_tail_inject_error is a TEST HOOK that sets the sticky word as a
 * timed-out wait would (synchronises the device).                                                                       */
NM_API int nm_sift_arena_tail_status(const nm_sift_arena *arena, int *status, void *stream);
NM_API int nm_sift_arena_tail_inject_error(nm_sift_arena *arena);
/* HOST function: the number of kernel launches one nm_sift_detect_describe[_batch] call of n frames on this arena issues
 * (1080p: 23 for n <= 2, where the octave tail is used; 51 above). */
NM_API int nm_sift_arena_launches_per_call(const nm_sift_arena *arena, int n);
/* HOST function (no device access): the octave-tail plan for a width x height frame whose tail starts at octave T (the frame
 * driver uses T = 2; NM_FRAME_TAIL=1..3 in the environment when an arena is created). segments: 8 ints each (kind, slot, items
 * per frame, first item per frame, octave, 1 if the whole plane is one item, octave width, octave height), in the order in
 * which the launch's tickets run through them -- an item only ever waits for items of EARLIER segments; info: items per
 * frame, LDS bytes of the tail launch, of the scan launch, number of tail octaves. Returns the number of segments; 0 when the
 * geometry is not covered (too few octaves, radii other than the SIFT defaults, LDS) and takes the per-octave launches. */
NM_API int nm_sift_tail_plan(int width, int height, int T, int *segments, int max_segments, int info[4]);
/* gray: width*height fp32 on the device. Outputs on the device: desc capacity x 128, x,y capacity (full-resolution
 * coordinates, descriptor.cu:75-77), d_num_items = number of descriptors written (<= capacity,
 * siftfunctions.cu:165-169). kpts (capacity float4) and orients (capacity float2) are optional (NULL).        */
NM_API int nm_sift_detect_describe(nm_sift_arena *arena, const float *gray, float *desc, float *x, float *y,
                                   float *kpts, float *orients, int *d_num_items, void *stream);
/* The same for n <= NM_SIFT_MAX_BATCH frames in ONE launch sequence (arrays of n pointers; the arenas must be distinct
 * and of equal width, height and capacity; kpts, orients, d_num_items may be NULL or hold NULLs). EVERY launch covers
 * all frames of the call (the frame is a grid dimension): a single 1080p frame is only ~4 workgroups per CU at octave
 * 0 and its higher octaves are pure launch latency, a batch fills the chip and shares the ~60 launches. Results are
 * identical to n separate nm_sift_detect_describe calls. No reference counterpart (the reference processes one frame
 * per call sequence, sift/siftfunctions.cu:42-181).
 * STREAM CAPTURE: the call uses events only (no synchronisation, no allocation) and can be captured into a HIP graph on
 * `stream`. Internally it forks the first arena's side stream(s) off `stream` and joins EVERY forked stream back into
 * `stream` directly before it returns (also on an error path). A client that captures around these calls must follow the
 * same rule for its own streams: on ROCm 7.2 a forked stream that is joined into ANOTHER forked stream (s1 -> s2 -> s3,
 * then s3 -> s2 -> s1) crashes the capturing process inside hipStreamEndCapture / instantiate; s1 -> s2, s1 -> s3 or
 * s1 -> s2 -> s3 with s2 -> s1 and s3 -> s1 are safe (tools/capture_shapes.py reproduces both). The same holds for
 * nm_sift_match_batch[_dev]_f32, which run on `stream` alone.                                                           */
#define NM_SIFT_MAX_BATCH 64
NM_API int nm_sift_detect_describe_batch(nm_sift_arena *const *arenas, int n, const float *const *gray,
                                         float *const *desc, float *const *x, float *const *y, float *const *kpts,
                                         float *const *orients, int *const *d_num_items, void *stream);
/* Measurement aid: only the scale-space launches of nm_sift_detect_describe_batch (base blur + every octave's Gaussian
 * levels, DoG planes, gradient planes and decimation), on `stream`, no detection. The planes it leaves in the arenas
 * are those of the last octave. bench.py times this sequence for the whole-pyramid roofline (SURVEY.md 8(d):
 * 108 B per octave-pixel + 36 B for the gradients).                                                              */
NM_API int nm_sift_scale_space_batch(nm_sift_arena *const *arenas, int n, const float *const *gray, void *stream);
/* The same chain with (write_dog = 1, what nm_sift_scale_space_batch does: levels + DoG + gradient planes, the 108 B per
 * octave-pixel yardstick) or without (0) materialised DoG planes. nm_sift_detect_describe[_batch] runs the latter: its
 * detection kernel subtracts consecutive levels itself, so 16 of the chain's 64 written bytes per pixel are never moved.
 * write_dog | 2: the same without the fused gradient planes, i.e. with write_dog = 3 exactly the work of the reference's
 * convolve + compute_dog loops (6 level writes + 5 DoG planes per octave: the 108 B per octave-pixel yardstick alone). */
NM_API int nm_sift_scale_space_batch_ex(nm_sift_arena *const *arenas, int n, const float *const *gray, int write_dog,
                                        void *stream);
/* Pointers into the arena for stage-level inspection (tests, profiling): Gaussian level l (0..5) and DoG d (0..4)
 * planes of the LAST processed octave geometry are overwritten per octave, so these are meaningful only after
 * nm_sift_octave_pyramid().                                                                                      */
NM_API float *nm_sift_arena_level(nm_sift_arena *arena, int level);
NM_API float *nm_sift_arena_dog(nm_sift_arena *arena, int dog);
NM_API float *nm_sift_arena_grad(nm_sift_arena *arena);
/* Gaussian levels 1..5 + DoG 0..4 + gradients 0..2 of one octave whose level 0 already sits in
 * nm_sift_arena_level(arena,0) with geometry ow x oh (the fused pyramid stage, timed by bench.py).            */
NM_API int nm_sift_octave_pyramid(nm_sift_arena *arena, int ow, int oh, void *stream);

#ifdef __cplusplus
}
#endif
#endif
