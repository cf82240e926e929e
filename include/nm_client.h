/*
 * include/nm_client.h -- a reference-style CLIENT of the drop-in C++ API, exported with C linkage so that tests can
 * drive the headers under niftymatch_amd/nm (SiftParams, PyramidData, SiftData, convolve, downsample_by_2, compute_dog,
 * compute_gradients, compute_keypoints, compute_orientations, compute_descriptors, compute_sift_matches) exactly the
 * way an application written against NiftyMatch would (the per-octave loop of SURVEY.md 3.1). Host pointers in/out.
 */
#ifndef NM_CLIENT_H
#define NM_CLIENT_H
#ifdef __cplusplus
extern "C" {
#endif
/* gray: width*height fp32 (host). desc: capacity*128, x/y: capacity (host). Returns the number of descriptors or a
 * negative value on a C++ exception. */
__attribute__((visibility("default"))) int nm_client_detect_describe(const float *gray, int width, int height,
                                                                    int capacity, float *desc, float *x, float *y);
/* Wall-clock microseconds per frame pair of the reference-style client loop (2 x detect+describe with the per-octave
 * API + compute_sift_matches), `reps` pairs after one warm-up, objects created once. gray0/gray1: DEVICE planes.
 * with_distance: pass the reference's mandatory N x M distance buffer (1) or NULL (0, extension). n_out (3 ints, host,
 * may be NULL): keypoints of frame 0, of frame 1, matches found. Negative on error. */
__attribute__((visibility("default"))) double nm_client_pair_loop(const float *gray0, const float *gray1, int width,
                                                                 int height, int capacity, int reps, int with_distance,
                                                                 int *n_out);
/* The same with streams = 2: the two frames of a pair driven concurrently from two host threads on two streams (a
 * PyramidData each), the match on the first stream once both are described. streams = 1 is nm_client_pair_loop. */
__attribute__((visibility("default"))) double nm_client_pair_loop_ex(const float *gray0, const float *gray1, int width,
                                                                    int height, int capacity, int reps, int with_distance,
                                                                    int streams, int *n_out);
/* Copies, assignments, moves and std::vector growth of PyramidData / SiftData, each checked by running the frame through the
 * object. gray: width*height fp32 (host). Returns the descriptor count when all variants agree, -2 on a mismatch, -1 on an
 * exception (a double free aborts the process). */
__attribute__((visibility("default"))) int nm_client_copy_semantics(const float *gray, int width, int height, int capacity);
/* Lazy counts (niftymatch_amd/nm/lazy_count.h) against the reference's observable state: the frame through the per-octave loop
 * (a) without looking at a count before the end, (b) looking at _orientations[l].size() / _num_items after every call, (c) with
 * one synchronisation per octave (NM_EAGER_COUNTS). watch: 4 ints per octave (three sizes, running item count) as (b) saw
 * them -- identical to (c)'s or the call returns -2; pending_seen: how many of (b)'s looks resolved a pending value. Returns
 * the item count (all three runs agree on counts, descriptors and coordinates), -2 on a mismatch, -1 on an exception. */
__attribute__((visibility("default"))) int nm_client_lazy_counts(const float *gray, int width, int height, int capacity,
                                                                int *watch, int max_octaves, int *pending_seen);
/* A: nA*128, B: nB*128 (host). distance: nA*nB (host) or NULL. result: nA ints, pre-filled by the caller. */
__attribute__((visibility("default"))) int nm_client_match(const float *A, int nA, const float *B, int nB,
                                                          float *distance, int *result, float ambiguity);
/* model 0/1/2 = ransac_translation / ransac_similarity / ransac_homography of the C++ API with the given sampler seed.
 * Host arrays in, H (9 floats, host) out. Returns 1 when the fit ran, 0 when there were too few points, -1 on error. */
__attribute__((visibility("default"))) int nm_client_ransac(int model, const float *src_x, const float *src_y,
                                                           const float *dst_x, const float *dst_y, int n,
                                                           float inlier_threshold, int iterations, unsigned int seed,
                                                           float *H);
#ifdef __cplusplus
}
#endif
#endif
