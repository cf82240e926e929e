// siftfunctions.cpp -- host orchestration (reference: src/gpu/sift/siftfunctions.cu:15-181) on the C ABI / L2 launchers.
#include "../siftfunctions.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>

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
// The scratch lives in A (grow-only): nothing is allocated per call.
// Ordering: the reference returns only after the device has drained (its two thrust temporaries are freed on return,
// which synchronises), so a client may read A->_match_indexes from the host or from another stream right after the
// call. That is kept by default: the call ends with hipStreamSynchronize(stream). NM_ASYNC_MATCHES=1 in the environment
// (or nm_set_async_matches(1)) makes it asynchronous on `stream` for clients that order their reads themselves.
// Under stream capture a synchronise would invalidate the capture, so a capturing stream is never synchronised: the call
// is then recorded asynchronously, and the graph's own ordering takes the place of the drain.
// The flag is an atomic: clients drive compute_sift_matches from several host threads (nm_client_pair_loop_ex does).
static std::atomic<int> g_async_matches{-1};
extern "C" __attribute__((visibility("default"))) void nm_set_async_matches(int on) { g_async_matches.store(on ? 1 : 0); }
static bool async_matches()
{
    int v = g_async_matches.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("NM_ASYNC_MATCHES");
        int expected = -1;
        g_async_matches.compare_exchange_strong(expected, (e && e[0] == '1') ? 1 : 0);   // a concurrent nm_set_async_matches wins
        v = g_async_matches.load(std::memory_order_relaxed);
    }
    return v == 1;
}
static bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Lazy counts (nm/lazy_count.h): NM_EAGER_COUNTS=1 (or nm_set_eager_counts(1)) restores one host synchronisation per octave.
static std::atomic<int> g_eager_counts{-1};
extern "C" __attribute__((visibility("default"))) void nm_set_eager_counts(int on) { g_eager_counts.store(on ? 1 : 0); }
static bool eager_counts()
{
    int v = g_eager_counts.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("NM_EAGER_COUNTS");
        int expected = -1;
        g_eager_counts.compare_exchange_strong(expected, (e && e[0] == '1') ? 1 : 0);
        v = g_eager_counts.load(std::memory_order_relaxed);
    }
    return v == 1;
}

// The device word that holds a container's item count for a device-sized launch: the pending count's own word, or -- when
// the host value is the authority -- that value written into the container's spare word on `stream`.
static const int *items_on_device(SiftData *S, hipStream_t stream)
{
    if (S->_items_dev.size() < 2) S->_items_dev = nm::device_vector<int>(2);
    if (S->_num_items.pending()) return S->_items_dev.data() + S->_items_cur;
    int *w = S->_items_dev.data() + (S->_items_cur ^ 1);
    nm_check(nm_fill_u32(w, 1, (unsigned)S->_num_items.host_value(), stream), "item count upload");
    return w;
}

void compute_sift_matches(SiftData *A, SiftData *B, float *distance, float ambiguity, hipStream_t stream)
{
    if (!distance && (A->_num_items.pending() || B->_num_items.pending()) && A->_capacity > 0 && B->_capacity > 0) {
        // Neither count has been read back (lazy_count.h) and no matrix is asked for: the matcher reads both sizes on the
        // device (nm_sift_match_batch_dev_f32) -- no host synchronisation between compute_descriptors and the match.
        const size_t need = nm_sift_match_batch_dev_workspace_bytes(1, A->_capacity, B->_capacity) / sizeof(int) + 1;
        if (A->_match_workspace.size() < need) {
            if (stream_is_capturing(stream)) RUNTIME_EXCEPTION("compute_sift_matches: the match workspace cannot grow under stream capture");
            nm_check((int)hipStreamSynchronize(stream), "SIFT matching failed");
            A->_match_workspace = nm::device_vector<int>();
            A->_match_workspace = nm::device_vector<int>(need);
        }
        const float *pa = A->_desc.data(), *pb = B->_desc.data();
        const int *na = items_on_device(A, stream), *nb = items_on_device(B, stream);
        int *res = A->_match_indexes.data();
        nm_check(nm_sift_match_batch_dev_f32(1, &pa, &na, &pb, &nb, A->_capacity, B->_capacity, &res, ambiguity,
                                             A->_match_workspace.data(), stream),
                 "SIFT matching failed");
        if (!async_matches() && !stream_is_capturing(stream)) nm_check((int)hipStreamSynchronize(stream), "SIFT matching failed");
        return;
    }
    const int A_size = A->_num_items;
    const int B_size = B->_num_items;
    if (A_size <= 0 || B_size <= 0) return;
    const size_t need = nm_sift_match_workspace_bytes(A_size, B_size) / sizeof(int) + 1;
    if (A->_match_workspace.size() < need) {
        // sized for the container's capacity so that later calls with more keypoints do not grow it again
        const int cap_a = A->_capacity > A_size ? A->_capacity : A_size, cap_b = B->_capacity > B_size ? B->_capacity : B_size;
        // an earlier match may still use the old one. (Under capture nothing has run yet and growing is not possible
        // without a synchronise: size A's workspace with one eager call before capturing, as INTEGRATION.md says.)
        if (stream_is_capturing(stream)) RUNTIME_EXCEPTION("compute_sift_matches: the match workspace cannot grow under stream capture");
        nm_check((int)hipStreamSynchronize(stream), "SIFT matching failed");
        A->_match_workspace = nm::device_vector<int>();
        // exact size (resize_uninitialized would add its 50 % growth headroom to what is already the capacity's bound)
        A->_match_workspace = nm::device_vector<int>(nm_sift_match_workspace_bytes(cap_a, cap_b) / sizeof(int) + 1);
    }
    nm_check(nm_sift_match_f32(A->_desc.data(), A_size, B->_desc.data(), B_size, distance, A->_match_indexes.data(),
                               ambiguity, A->_match_workspace.data(), stream),
             "SIFT matching failed");
    if (!async_matches() && !stream_is_capturing(stream)) nm_check((int)hipStreamSynchronize(stream), "SIFT matching failed");
}

void compute_dog(PyramidData &pydata, const int octave_width, const int octave_height, hipStream_t stream)
{
    const float *a[19], *b[19];
    float *c[19];
    const int n = pydata._num_dogs;
    for (int i = 0; i < n; ++i) { a[i] = pydata._octave[i + 1].data(); b[i] = pydata._octave[i].data(); c[i] = pydata._dog[i].data(); }
    for (int i = 0; i < n; i += 8)         // one launch per 8 planes (5 in the SIFT configuration)
        nm_check(nm_subtract_batch_f32(n - i < 8 ? n - i : 8, a + i, b + i, c + i, octave_width, octave_height, stream),
                 "Subtract launch failed");
}

void compute_gradients(PyramidData &pydata, const SiftParams &params, const int octave_width, const int octave_height,
                       hipStream_t stream)
{
    float2 *g = pydata._grad.data();
    const size_t offset = (size_t)octave_width * octave_height;
    const float *src[19];
    float *dst[19];
    int n = 0;
    for (int i = params._level_min + 1; i <= params._level_max - 2; ++i, ++n) {
        src[n] = pydata._octave[i + 1].data();
        dst[n] = reinterpret_cast<float *>(g + i * offset);
    }
    for (int i = 0; i < n; i += 3)
        nm_check(nm_gradient_batch_f32(n - i < 3 ? n - i : 3, src + i, dst + i, octave_width, octave_height, stream),
                 "Set gradient launch failed");
}

static void keypoints_impl(PyramidData &pydata, const SiftParams &params, const float *mask, int mask_w, int mask_h,
                           const int octave, const int ow, const int oh, hipStream_t stream)
{
    const float xper = std::pow(2.0, octave);
    const size_t region = (size_t)ow * oh;
    if (pydata._num_dogs == 5 && params._num_dog_levels == 3) {
        // The reference resets the whole full-resolution maps and lets the kernel write the keypoints (siftfunctions.cu:
        // 120-125). Here the fused kernel writes EVERY pixel of the octave's region (keypoint or -1), so only what an
        // earlier, larger octave may have left beyond the region is reset: same final contents, ~6x fewer bytes.
        // (round 5: the reset rides in the detection launch -- three launches per octave fewer)
        size_t reset_end[3];
        for (int l = 0; l < 3; ++l) {
            reset_end[l] = pydata._dirty[l] > region ? pydata._dirty[l] : 0;
            pydata._dirty[l] = region < pydata._key_pts[l].size() ? region : pydata._key_pts[l].size();
        }
        const float *dog[5];
        float *res[3];
        for (int i = 0; i < 5; ++i) dog[i] = pydata._dog[i].data();
        for (int l = 0; l < 3; ++l) res[l] = reinterpret_cast<float *>(pydata._key_pts[l].data());
        nm_check(nm_find_keypoints3_reset_f32(dog, mask, mask_w, mask_h, ow, oh, params._peak_threshold, params._edge_threshold,
                                              xper, params._sigma_0, params._num_dog_levels, res, reset_end, stream),
                 "Keypoint detection launch failed");
        return;
    }
    for (int i = 1; i < pydata._num_dogs - 1; ++i) {    // any other level structure: level by level, as the reference
        nm_check(nm_fill_u32(pydata._key_pts[i - 1].data(), pydata._key_pts[i - 1].size() * 4, 0xBF800000u, stream),
                 "Keypoint map reset failed");
        pydata._dirty[i - 1] = region;
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

void compute_keypoints_with_mask(PyramidData &pydata, SiftParams &params, NmTexture mask, const int octave,
                                 const int octave_width, const int octave_height, hipStream_t stream)
{
    if (mask.format != NM_TEXEL_F32) RUNTIME_EXCEPTION("compute_keypoints_with_mask: the mask texture must hold floats");
    keypoints_impl(pydata, params, static_cast<const float *>(mask.data), mask.width, mask.height, octave, octave_width,
                   octave_height, stream);
}

void compute_orientations(PyramidData &pydata, const SiftParams &params, const int octave, const int octave_width,
                          const int octave_height, hipStream_t stream)
{
    const float xper = std::pow(2.0, octave);
    const int num_pixels_for_octave = octave_width * octave_height;
    if (params._num_dog_levels == 3 && (!eager_counts() || stream_is_capturing(stream))) {
        // Lazy counts (nm/lazy_count.h): the collation's three counts stay on the device, the orientation kernel reads them
        // there, and _orientations[l].size() becomes pending -- no host synchronisation in this call. It is also the ONLY
        // capture-compatible path (the eager one reads the counts back), so a capturing stream always takes it.
        pydata.gpu_collate_keypoints_for_octave_dev(num_pixels_for_octave, stream);
        int *words = nullptr;
        std::shared_ptr<nm::pending_counts> rec = pydata._ring.take(stream, 0x7u, &words);      // the three level counts
        const float *kp[3];
        float *res[3];
        for (int i = 0; i < 3; ++i) {
            pydata._orientations[i].reserve_uninitialized(nm_keypoint_bound(octave_width, octave_height));
            kp[i] = reinterpret_cast<const float *>(pydata._collated_kpts[i].data());
            res[i] = reinterpret_cast<float *>(pydata._orientations[i].data());
        }
        nm_check(nm_detect_orientations_levels_dev(kp, pydata.lazy_counts_dev(), num_pixels_for_octave,
                                                   reinterpret_cast<const float *>(pydata._grad.data()), octave_width,
                                                   octave_height, 1.5f, xper, res, words, stream),
                 "Orientation histogram launch failed");
        for (int i = 0; i < 3; ++i) pydata._orientations[i].defer_size(rec, i);
        pydata._lazy_rec = rec; pydata._lazy_octave = octave;
        return;
    }
    if (params._num_dog_levels == 3) {
        // one batched collation and ONE host synchronisation for the octave (the reference synchronises per level)
        int counts[3];
        pydata.gpu_collate_keypoints_for_octave(num_pixels_for_octave, counts, stream);
        const float *kp[3];
        float *res[3];
        int n[3], levels = 0;
        for (int i = 0; i < 3; ++i) pydata._orientations[i].resize_uninitialized(0);   // levels behind an empty one count as empty (Q9)
        for (int i = 0; i < 3; ++i) {
            pydata._orientations[i].resize_uninitialized((size_t)counts[i]);
            if (counts[i] == 0) break;                     // an empty level ends the octave (siftfunctions.cu:145)
            kp[levels] = reinterpret_cast<const float *>(pydata._collated_kpts[i].data());
            res[levels] = reinterpret_cast<float *>(pydata._orientations[i].data());
            n[levels++] = counts[i];
        }
        nm_check(nm_detect_orientations_levels(levels, kp, n, reinterpret_cast<const float *>(pydata._grad.data()),
                                               octave_width, octave_height, 1.5f, xper, res, stream),
                 "Orientation histogram launch failed");
        return;
    }
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
    const float *kp[3], *ori[3];
    float *desc[3], *xs[3], *ys[3];
    int n[3], levels = 0;
    // Lazy counts: compute_orientations of THIS octave left its counts on the device and nobody has looked at them since --
    // the descriptor kernel reads them (and the running item count) there; data._num_items becomes pending.
    const std::shared_ptr<nm::pending_counts> &rec = pydata._lazy_rec;
    if (params._num_dog_levels == 3 && rec && !rec->resolved() && pydata._lazy_octave == octave &&
        pydata._orientations[0].pending_record() == rec && pydata._orientations[1].pending_record() == rec &&
        pydata._orientations[2].pending_record() == rec && data._capacity > 0) {
        if (data._items_dev.size() < 2) data._items_dev = nm::device_vector<int>(2);
        const int capacity = (int)(data._desc.size() / SIFT_VECTOR_SIZE);
        const bool dev_base = data._num_items.pending();
        const int *base_in = dev_base ? data._items_dev.data() + data._items_cur : nullptr;
        int *items_out = data._items_dev.data() + (data._items_cur ^ 1);
        int *words = nullptr;
        std::shared_ptr<nm::pending_counts> items = pydata._ring.take(stream, 0x8u, &words);    // the running item count
        for (int i = 0; i < 3; ++i) {
            kp[i] = reinterpret_cast<const float *>(pydata._collated_kpts[i].data());
            ori[i] = reinterpret_cast<const float *>(pydata._orientations[i].data());
        }
        const int bound = octave_width * octave_height;
        nm_check(nm_compute_sift_descriptors_levels_dev(kp, ori, pydata.lazy_counts_dev(), bound, base_in,
                                                        dev_base ? 0 : data._num_items.host_value(), capacity, items_out,
                                                        words + 3, reinterpret_cast<const float *>(pydata._grad.data()),
                                                        octave_width, octave_height, params._num_dog_levels, xper,
                                                        data._desc.data(), data._x.data(), data._y.data(), stream),
                 "SIFT descriptor detection launch failed");
        data._items_cur ^= 1;
        data._num_items.defer(items);
        return;
    }
    for (int i = 0; i < params._num_dog_levels; ++i) {
        if (pydata._orientations[i].size() == 0) break;       // siftfunctions.cu:160
        int num_pts = (int)pydata._orientations[i].size();
        const int capacity = (int)(data._desc.size() / SIFT_VECTOR_SIZE);
        if (num_pts + data._num_items > capacity) num_pts = capacity - data._num_items;
        if (num_pts > 0) {
            kp[levels] = reinterpret_cast<const float *>(pydata._collated_kpts[i].data());
            ori[levels] = reinterpret_cast<const float *>(pydata._orientations[i].data());
            desc[levels] = data._desc.data() + (size_t)data._num_items * SIFT_VECTOR_SIZE;
            xs[levels] = data._x.data() + data._num_items;
            ys[levels] = data._y.data() + data._num_items;
            n[levels++] = num_pts;
            data._num_items += num_pts;
        }
        if (levels == 3 || i + 1 == params._num_dog_levels) {
            nm_check(nm_compute_sift_descriptors_levels(levels, kp, ori, n, reinterpret_cast<const float *>(pydata._grad.data()),
                                                        octave_width, octave_height, params._num_dog_levels, xper, desc,
                                                        xs, ys, stream),
                     "SIFT descriptor detection launch failed");
            levels = 0;
        }
    }
    if (levels > 0)
        nm_check(nm_compute_sift_descriptors_levels(levels, kp, ori, n, reinterpret_cast<const float *>(pydata._grad.data()),
                                                    octave_width, octave_height, params._num_dog_levels, xper, desc, xs,
                                                    ys, stream),
                 "SIFT descriptor detection launch failed");
}
