// nm_client.cpp -- example client of the drop-in C++ API (see include/nm_client.h): the frame loop an application
// written against NiftyMatch owns, using only the public headers.
#include "../../../include/nm_client.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <condition_variable>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../convolution.h"
#include "../downsample.h"
#include "../ransac.h"
#include "../siftfunctions.h"

// The frame loop a NiftyMatch application owns: every call below is a function of the reference's public
// headers with the reference's argument list. `gray` is on the device; `out` is reset first.
static void client_frame(const SiftParams &params, PyramidData &py, const float *gray, SiftData &out, hipStream_t stream)
{
    const int width = params._width, height = params._height;
    out._num_items = 0;
    convolve<float>(py._octave[0].data(), gray, py._buffer.data(), width, height, py._base_kernel.data(), py._base_radius,
                    stream);
    for (int o = 0; o < params._num_octaves; ++o) {
        const int ow = width >> o, oh = height >> o;
        if (o > 0)   // level 3 has twice the base sigma: it seeds the next octave
            downsample_by_2<float>(py._octave[0].data(), ow, oh, py._octave[3].data(), width >> (o - 1), height >> (o - 1),
                                   stream);
        for (int i = 1; i < py._num_octaves; ++i)
            convolve<float>(py._octave[i].data(), py._octave[i - 1].data(), py._buffer.data(), ow, oh,
                            py._kernels[i - 1].data(), py._kernel_radii[i - 1], stream);
        compute_dog(py, ow, oh, stream);
        compute_gradients(py, params, ow, oh, stream);
        compute_keypoints(py, params, o, ow, oh, stream);
        compute_orientations(py, params, o, ow, oh, stream);
        compute_descriptors(py, params, o, ow, oh, out, stream);
    }
}

extern "C" int nm_client_detect_describe(const float *gray, int width, int height, int capacity, float *desc, float *x,
                                         float *y)
{
    try {
        SiftParams params(width, height);
        PyramidData py(params);
        SiftData out(capacity);
        const size_t npix = (size_t)width * height;
        nm::device_vector<float> d_gray(std::vector<float>(gray, gray + npix));
        client_frame(params, py, d_gray.data(), out, 0);
        const int n = out._num_items;
        if (n > 0) {
            std::vector<float> h = out._desc.to_host();
            std::memcpy(desc, h.data(), (size_t)n * 128 * sizeof(float));
            h = out._x.to_host(); std::memcpy(x, h.data(), (size_t)n * sizeof(float));
            h = out._y.to_host(); std::memcpy(y, h.data(), (size_t)n * sizeof(float));
        }
        return n;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}

// Value semantics of the containers, as a client written against the reference may rely on them (thrust vectors are freely
// copyable): the same frame through (0) the original PyramidData, (1) a copy, (2) an object assigned from a temporary, (3) an
// element of a std::vector<PyramidData> that has reallocated, (4) a moved-to object; SiftData through its copy constructor and
// assignment. Returns the number of descriptors when every variant produced identical descriptors, -2 on a mismatch, -1 on an
// exception. With the raw pinned pointer PyramidData used to hold, (1)-(3) freed that buffer twice.
extern "C" int nm_client_copy_semantics(const float *gray, int width, int height, int capacity)
{
    try {
        SiftParams params(width, height);
        const size_t npix = (size_t)width * height;
        nm::device_vector<float> d_gray(std::vector<float>(gray, gray + npix));
        auto run = [&](PyramidData &py, std::vector<float> &desc) -> int {
            SiftData out(capacity);
            client_frame(params, py, d_gray.data(), out, 0);
            SiftData copy(out), assigned;                  // deep copies: own vectors, own raw pointers
            assigned = copy;
            if (assigned._x_ptr != assigned._x.data() || assigned._x_ptr == out._x_ptr || assigned._num_items != out._num_items)
                return -2;
            desc = assigned._desc.to_host();
            desc.resize((size_t)out._num_items * 128);
            return out._num_items;
        };
        std::vector<float> ref, got;
        PyramidData original(params);
        const int n = run(original, ref);
        if (n < 0) return n;
        {
            PyramidData copy(original);                    // (1)
            if (run(copy, got) != n || got != ref) return -2;
            PyramidData assigned;
            assigned = PyramidData(params);                // (2) move-assignment from a temporary
            if (run(assigned, got) != n || got != ref) return -2;
            assigned = original;                           // copy-assignment over a live object
            if (run(assigned, got) != n || got != ref) return -2;
            std::vector<PyramidData> many;
            many.push_back(original);
            many.push_back(original);                      // (3) growth: elements are moved or copied, old ones destroyed
            many.emplace_back(params);
            for (PyramidData &py : many)
                if (run(py, got) != n || got != ref) return -2;
            PyramidData moved(std::move(many[0]));         // (4)
            if (run(moved, got) != n || got != ref) return -2;
        }                                                  // everything is destroyed here: no double free
        if (run(original, got) != n || got != ref) return -2;
        return n;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}

// Lazy counts (nm/lazy_count.h) against the reference's observable state. The frame is run three ways with the same per-octave
// loop: (a) never looking at a count until the end -- no host synchronisation inside the frame; (b) LOOKING after every call, as
// a curious client may (pydata._orientations[l].size() after compute_orientations, data._num_items after compute_descriptors):
// every look resolves a pending count; (c) NM_EAGER_COUNTS behaviour (one synchronisation per octave, the counts on the host
// before compute_orientations returns). watch (host, 4 ints per octave: the three sizes and the running item count) must be
// identical for (b) and (c), the descriptors / coordinates identical for all three, `pending_seen` reports how many of (b)'s
// looks found a pending value (> 0: the lazy path was really taken). A capacity below the frame's keypoint count exercises
// the clipping on the device. Returns the item count, -2 on a mismatch, -1 on an exception.
extern "C" void nm_set_eager_counts(int on);
extern "C" int nm_client_lazy_counts(const float *gray, int width, int height, int capacity, int *watch, int max_octaves,
                                     int *pending_seen)
{
    try {
        SiftParams params(width, height);
        const size_t npix = (size_t)width * height;
        nm::device_vector<float> d_gray(std::vector<float>(gray, gray + npix));
        hipStream_t st = nullptr;
        nm_check((int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream");
        struct Out { int n; std::vector<float> desc, x, y; std::vector<int> watch; int pending; };
        auto run = [&](int look, bool eager) -> Out {
            nm_set_eager_counts(eager ? 1 : 0);
            PyramidData py(params);
            SiftData out(capacity);
            Out r{0, {}, {}, {}, std::vector<int>(4 * (size_t)max_octaves, -1), 0};
            out._num_items = 0;
            convolve<float>(py._octave[0].data(), d_gray.data(), py._buffer.data(), width, height, py._base_kernel.data(),
                            py._base_radius, st);
            for (int o = 0; o < params._num_octaves; ++o) {
                const int ow = width >> o, oh = height >> o;
                if (o > 0)
                    downsample_by_2<float>(py._octave[0].data(), ow, oh, py._octave[3].data(), width >> (o - 1), height >> (o - 1), st);
                for (int i = 1; i < py._num_octaves; ++i)
                    convolve<float>(py._octave[i].data(), py._octave[i - 1].data(), py._buffer.data(), ow, oh,
                                    py._kernels[i - 1].data(), py._kernel_radii[i - 1], st);
                compute_dog(py, ow, oh, st);
                compute_gradients(py, params, ow, oh, st);
                compute_keypoints(py, params, o, ow, oh, st);
                compute_orientations(py, params, o, ow, oh, st);
                if (look == 1 && o < max_octaves) {
                    // (the three sizes share one pending record: the first look resolves it for all of them, so "pending" is
                    // asked of all three before any size is read)
                    for (int l = 0; l < 3; ++l) r.pending += py._orientations[l].size_pending() ? 1 : 0;
                    for (int l = 0; l < 3; ++l) r.watch[4 * o + l] = (int)py._orientations[l].size();
                }
                compute_descriptors(py, params, o, ow, oh, out, st);
                if (look && o < max_octaves) {
                    r.pending += out._num_items.pending() ? 1 : 0;
                    r.watch[4 * o + 3] = out._num_items;
                }
            }
            r.n = out._num_items;                              // (a): the one and only look
            nm_check((int)hipStreamSynchronize(st), "sync");
            r.desc = out._desc.to_host(); r.x = out._x.to_host(); r.y = out._y.to_host();
            r.desc.resize((size_t)r.n * 128); r.x.resize(r.n); r.y.resize(r.n);
            return r;
        };
        // d: looks at the item count only -- compute_descriptors then takes the device-sized path in every octave and the look
        // resolves ITS pending word
        const Out a = run(0, false), b = run(1, false), c = run(1, true), d = run(2, false);
        nm_set_eager_counts(0);
        (void)hipStreamDestroy(st);
        if (pending_seen) *pending_seen = b.pending + d.pending;
        for (int o = 0; o < max_octaves; ++o)
            if (d.watch[4 * o + 3] != c.watch[4 * o + 3]) return -2;
        if (d.n != a.n || d.desc != a.desc || d.x != a.x || d.y != a.y) return -2;
        for (int i = 0; watch && i < 4 * max_octaves; ++i) watch[i] = b.watch[i];
        if (a.n != b.n || a.n != c.n || a.desc != b.desc || a.desc != c.desc || a.x != c.x || a.y != c.y || b.x != c.x) return -2;
        if (b.watch != c.watch || c.pending != 0) return -2;
        return a.n;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        nm_set_eager_counts(0);
        return -1;
    }
}

// Throughput of the drop-in path as an application would drive it: `reps` times { detect+describe both frames with the
// reference's per-octave client loop, compute_sift_matches(A, B, distance) }. Objects are created once, as a real client
// does. gray0/gray1 are DEVICE planes. with_distance != 0 passes a caller-allocated N x M `distance` (the reference's
// mandatory argument); 0 passes NULL (the extension). Returns the wall-clock microseconds per pair (host clock around
// the loop, device drained), or a negative value on error; n_out receives the two keypoint counts and the match count.
// streams = 1: the loop above on one stream from one host thread (what round 2 measured). streams = 2: the two frames of a
// pair are independent until the match, so a client may drive them from two host threads on two streams with a PyramidData
// each (the per-octave API costs ~110 launches and 6 count read-backs per frame: one host thread is the bottleneck, not
// the GPU); the match follows on the first stream once both frames are described.
extern "C" double nm_client_pair_loop_ex(const float *gray0, const float *gray1, int width, int height, int capacity, int reps,
                                         int with_distance, int streams, int *n_out)
{
    try {
        SiftParams params(width, height);
        PyramidData py(params);
        SiftData a(capacity), b(capacity);
        nm::device_vector<float> dist(with_distance ? (size_t)capacity * capacity : 0);
        hipStream_t st = nullptr, st1 = nullptr;
        int device = 0;
        nm_check((int)hipGetDevice(&device), "hipGetDevice");
        // second frame's worker: one persistent host thread, woken per pair
        std::unique_ptr<PyramidData> py1;
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        int posted = 0, done = 0;
        bool quit = false, failed = false;
        if (streams >= 2) {
            py1.reset(new PyramidData(params));
            nm_check((int)hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream");
            nm_check((int)hipStreamCreateWithFlags(&st1, hipStreamNonBlocking), "stream");
            worker = std::thread([&]() {
                if (hipSetDevice(device) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); failed = true; }
                int seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return quit || posted > seen; });
                        if (quit) return;
                        seen = posted;
                    }
                    bool bad = false;
                    try {
                        client_frame(params, *py1, gray1, b, st1);
                        if (hipStreamSynchronize(st1) != hipSuccess) bad = true;
                    } catch (...) { bad = true; }
                    { std::lock_guard<std::mutex> lk(mu); done = seen; failed = failed || bad; }
                    cv.notify_all();
                }
            });
        }
        double trace_us[2] = {0, 0};                        // host time inside the frames' calls / inside compute_sift_matches
        auto pair = [&]() {
            const auto t0 = std::chrono::steady_clock::now();
            if (streams >= 2) {
                { std::lock_guard<std::mutex> lk(mu); ++posted; }
                cv.notify_all();
                client_frame(params, py, gray0, a, st);
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done == posted; });
            } else {
                client_frame(params, py, gray0, a, st);
                client_frame(params, py, gray1, b, st);
            }
            const auto t1 = std::chrono::steady_clock::now();
            compute_sift_matches(&a, &b, with_distance ? dist.data() : nullptr, 0.8f, st);
            const auto t2 = std::chrono::steady_clock::now();
            trace_us[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
            trace_us[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
        };
        // Runs on EVERY way out of this scope, an exception thrown by pair() included: a joinable std::thread that is
        // destroyed calls std::terminate, and the streams would leak.
        struct Cleanup {
            std::thread &worker; std::mutex &mu; std::condition_variable &cv; bool &quit; hipStream_t &st, &st1;
            ~Cleanup()
            {
                if (worker.joinable()) {
                    { std::lock_guard<std::mutex> lk(mu); quit = true; }
                    cv.notify_all();
                    worker.join();
                }
                if (st) { (void)hipStreamDestroy(st); st = nullptr; }
                if (st1) { (void)hipStreamDestroy(st1); st1 = nullptr; }
            }
        } cleanup{worker, mu, cv, quit, st, st1};
        double us = -1.0;
        pair();                                                       // warm-up (workspace growth, code loading)
        trace_us[0] = trace_us[1] = 0;
        if (hipDeviceSynchronize() == hipSuccess) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) pair();
            if (hipDeviceSynchronize() == hipSuccess)
                us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        }
        { std::lock_guard<std::mutex> lk(mu); if (failed) return -1.0; }
        if (getenv("NM_CLIENT_TRACE") && reps > 0)
            std::cerr << "client loop: host inside the two frames' calls " << trace_us[0] / reps << " us, inside compute_sift_matches "
                      << trace_us[1] / reps << " us per pair" << std::endl;
        if (n_out && us >= 0) {
            n_out[0] = a._num_items; n_out[1] = b._num_items;
            std::vector<int> m = a._match_indexes.to_host();
            int cnt = 0;
            for (int i = 0; i < a._num_items; ++i) cnt += m[i] >= 0;
            n_out[2] = cnt;
        }
        return us;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1.0;
    }
}

extern "C" double nm_client_pair_loop(const float *gray0, const float *gray1, int width, int height, int capacity, int reps,
                                      int with_distance, int *n_out)
{
    return nm_client_pair_loop_ex(gray0, gray1, width, height, capacity, reps, with_distance, 1, n_out);
}

extern "C" int nm_client_match(const float *A, int nA, const float *B, int nB, float *distance, int *result,
                               float ambiguity)
{
    try {
        SiftData a(nA), b(nB);
        a._desc = nm::device_vector<float>(std::vector<float>(A, A + (size_t)nA * 128));
        b._desc = nm::device_vector<float>(std::vector<float>(B, B + (size_t)nB * 128));
        a._match_indexes = nm::device_vector<int>(std::vector<int>(result, result + nA));
        a._num_items = nA; b._num_items = nB;
        nm::device_vector<float> d_dist(distance ? (size_t)nA * nB : 0);
        compute_sift_matches(&a, &b, distance ? d_dist.data() : nullptr, ambiguity);
        std::vector<int> r = a._match_indexes.to_host();
        std::memcpy(result, r.data(), (size_t)nA * sizeof(int));
        if (distance) {
            std::vector<float> h = d_dist.to_host();
            std::memcpy(distance, h.data(), h.size() * sizeof(float));
        }
        return 0;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}

extern "C" int nm_client_ransac(int model, const float *src_x, const float *src_y, const float *dst_x, const float *dst_y,
                                int n, float inlier_threshold, int iterations, unsigned int seed, float *H)
{
    try {
        nm::device_vector<float> sx(std::vector<float>(src_x, src_x + n)), sy(std::vector<float>(src_y, src_y + n));
        nm::device_vector<float> dx(std::vector<float>(dst_x, dst_x + n)), dy(std::vector<float>(dst_y, dst_y + n));
        nm::device_vector<float> h(9, 0.f);
        nm_ransac_seed(seed);
        bool ok = false;
        if (model == 0) ok = ransac_translation(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        else if (model == 1) ok = ransac_similarity(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        else ok = ransac_homography(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        nm_ransac_seed(0);
        std::vector<float> hh = h.to_host();
        std::memcpy(H, hh.data(), 9 * sizeof(float));
        return ok ? 1 : 0;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}
