// pairs_native.cpp -- the frame-pair loop of bench.py with no Python and no torch: C ABI (include/nm_abi.h) + HIP runtime.
// Same synthetic input as bench.py / tests (SplitMix64 counter noise frames, sigma-4 zero-padded Gaussian pre-blur done
// by nm_convolve_f32), same two-phase step (batched detect+describe calls over a few streams, then batched DEVICE-SIZED
// match calls: the matcher reads the keypoint counts the frame driver left on the device, nothing comes back to the host
// inside the loop). Prints the keypoint counts of the first pair (12223 / 12080 for seeds 0 / 1 at 1080p, as everywhere
// else), read back AFTER the timed loop, and the throughput.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/pairs_native.cpp -Lniftymatch_amd/lib -lnm_hip \
//         -Wl,-rpath,$PWD/niftymatch_amd/lib -o pairs_native && ./pairs_native [pairs=32] [batch=16] [steps=10] [WxH]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nm_abi.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        int rc_ = (int)(x);                                                                        \
        if (rc_) { std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, nm_error_string(rc_)); std::exit(1); } \
    } while (0)

static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const int P = argc > 1 ? std::atoi(argv[1]) : 32;
    int B = argc > 2 ? std::atoi(argv[2]) : 16;
    const int steps = argc > 3 ? std::atoi(argv[3]) : 10;
    int W = 1920, H = 1080;
    if (argc > 4) std::sscanf(argv[4], "%dx%d", &W, &H);
    const int CAP = 16384, F = 2 * P, S = 4;
    if (B > NM_SIFT_MAX_BATCH) B = NM_SIFT_MAX_BATCH;
    while (F % B) --B;
    const int NB = F / B;
    const size_t npix = (size_t)W * H;

    // synthetic frames: uniform [0,255) noise, then the sigma-4 (sigma-3 at 640x480) pre-blur on the device
    const float sigma = (W == 640 && H == 480) ? 3.0f : 4.0f;
    const int radius = nm_create_kernel_for_sigma(sigma, nullptr);
    std::vector<float> taps(2 * radius + 1);
    nm_create_kernel_for_sigma(sigma, taps.data());
    float *d_taps, *d_raw, *d_buf;
    CHECK(hipMalloc(&d_taps, taps.size() * 4));
    CHECK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_raw, npix * 4));
    CHECK(hipMalloc(&d_buf, npix * 4));
    std::vector<float *> gray(F);
    std::vector<float> host(npix);
    for (int f = 0; f < F; ++f) {
        const uint64_t base = (uint64_t)f << 40;
        for (size_t i = 0; i < npix; ++i)
            host[i] = (float)(splitmix64(base + i) >> 40) * (1.0f / 16777216.0f) * 255.0f;
        CHECK(hipMemcpy(d_raw, host.data(), npix * 4, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&gray[f], npix * 4));
        CHECK(nm_convolve_f32(gray[f], d_raw, d_buf, W, H, d_taps, radius, nullptr));
    }
    CHECK(hipDeviceSynchronize());

    std::vector<nm_sift_arena *> arena(F);
    std::vector<float *> desc(F), x(F), y(F);
    std::vector<int *> cnt(F);
    for (int f = 0; f < F; ++f) {
        CHECK(nm_sift_arena_create(W, H, CAP, &arena[f]));
        CHECK(hipMalloc(&desc[f], (size_t)CAP * 128 * 4));
        CHECK(hipMalloc(&x[f], CAP * 4));
        CHECK(hipMalloc(&y[f], CAP * 4));
        CHECK(hipMalloc(&cnt[f], 4));
    }
    const int MB = P < NM_SIFT_MATCH_MAX_BATCH ? P : NM_SIFT_MATCH_MAX_BATCH;     // pairs per batched match call
    void *ws;
    CHECK(hipMalloc(&ws, nm_sift_match_batch_dev_workspace_bytes(MB, CAP, CAP)));
    std::vector<int *> result(P);
    for (int i = 0; i < P; ++i) CHECK(hipMalloc(&result[i], CAP * 4));
    hipStream_t st[S], ms;
    for (auto &s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
    hipEvent_t done[S], mdone;
    for (auto &e : done) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&mdone, hipEventDisableTiming));

    auto detect = [&]() {
        for (int c = 0; c < NB; ++c) {
            const float *g[NM_SIFT_MAX_BATCH];
            for (int k = 0; k < B; ++k) g[k] = gray[c * B + k];
            CHECK(nm_sift_detect_describe_batch(&arena[c * B], B, g, &desc[c * B], &x[c * B], &y[c * B], nullptr, nullptr,
                                                &cnt[c * B], st[c % S]));
        }
    };
    auto step = [&]() {
        detect();
        for (int s = 0; s < S; ++s) {
            CHECK(hipEventRecord(done[s], st[s]));
            CHECK(hipStreamWaitEvent(ms, done[s], 0));
        }
        for (int i0 = 0; i0 < P; i0 += MB) {
            const float *a[NM_SIFT_MATCH_MAX_BATCH], *b[NM_SIFT_MATCH_MAX_BATCH];
            const int *na[NM_SIFT_MATCH_MAX_BATCH], *nb[NM_SIFT_MATCH_MAX_BATCH];
            int *res[NM_SIFT_MATCH_MAX_BATCH];
            const int m = (P - i0 < MB) ? P - i0 : MB;
            for (int k = 0; k < m; ++k) {
                const int i = i0 + k;
                a[k] = desc[2 * i]; b[k] = desc[2 * i + 1]; na[k] = cnt[2 * i]; nb[k] = cnt[2 * i + 1]; res[k] = result[i];
            }
            CHECK(nm_sift_match_batch_dev_f32(m, a, na, b, nb, CAP, CAP, res, 0.8f, ws, ms));
        }
        CHECK(hipEventRecord(mdone, ms));
        for (int s = 0; s < S; ++s) CHECK(hipStreamWaitEvent(st[s], mdone, 0));   // next step reuses the arenas
    };
    for (int w = 0; w < 3; ++w) step();
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) step();
    CHECK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<int> n(F);
    for (int f = 0; f < F; ++f) CHECK(hipMemcpy(&n[f], cnt[f], 4, hipMemcpyDeviceToHost));
    std::vector<int> r0(n[0]);
    CHECK(hipMemcpy(r0.data(), result[0], (size_t)n[0] * 4, hipMemcpyDeviceToHost));
    int matched = 0;
    for (int v : r0) matched += (v >= 0);
    std::printf("{\"frame\": \"%dx%d\", \"pairs_per_step\": %d, \"frames_per_detect_call\": %d, \"steps\": %d, "
                "\"keypoints_pair0\": [%d, %d], \"matches_pair0\": %d, \"frame_pairs_per_s\": %.1f, \"ms_per_step\": %.3f}\n",
                W, H, P, B, steps, n[0], n[1], matched, P * steps / dt, 1e3 * dt / steps);
    for (int f = 0; f < F; ++f) nm_sift_arena_destroy(arena[f]);
    return 0;
}
