// allpairs_rccl.cpp -- BASELINE config 5 with no Python: all-pairs brute-force match of N x M random 128-D descriptors with
// the candidates row-sharded over the GPUs of one node, through the native entry nm_sift_match_allgather_f32 (shard ->
// ONE ncclAllGather of 12 B per row per rank over xGMI -> merge). One process, one host thread per GPU, a communicator
// from ncclCommInitAll. With one visible GPU it runs as a single rank (the all-gather is then skipped by the library).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/allpairs_rccl.cpp -Lniftymatch_amd/lib -lnm_hip -lrccl \
//         -Wl,-rpath,$PWD/niftymatch_amd/lib -o allpairs_rccl && ./allpairs_rccl [N=100000] [M=100000] [gpus=all] [steps=3]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "nm_abi.h"

#define CHECK(x)                                                                                    \
    do {                                                                                            \
        int rc_ = (int)(x);                                                                         \
        if (rc_) { std::fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, rc_); std::exit(1); } \
    } while (0)

static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void fill(std::vector<float> &v, uint64_t seed)
{
    for (size_t i = 0; i < v.size(); ++i) v[i] = (float)(splitmix64((seed << 40) + i) >> 40) * (1.0f / 16777216.0f);
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 100000, M = argc > 2 ? std::atoi(argv[2]) : 100000;
    int ndev = 0;
    CHECK(hipGetDeviceCount(&ndev));
    int G = argc > 3 ? std::atoi(argv[3]) : ndev;
    if (G < 1 || G > ndev) G = ndev;
    const int steps = argc > 4 ? std::atoi(argv[4]) : 3;
    std::vector<float> hA((size_t)N * 128), hB((size_t)M * 128);
    fill(hA, 1); fill(hB, 2);
    std::vector<ncclComm_t> comms(G);
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) devs[g] = g;
    if (G > 1) CHECK(ncclCommInitAll(comms.data(), G, devs.data()));
    std::vector<std::vector<int>> results(G, std::vector<int>(N));
    std::vector<double> ms(G);
    auto rank = [&](int g) {
        CHECK(hipSetDevice(g));
        const int base = M / G, extra = M % G;
        const int begin = g * base + (g < extra ? g : extra), rows = base + (g < extra ? 1 : 0);
        float *dA, *dB; int *dR; void *ws; hipStream_t st;
        CHECK(hipStreamCreate(&st));
        CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, (size_t)(rows ? rows : 1) * 512));
        CHECK(hipMalloc(&dR, (size_t)N * 4));
        CHECK(hipMalloc(&ws, nm_sift_match_allgather_workspace_bytes(N, rows, G)));
        CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, hB.data() + (size_t)begin * 128, (size_t)rows * 512, hipMemcpyHostToDevice));
        CHECK(hipMemset(dR, 0xff, (size_t)N * 4));
        auto call = [&] { CHECK(nm_sift_match_allgather_f32(dA, N, dB, rows, begin, G, dR, 0.8f, ws, G > 1 ? comms[g] : nullptr, st)); };
        call();
        CHECK(hipStreamSynchronize(st));
        const auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) call();
        CHECK(hipStreamSynchronize(st));
        ms[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        CHECK(hipMemcpy(results[g].data(), dR, (size_t)N * 4, hipMemcpyDeviceToHost));
    };
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) th.emplace_back(rank, g);
    for (auto &t : th) t.join();
    // every rank must hold the same answer; spot-check 32 queries against a host brute force
    int bad = 0;
    for (int g = 1; g < G; ++g) bad += results[g] != results[0];
    for (int q = 0; q < 32; ++q) {
        const int i = (int)(splitmix64(777 + q) % (uint64_t)N);
        double m1 = 1e300, m2 = 1e300; int idx = -1;
        for (int j = 0; j < M; ++j) {
            double d = 0;
            for (int k = 0; k < 128; ++k) { const double t = (double)hA[(size_t)i * 128 + k] - hB[(size_t)j * 128 + k]; d += t * t; }
            if (d < m1) { m2 = m1; m1 = d; idx = j; } else if (d < m2) m2 = d;
        }
        const int want = (m1 / m2 < 0.8) ? idx : -1;
        if (results[0][i] != want) ++bad;
    }
    double worst = 0;
    for (double m : ms) worst = m > worst ? m : worst;
    std::printf("%d x %d over %d rank(s): %.3f ms per call = %.2f TFLOP/s (2NM128) aggregate; mismatches %d\n", N, M, G, worst,
                256.0 * N * M / (worst * 1e-3) / 1e12, bad);
    return bad != 0;
}
