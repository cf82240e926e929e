// hbm_mix.hip -- what this MI355X moves per second for a given READ : WRITE mix of plain streaming traffic (float4 per lane,
// fully coalesced, arrays far larger than the 256 MB Infinity Cache): mixes 1:0, 1:1, 1:2, 1:3 and 0:1.
// Round 3: the round-2 form of this probe (4 096 blocks, grid-stride, 4 strided float4 per thread) under-drove the device
// (4.65 TB/s for a copy where MI355X_MICROARCH.md measures 6.29). This version SWEEPS the launch shape -- contiguous chunk per
// workgroup or one float4 per thread, 1/2/4/8 float4 in flight per thread, grids from 2 048 workgroups to one per 4 KiB --
// and reports the best of the sweep per mix, so that a number from it is a property of the device, not of one launch shape.
// It may be cited as a ceiling only if its copy row reproduces the guide's >= 6 TB/s.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_mix.hip -o niftymatch_amd/lib/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// Workgroup b owns the contiguous range [b * per_block, (b + 1) * per_block) float4; U float4 per thread are in flight together.
template <int NR, int NW, int U>
__global__ __launch_bounds__(256) void mix(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4, size_t per_block,
                                           size_t stride4, float *sink)
{
    float acc = 0.f;
    const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
    for (size_t i0 = b0 + threadIdx.x; i0 < b1; i0 += (size_t)U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = make_float4(1.f, 2.f, 3.f, 4.f);
            if (NR && i0 + u * 256 < b1) v[u] = in[i0 + u * 256];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + u * 256;
            if (i >= b1) break;
            if (NW == 0) acc += v[u].x + v[u].y + v[u].z + v[u].w;
#pragma unroll
            for (int w = 0; w < NW; ++w) out[i + w * stride4] = make_float4(v[u].x + w, v[u].y, v[u].z, v[u].w);
        }
    }
    if (NW == 0 && acc == 12345.678f) *sink = acc;
}

template <int NR, int NW, int U>
static double run_shape(const float4 *in, float4 *out, size_t n4, size_t blocks, float *sink)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t per_block = ((n4 + blocks - 1) / blocks + 255) / 256 * 256;
    const unsigned grid = (unsigned)((n4 + per_block - 1) / per_block);
    hipLaunchKernelGGL((mix<NR, NW, U>), dim3(grid), dim3(256), 0, 0, in, out, n4, per_block, n4, sink);
    CK(hipDeviceSynchronize());
    const int reps = 6;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix<NR, NW, U>), dim3(grid), dim3(256), 0, 0, in, out, n4, per_block, n4, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)n4 * 16.0 * (NR + NW) * reps / (ms * 1e-3) / 1e9;
}

template <int NR, int NW>
static void run(const char *name, const float4 *in, float4 *out, size_t n4, float *sink)
{
    static const size_t grids[] = {2048, 4096, 8192, 16384, 65536, 262144, 0};      // 0: one 256-float4 chunk (4 KiB) per workgroup
    double best = 0; size_t bg = 0; int bu = 0;
    double at_r2 = 0;
    for (size_t g : grids) {
        const size_t blocks = g ? g : (n4 + 255) / 256;
        const double r1 = run_shape<NR, NW, 1>(in, out, n4, blocks, sink);
        const double r2 = run_shape<NR, NW, 2>(in, out, n4, blocks, sink);
        const double r4 = run_shape<NR, NW, 4>(in, out, n4, blocks, sink);
        const double r8 = run_shape<NR, NW, 8>(in, out, n4, blocks, sink);
        if (g == 4096) at_r2 = r4;
        const double r[4] = {r1, r2, r4, r8};
        for (int k = 0; k < 4; ++k) if (r[k] > best) { best = r[k]; bg = blocks; bu = 1 << k; }
    }
    printf("%-22s best %7.1f GB/s total (%6.1f read + %6.1f written) at %zu workgroups x %d float4 in flight;  4096 x 4: %7.1f\n", name,
           best, best * NR / (NR + NW), best * NW / (NR + NW), bg, bu, at_r2);
}

int main()
{
    const size_t n4 = (size_t)32 << 20;                       // 512 MiB per array
    float4 *in, *out; float *sink;
    CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16 * 3)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, n4 * 16)); CK(hipMemset(out, 0, n4 * 16 * 3));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %zu MiB per array, best launch shape of a sweep per mix\n", p.name, n4 * 16 >> 20);
    for (int pass = 0; pass < 2; ++pass) {
        run<1, 0>("read only (1 : 0)", in, out, n4, sink);
        run<1, 1>("copy (1 : 1)", in, out, n4, sink);
        run<1, 2>("1 read : 2 written", in, out, n4, sink);
        run<1, 3>("1 read : 3 written", in, out, n4, sink);
        run<0, 1>("write only (0 : 1)", in, out, n4, sink);
    }
    return 0;
}
