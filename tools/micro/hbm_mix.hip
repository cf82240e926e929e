// hbm_mix.hip -- what this MI355X moves per second for a given READ : WRITE mix of plain streaming traffic (float4 per lane,
// fully coalesced, arrays far larger than the 256 MB Infinity Cache). The scale-space chain writes 2.4 bytes for every
// byte it reads (levels + DoG + gradient planes: 177 MB written, 74 MB read per 1080p frame), so the copy figure
// (1 : 1) is not its ceiling; this probe measures the mixes 1:0, 1:1, 1:2, 1:3 and 0:1.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_mix.hip -o niftymatch_amd/lib/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// NT: nontemporal stores. Four float4 per thread and iteration are in flight together.
template <int NR, int NW, bool NT = false>
__global__ __launch_bounds__(256) void mix(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4, size_t stride4, float *sink)
{
    float acc = 0.f;
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * step) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = make_float4(1.f, 2.f, 3.f, 4.f);
            if (NR && i0 + u * step < n4) v[u] = in[i0 + u * step];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * step;
            if (i >= n4) break;
            if (NW == 0) acc += v[u].x + v[u].y + v[u].z + v[u].w;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (NT) __builtin_nontemporal_store((v4f){v[u].x + w, v[u].y, v[u].z, v[u].w}, reinterpret_cast<v4f *>(out + i + w * stride4));
                else out[i + w * stride4] = make_float4(v[u].x + w, v[u].y, v[u].z, v[u].w);
            }
        }
    }
    if (NW == 0 && acc == 12345.678f) *sink = acc;
}

template <int NR, int NW, bool NT = false>
static void run(const char *name, const float4 *in, float4 *out, size_t n4, float *sink)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, in, out, n4, n4, sink);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, in, out, n4, n4, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n4 * 16.0 * (NR + NW) * reps;
    printf("%-28s %8.1f us per pass  %7.1f GB/s total (%6.1f read + %6.1f written)\n", name, 1e3 * ms / reps, bytes / (ms * 1e-3) / 1e9,
           (double)n4 * 16.0 * NR * reps / (ms * 1e-3) / 1e9, (double)n4 * 16.0 * NW * reps / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t n4 = (size_t)32 << 20;                       // 512 MiB per array
    float4 *in, *out; float *sink;
    CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16 * 3)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, n4 * 16)); CK(hipMemset(out, 0, n4 * 16 * 3));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %zu MiB per array\n", p.name, n4 * 16 >> 20);
    for (int pass = 0; pass < 2; ++pass) {
        run<1, 0>("read only (1 : 0)", in, out, n4, sink);
        run<1, 1>("copy (1 : 1)", in, out, n4, sink);
        run<1, 2>("1 read : 2 written", in, out, n4, sink);
        run<1, 3>("1 read : 3 written", in, out, n4, sink);
        run<0, 1>("write only (0 : 1)", in, out, n4, sink);
        run<1, 2, true>("1 : 2, nontemporal stores", in, out, n4, sink);
        run<1, 3, true>("1 : 3, nontemporal stores", in, out, n4, sink);
        run<0, 1, true>("write only, nontemporal", in, out, n4, sink);
    }
    return 0;
}
