// mfma_bf16_peak.hip -- what this MI355X sustains on v_mfma_f32_32x32x16_bf16 (the bf16x3 screen's instruction), and at
// which clock: bare loops on random operands in registers, 1 or 2 waves per SIMD, 2 alternating accumulator chains per
// wave (the matcher's pattern), wall time over back-to-back launches plus the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz). The question it answers for DESIGN.md section 4: the matcher's MFMA
// pipe is busy for 114 k cycles per SIMD per launch -- how long is that?  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_peak.hip -o niftymatch_amd/lib/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(512, 2) void loop(float *out, unsigned long long *clk, int iters, unsigned seed)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u; a[q][k] = (short)(0x3C00 | ((s >> 9) & 0x03FF) | ((s >> 3) & 0x8000));   // +-[0.0078, 0.031)
            s = s * 1664525u + 1013904223u; b[q][k] = (short)(0x3C00 | ((s >> 9) & 0x03FF) | ((s >> 3) & 0x8000));
        }
    f32x16 acc[NACC];
    for (int g = 0; g < NACC; ++g) for (int e = 0; e < 16; ++e) acc[g][e] = (float)(lane + e);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[(q + g) & 3], acc[g], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int g = 0; g < NACC; ++g) for (int e = 0; e < 16; ++e) sum += acc[g][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
static void run(const char *name, int threads, float *out, unsigned long long *clk, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, clk, iters, 7u);
    CK(hipDeviceSynchronize());
    const int reps = 40;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, clk, iters, 7u + r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * grid);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(grid);
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double n_mfma = (double)grid * (threads / 64) * (double)iters * 4 * NACC * reps;       // wave-instructions
    const double flops = n_mfma * 32.0 * 32.0 * 16.0 * 2.0;
    const double us = 1e3 * ms / reps;
    // cycles per MFMA per SIMD = (kernel cycles) / (MFMAs per SIMD): 4 SIMDs per CU, one CU per workgroup
    const double per_simd = n_mfma / reps / (grid * 4.0);
    printf("%-34s %d waves/SIMD  %7.1f us/launch  %7.1f TFLOP/s  clock median %.3f GHz (min %.3f max %.3f)  %.1f cycles per MFMA per SIMD\n",
           name, threads / 256, us, flops / (ms * 1e-3) / 1e12, ghz[grid / 2], ghz[0], ghz[grid - 1], us * 1e-6 * ghz[grid / 2] * 1e9 / per_simd);
}

int main()
{
    float *out; unsigned long long *clk;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 256 * 2 * 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s  CUs %d  clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    for (int pass = 0; pass < 2; ++pass) {
        run<2>("32x32x16 bf16 regs, 2 acc", 256, out, clk, 1800);      // ~the matcher's MFMA count per SIMD per launch (3.6 k)
        run<2>("32x32x16 bf16 regs, 2 acc", 512, out, clk, 900);
        run<4>("32x32x16 bf16 regs, 4 acc", 512, out, clk, 450);
        run<2>("32x32x16 bf16 regs, 2 acc, long", 512, out, clk, 9000);
    }
    return 0;
}
