// mfma_peak.hip -- what this MI355X sustains on fp32 MFMA (the matcher's ceiling), measured on the device the matcher runs
// on: bare v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 loops on random operands, 1 or 2 waves per SIMD, with and
// without an LDS operand stream, wall time over back-to-back launches plus the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz). Diagnostic only; nothing in the product links it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o niftymatch_amd/lib/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// SHAPE 0: 32x32x2 with NACC accumulators; SHAPE 1: 16x16x4 with 4*NACC accumulators (same registers / flops per step).
// LDS 0: operands in registers; 1: the A operand of every 4 MFMAs comes from one ds_read_b128 (the matcher's pattern).
template <int SHAPE, int NACC, int LDSOP>
__global__ __launch_bounds__(512, 2) void mfma_loop(float *out, unsigned long long *clk, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 132; i += blockDim.x) lds[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-3f - 1.0f;
    __syncthreads();
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = seed * (lane * 4 + k + 1) * 0.37f - 3.0f; b[k] = seed * (lane * 7 + k + 3) * 0.11f - 2.0f; }
    f32x16 acc[NACC];
    for (int g = 0; g < NACC; ++g) for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    const float4 *row = reinterpret_cast<const float4 *>(&lds[(lane & 31) * 132 + 4 * (lane >> 5)]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        float4 cf[NACC];
        if (LDSOP) {
#pragma unroll
            for (int g = 0; g < NACC; ++g) cf[g] = row[(2 * ((it + g) & 15)) + g * 33 * 32];
        }
#pragma unroll
        for (int g = 0; g < NACC; ++g) {
            const float a0 = LDSOP ? cf[g].x : a[0], a1 = LDSOP ? cf[g].y : a[1], a2 = LDSOP ? cf[g].z : a[2], a3 = LDSOP ? cf[g].w : a[3];
            if (SHAPE == 0) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[0], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[1], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[2], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b[3], acc[g], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {           // four 16x16 accumulators live in one f32x16
                    f32x4 c = {acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]};
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[q], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[(q + 1) & 3], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[(q + 2) & 3], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[(q + 3) & 3], c, 0, 0, 0);
                    acc[g][4 * q] = c[0]; acc[g][4 * q + 1] = c[1]; acc[g][4 * q + 2] = c[2]; acc[g][4 * q + 3] = c[3];
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int g = 0; g < NACC; ++g) for (int e = 0; e < 16; ++e) s += acc[g][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int NACC, int LDSOP>
static void run(const char *name, int threads, float *out, unsigned long long *clk, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((mfma_loop<SHAPE, NACC, LDSOP>), dim3(grid), dim3(threads), 0, 0, out, clk, iters, 0.731f);
    CK(hipDeviceSynchronize());
    const int reps = 40;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop<SHAPE, NACC, LDSOP>), dim3(grid), dim3(threads), 0, 0, out, clk, iters, 0.731f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * grid);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz(grid);
    for (int i = 0; i < grid; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    // flops per MFMA step (per wave): NACC x 4 x (32x32x2 = 4096 flop) either shape
    const double flops = (double)grid * (threads / 64) * (double)iters * NACC * 4 * 4096.0 * reps;
    printf("%-44s %d waves/SIMD  %7.1f us/launch  %6.1f TFLOP/s  in-kernel clock median %.3f GHz (min %.3f max %.3f)\n", name,
           threads / 256, 1e3 * ms / reps, flops / (ms * 1e-3) / 1e12, ghz[grid / 2], ghz[0], ghz[grid - 1]);
}

int main()
{
    float *out; unsigned long long *clk;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 256 * 2 * 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s  CUs %d  clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    for (int pass = 0; pass < 2; ++pass) {
        run<0, 2, 0>("32x32x2 regs, 2 acc", 256, out, clk, 2400);
        run<0, 2, 0>("32x32x2 regs, 2 acc", 512, out, clk, 1200);
        run<0, 4, 0>("32x32x2 regs, 4 acc", 256, out, clk, 1200);
        run<1, 2, 0>("16x16x4 regs, 8 acc", 256, out, clk, 2400);
        run<1, 2, 0>("16x16x4 regs, 8 acc", 512, out, clk, 1200);
        run<0, 2, 1>("32x32x2 A from ds_read_b128, 2 acc", 256, out, clk, 2400);
        run<0, 2, 1>("32x32x2 A from ds_read_b128, 2 acc", 512, out, clk, 1200);
        run<1, 2, 1>("16x16x4 A from ds_read_b128, 8 acc", 512, out, clk, 1200);
    }
    return 0;
}
