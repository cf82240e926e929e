// mfma_q64.hip -- would 64 queries per wave pay for the matcher? Per loop body: two ds_read_b128 (candidate fragments of
// the NEXT body) feed 8 MFMAs (32 queries per wave: two accumulators, the matcher's current shape) or 16 MFMAs (64 queries
// per wave: four accumulators), at one or two waves per SIMD. TFLOP/s of the MFMAs alone. Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define M "v_mfma_f32_32x32x2_f32 "

template <int Q64>
__global__ __launch_bounds__(512, 1) void loop_kernel(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 132; i += blockDim.x) lds[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-3f - 1.0f;
    __syncthreads();
    const float q0 = seed * lane, q1 = q0 + 1.f, q2 = q0 * 0.5f, q3 = q0 - 3.f;
    const float p0 = q0 * 1.5f, p1 = q1 * 0.25f, p2 = q2 + 2.f, p3 = q3 * 0.75f;
    f32x16 a0, a1, b0, b1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; b0[e] = 0.f; b1[e] = 0.f; }
    const float4 *row = reinterpret_cast<const float4 *>(&lds[(lane & 31) * 132 + 4 * (lane >> 5)]);
    float4 A = row[0], B = row[33 * 32];
    for (int it = 0; it < iters; ++it) {
        const float4 nA = row[2 * ((it + 1) & 15)], nB = row[2 * ((it + 1) & 15) + 33 * 32];
        if (Q64) {
            asm volatile(M "%0, %4, %12, %0\n" M "%1, %8, %12, %1\n" M "%2, %4, %16, %2\n" M "%3, %8, %16, %3\n"
                         M "%0, %5, %13, %0\n" M "%1, %9, %13, %1\n" M "%2, %5, %17, %2\n" M "%3, %9, %17, %3\n"
                         M "%0, %6, %14, %0\n" M "%1, %10, %14, %1\n" M "%2, %6, %18, %2\n" M "%3, %10, %18, %3\n"
                         M "%0, %7, %15, %0\n" M "%1, %11, %15, %1\n" M "%2, %7, %19, %2\n" M "%3, %11, %19, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)
                         : "v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w), "v"(q0), "v"(q1),
                           "v"(q2), "v"(q3), "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                         : "memory");
        } else {
            asm volatile(M "%0, %2, %10, %0\n" M "%1, %6, %10, %1\n" M "%0, %3, %11, %0\n" M "%1, %7, %11, %1\n"
                         M "%0, %4, %12, %0\n" M "%1, %8, %12, %1\n" M "%0, %5, %13, %0\n" M "%1, %9, %13, %1\n"
                         : "+v"(a0), "+v"(a1)
                         : "v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w), "v"(q0), "v"(q1),
                           "v"(q2), "v"(q3)
                         : "memory");
        }
        A = nA; B = nB;
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + b0[e] + b1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int Q64>
static void run(int threads, float *out, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<Q64>), dim3(256), dim3(threads), 0, 0, out, iters, 0.731f);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((loop_kernel<Q64>), dim3(256), dim3(threads), 0, 0, out, iters, 0.731f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * (threads / 64) * (double)iters * (Q64 ? 16 : 8) * 4096.0 * reps;
    printf("queries/wave %d  %d waves/SIMD  %7.1f us/launch  %6.1f TFLOP/s\n", Q64 ? 64 : 32, threads / 256, 1e3 * ms / reps,
           flops / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    for (int pass = 0; pass < 2; ++pass) {
        run<0>(256, out, 2400); run<0>(512, out, 1200); run<1>(256, out, 1200); run<1>(512, out, 600);
    }
    return 0;
}
