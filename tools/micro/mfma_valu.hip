// mfma_valu.hip -- does VALU work hide under v_mfma_f32_32x32x2_f32 on gfx950? (the matcher's selection epilogue question)
// A hand-placed loop body: 8 MFMAs on two alternating accumulator chains, K independent v_med3_i32 after every MFMA, and
// optionally the next iteration's A operands fetched by two ds_read_b128 issued at the top of the body (waited for only
// at its end). Wall time per launch, TFLOP/s of the MFMAs alone. Diagnostic; nothing in the product links it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define REP0(x)
#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP12(x) REP8(x) REP4(x)
#define VALU "v_med3_i32 %[k0], %[k0], %[k1], %[k2]\n v_med3_i32 %[k1], %[k1], %[k2], %[k3]\n"   /* 2 VALU */

#define BODY(REP)                                                        \
    asm volatile(                                                        \
        "v_mfma_f32_32x32x2_f32 %[c0], %[a0], %[b0], %[c0]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c1], %[a4], %[b0], %[c1]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c0], %[a1], %[b1], %[c0]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c1], %[a5], %[b1], %[c1]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c0], %[a2], %[b2], %[c0]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c1], %[a6], %[b2], %[c1]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c0], %[a3], %[b3], %[c0]\n" REP(VALU)  \
        "v_mfma_f32_32x32x2_f32 %[c1], %[a7], %[b3], %[c1]\n" REP(VALU)  \
        : [c0] "+v"(acc0), [c1] "+v"(acc1), [k0] "+v"(k0), [k1] "+v"(k1)  \
        : [a0] "v"(A.x), [a1] "v"(A.y), [a2] "v"(A.z), [a3] "v"(A.w), [a4] "v"(B.x), [a5] "v"(B.y), [a6] "v"(B.z),  \
          [a7] "v"(B.w), [b0] "v"(q0), [b1] "v"(q1), [b2] "v"(q2), [b3] "v"(q3), [k2] "v"(k2), [k3] "v"(k3))

// VK: VALU instructions per MFMA (0, 2, 4, 8, 16, 24); LDSOP: operands of the next body by ds_read_b128
template <int VK, int LDSOP>
__global__ __launch_bounds__(512, 2) void loop_kernel(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 132; i += blockDim.x) lds[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-3f - 1.0f;
    __syncthreads();
    const float q0 = seed * lane, q1 = q0 + 1.f, q2 = q0 * 0.5f, q3 = q0 - 3.f;
    int k0 = lane, k1 = lane * 3, k2 = lane * 5, k3 = lane * 7;
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const float4 *row = reinterpret_cast<const float4 *>(&lds[(lane & 31) * 132 + 4 * (lane >> 5)]);
    float4 A = row[0], B = row[33 * 32];
    for (int it = 0; it < iters; ++it) {
        float4 nA = A, nB = B;
        if (LDSOP) { nA = row[2 * ((it + 1) & 15)]; nB = row[2 * ((it + 1) & 15) + 33 * 32]; }
        if (VK == 0) BODY(REP0); else if (VK == 2) BODY(REP1); else if (VK == 4) BODY(REP2); else if (VK == 8) BODY(REP4);
        else if (VK == 16) BODY(REP8); else BODY(REP12);
        A = nA; B = nB;
    }
    float s = (float)(k0 + k1);
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VK, int LDSOP>
static void run(int threads, float *out, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<VK, LDSOP>), dim3(256), dim3(threads), 0, 0, out, iters, 0.731f);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((loop_kernel<VK, LDSOP>), dim3(256), dim3(threads), 0, 0, out, iters, 0.731f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * (threads / 64) * (double)iters * 8 * 4096.0 * reps;
    printf("VALU/MFMA %2d  lds-operands %d  %d waves/SIMD  %7.1f us/launch  %6.1f TFLOP/s (MFMA only)\n", VK, LDSOP, threads / 256,
           1e3 * ms / reps, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    for (int threads = 256; threads <= 512; threads += 256) {
        const int it = threads == 256 ? 2400 : 1200;
        run<0, 0>(threads, out, it); run<2, 0>(threads, out, it); run<4, 0>(threads, out, it); run<8, 0>(threads, out, it);
        run<16, 0>(threads, out, it); run<24, 0>(threads, out, it);
        run<0, 1>(threads, out, it); run<4, 1>(threads, out, it);
    }
    return 0;
}
