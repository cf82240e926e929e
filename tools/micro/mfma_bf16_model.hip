// mfma_bf16_model.hip -- how v_mfma_f32_32x32x16_bf16 lays out its operands and how it rounds, probed on the device the
// matcher's bf16x3 screen runs on. The screen's error bound needs an upper bound on the rounding
// error of one instruction  D = C + sum_{k<16} a_k b_k ; the probes below distinguish the candidate hardware models:
//   (L) layout: lane l supplies row/column l % 32, k = 8 (l / 32) .. + 7; D register e of lane l = row (e&3) + 8 (e>>2) +
//       4 (l / 32), column l % 32  -- checked with integer-valued operands (every sum exact);
//   (1) C = 1, sixteen products of 2^-25: a chain that adds the products to C one at a time with RN returns 1, a dot
//       product formed first (exactly, or in a tree) returns 1 + 2^-21;
//   (2) C = 1, one product 2^-24 (1 + 2^-6): RN gives 1 + 2^-23, truncation gives 1;
//   (3) C = 0, products {1, 15 x 2^-25}: exact-then-round gives 1 + 2^-21 (3.75 ulp -> 4), anything sequential gives 1;
//   (4) C = 2^24, products {16 x 1}: exact-then-round gives 2^24 + 16; a chain of RN adds stays at 2^24;
//   (5) random operands: largest observed |D - exact| / (|C| + sum |a_k b_k|) in units of 2^-24 over 2^20 instructions.
// Diagnostic only; nothing in the product links it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_model.hip -o niftymatch_amd/lib/mfma_bf16_model
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one instruction per wave: A (32 x 16, row-major bf16 bits), B (16 x 32 given as Bt: 32 columns x 16), C and D 32 x 32
__global__ __launch_bounds__(64) void one_mfma(const uint16_t *A, const uint16_t *Bt, const float *C, float *D, int n)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const uint16_t *a = A + (size_t)t * 512, *b = Bt + (size_t)t * 512;
        const float *c = C + (size_t)t * 1024;
        bf16x8 fa, fb;
        for (int k = 0; k < 8; ++k) { fa[k] = (short)a[r * 16 + 8 * h + k]; fb[k] = (short)b[r * 16 + 8 * h + k]; }
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = c[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) D[(size_t)t * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
    }
}

static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }          // exact inputs only
static float fb(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }

int main()
{
    const int NR = 1 << 10;                        // random instructions (x 1024 outputs each = 2^20 results)
    const int NP = 4 * 32;                        // alignment probes (6)
    const int N = 8 + NR + NP;
    std::vector<uint16_t> A((size_t)N * 512, 0), Bt((size_t)N * 512, 0);
    std::vector<float> C((size_t)N * 1024, 0.f), D((size_t)N * 1024, 0.f);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    // t = 0: layout, integers in [-8, 8)
    for (int i = 0; i < 512; ++i) { A[i] = bf((float)((int)(rnd() % 16) - 8)); Bt[i] = bf((float)((int)(rnd() % 16) - 8)); }
    for (int i = 0; i < 1024; ++i) C[i] = (float)((int)(rnd() % 64) - 32);
    auto fill = [&](int t, const float *av, const float *bv, float c) {      // every row / column the same 16 values
        for (int m = 0; m < 32; ++m) for (int k = 0; k < 16; ++k) { A[(size_t)t * 512 + m * 16 + k] = bf(av[k]); Bt[(size_t)t * 512 + m * 16 + k] = bf(bv[k]); }
        for (int i = 0; i < 1024; ++i) C[(size_t)t * 1024 + i] = c;
    };
    float av[16], bv[16];
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -13); bv[k] = ldexpf(1.f, -12); }
    fill(1, av, bv, 1.0f);
    for (int k = 0; k < 16; ++k) { av[k] = 0.f; bv[k] = 0.f; }
    av[0] = ldexpf(1.f + ldexpf(1.f, -6), -12); bv[0] = ldexpf(1.f, -12);
    fill(2, av, bv, 1.0f);
    av[5] = av[0]; bv[5] = bv[0]; av[0] = bv[0] = 0.f;                        // the same product at k = 5 and at k = 13
    fill(6, av, bv, 1.0f);
    av[13] = av[5]; bv[13] = bv[5]; av[5] = bv[5] = 0.f;
    fill(7, av, bv, 1.0f);
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -13); bv[k] = ldexpf(1.f, -12); }
    av[0] = 1.f; bv[0] = 1.f;
    fill(3, av, bv, 0.0f);
    for (int k = 0; k < 16; ++k) { av[k] = 1.f; bv[k] = 1.f; }
    fill(4, av, bv, 16777216.0f);
    // t = 5: as (3) with the large product in the LAST slot
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -13); bv[k] = ldexpf(1.f, -12); }
    av[15] = 1.f; bv[15] = 1.f;
    fill(5, av, bv, 0.0f);
    for (int t = 8; t < 8 + NR; ++t) {
        for (int i = 0; i < 512; ++i) {
            A[(size_t)t * 512 + i] = (uint16_t)((rnd() & 0x807F) | ((120 + rnd() % 14) << 7));
            Bt[(size_t)t * 512 + i] = (uint16_t)((rnd() & 0x807F) | ((120 + rnd() % 14) << 7));
        }
        for (int i = 0; i < 1024; ++i) { const float c = fb((uint16_t)((rnd() & 0x807F) | ((122 + rnd() % 14) << 7))); C[(size_t)t * 1024 + i] = c * (1.f + (float)(rnd() & 0xFFFF) * 1e-6f); }
    }
    // (6) product 1 at k = 0 plus ONE product +-(255/128)^2 2^-j at k = 1 (same half) or k = 8 (other half), C = 0
    for (int q = 0; q < NP; ++q) {
        const int j = q & 31, slot = (q & 32) ? 8 : 1; const float sg = (q & 64) ? -1.f : 1.f;
        for (int k = 0; k < 16; ++k) { av[k] = 0.f; bv[k] = 0.f; }
        av[0] = bv[0] = 1.f;
        av[slot] = sg * 255.f / 128.f; bv[slot] = ldexpf(255.f / 128.f, -j);
        fill(8 + NR + q, av, bv, 0.0f);
    }
    uint16_t *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, Bt.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one_mfma, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, N);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));

    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double e = C[m * 32 + n];
        for (int k = 0; k < 16; ++k) e += (double)fb(A[m * 16 + k]) * (double)fb(Bt[n * 16 + k]);
        if ((double)D[m * 32 + n] != e) ++bad;
    }
    printf("(L) layout probe: %d of 1024 outputs differ from the assumed layout\n", bad);
    printf("(1) C=1 + 16 x 2^-25          : D = 1 + %g ulp   (0: chain of RN adds from C; 4: dot product formed first)\n", (D[1024] - 1.0) / ldexp(1.0, -23));
    printf("(2) C=1 + 2^-24(1+2^-6) @k=0  : D = 1 + %g ulp   (1: round to nearest; 0: truncation)\n", (D[2048] - 1.0) / ldexp(1.0, -23));
    printf("(2') the same product @k=5    : D = 1 + %g ulp\n", (D[6 * 1024] - 1.0) / ldexp(1.0, -23));
    printf("(2'') the same product @k=13  : D = 1 + %g ulp\n", (D[7 * 1024] - 1.0) / ldexp(1.0, -23));
    printf("(3) 1 @k=0 + 15 x 2^-25       : D = 1 + %g ulp   (4: exact sum rounded once; 0: sequential)\n", (D[3 * 1024] - 1.0) / ldexp(1.0, -23));
    printf("(3') 1 @k=15 + 15 x 2^-25     : D = 1 + %g ulp\n", (D[5 * 1024] - 1.0) / ldexp(1.0, -23));
    printf("(4) C=2^24 + 16 x 1           : D = 2^24 + %g     (16: exact sum rounded once; 0: chain of RN adds)\n", (double)D[4 * 1024] - 16777216.0);
    double worst = 0.0, worst_signed_lo = 0.0, worst_signed_hi = 0.0;
    for (int t = 8; t < 8 + NR; ++t)
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            double e = C[(size_t)t * 1024 + m * 32 + n], mag = fabs(e);
            for (int k = 0; k < 16; ++k) {
                const double p = (double)fb(A[(size_t)t * 512 + m * 16 + k]) * (double)fb(Bt[(size_t)t * 512 + n * 16 + k]);
                e += p; mag += fabs(p);
            }
            const double err = ((double)D[(size_t)t * 1024 + m * 32 + n] - e) / mag / ldexp(1.0, -24);
            if (fabs(err) > worst) worst = fabs(err);
            if (err < worst_signed_lo) worst_signed_lo = err;
            if (err > worst_signed_hi) worst_signed_hi = err;
        }
    printf("(5) random operands, %d results: max |D - exact| / (|C| + sum |a b|) = %.3f x 2^-24 (signed range %.3f .. %.3f)\n",
           NR * 1024, worst, worst_signed_lo, worst_signed_hi);
    printf("(6) 1 @k=0 + s (255/128)^2 2^-j @k=slot, C=0: (D - exact) in units of 2^-23; RN of the exact sum would give |.| <= 0.5\n");
    for (int q = 0; q < NP; ++q) {
        const int j = q & 31, slot = (q & 32) ? 8 : 1; const double sg = (q & 64) ? -1.0 : 1.0;
        const double exact = 1.0 + sg * (255.0 / 128.0) * (255.0 / 128.0) * ldexp(1.0, -j);
        if (j == 0) printf("    slot %d sign %+d:", slot, (int)sg);
        printf(" %.2f", ((double)D[(size_t)(8 + NR + q) * 1024] - exact) / ldexp(1.0, -23) * (exact >= 2.0 ? 0.5 : exact >= 4.0 ? 0.25 : 1.0));
        if (j == 31) printf("\n");
    }
    return bad ? 1 : 0;
}
