// valu_rate.hip -- what a SIMD of this device issues per cycle for plain VALU instruction streams, by instruction kind and by
// waves per SIMD (1, 2, 4, 8): the yardstick of the VALU-bound kernels (descriptors, orientation, detection, scale space).
// Each wave runs `iters` rounds of 32 INDEPENDENT instructions of one kind (8 accumulator chains x 4), no memory traffic.
// Prints wave-instructions per cycle per SIMD at the clock the run sustained (s_memtime / s_memrealtime). Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, int iters, float seed, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    float a[8];
    double d[8];
    int n[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (lane + i); d[i] = seed * (lane + 2 * i); n[i] = lane * 3 + i; p[i] = (f2){a[i], a[i] + 1.f}; }
    const float m = 1.0001f + seed * 1e-9f, c = seed * 1e-3f;
    const double md = 1.0001 + seed * 1e-12, cd = seed * 1e-3;
    const f2 mp = (f2){m, m}, cp = (f2){c, c};
    const unsigned long long smask = __ballot(seed * lane > 3.f), smask2 = __ballot(seed * lane > 7.f);
    unsigned long long sm[2] = {smask, smask2};
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(c) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mp), "v"(cp));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(lane));
                if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(md));
                if (KIND == 7) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 8) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(a[i]));
                if (KIND == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
                if (KIND == 10) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 11) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(lane), "v"(n[(i + 1) & 7]));
                if (KIND == 12) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(n[i]) : "v"(lane), "s"(smask));
                if (KIND == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 16) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (KIND == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(lane), "v"(n[(i + 1) & 7]));
                if (KIND == 18) asm volatile("v_min_i32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 19) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
                if (KIND == 20) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
                if (KIND == 21) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
                if (KIND == 22) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mp));
                if (KIND == 23) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i]));
                if (KIND == 24) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 25) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 26) asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(n[i]) : "v"(a[i]));
                if (KIND == 27) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 28) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 29) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 30) asm volatile("v_or_b32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 31) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(n[i]));
                if (KIND == 33) asm volatile("v_mov_b32 %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if (KIND == 34) asm volatile("v_min_u32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 35) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(lane));
                if (KIND == 36) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 37) asm volatile("v_mac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 38) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 39) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 40) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(lane) : );
                if (KIND == 41) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(n[i]), "v"(lane) : "vcc");
                if (KIND == 42) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 43) asm volatile("v_subrev_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 44) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 45) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 46) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 47) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 48) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if (KIND == 49) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(lane));
                if (KIND == 50) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(n[i]) : "v"(d[i]));
                // round 6: what reading / writing VCC costs (the VOP2 v_cndmask_b32 above issues at 1 / 23 cycles)
                if (KIND == 51) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(lane));
                if (KIND == 52) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %2, %2, %3" : "+v"(n[i]), "+v"(a[i]) : "v"(lane), "v"(c));
                if (KIND == 53) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %2, %2, %3\n\tv_mul_f32 %2, %2, %4\n\tv_sub_f32 %2, %2, %3" : "+v"(n[i]), "+v"(a[i]) : "v"(lane), "v"(c), "v"(m));
                if (KIND == 54) asm volatile("v_subbrev_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n[i]) : : "vcc");
                if (KIND == 55) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(sm[i & 1]) : "v"(a[i]), "v"(c));
                if (KIND == 56) asm volatile("s_and_b64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(lane), "s"(smask), "s"(smask2) : "vcc");
                if (KIND == 57) asm volatile("s_and_b64 %4, %2, %3\n\tv_cndmask_b32_e64 %0, %0, %1, %4" : "+v"(n[i]) : "v"(lane), "s"(smask), "s"(smask2), "s"(sm[0]));
                if (KIND == 58) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(n[i]) : "v"(lane));
                if (KIND == 59) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(n[i]) : "v"(lane), "v"(n[(i + 1) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)n[i] + p[i].x + p[i].y;
    s += (float)(sm[0] + sm[1]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *clk)
{
    printf("%-16s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {                     // waves per SIMD: one workgroup of 256 * wps threads per CU
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps, wg_per_cu = 256 * wps / threads;
        const int iters = 20000 / wps;
        hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256 * wg_per_cu), dim3(threads), 0, 0, out, iters, 0.731f, clk);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256 * wg_per_cu), dim3(threads), 0, 0, out, iters, 0.731f, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        const double ghz = (double)h[0] / ((double)h[1] * 10.0) ;          // s_memrealtime ticks at 100 MHz
        const double insts = (double)iters * 32.0 * wps;                    // wave-instructions per SIMD per launch
        const double cyc = (ms / 5.0) * 1e-3 * ghz * 1e9;
        printf("  %dw: %5.3f/cyc (%4.2f GHz)", wps, insts / cyc, ghz);
    }
    printf("\n");
}

int main(int argc, char **argv)
{
    float *out; unsigned long long *clk;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipMalloc(&out, 2048 * 1024 * 4)); CK(hipMalloc(&clk, 16));
    printf("wave-instructions per cycle per SIMD, by waves per SIMD\n");
    if (argc > 1 && !strcmp(argv[1], "vcc")) {        // round 6: the VCC group only (52 / 53 / 56 / 57 issue 2 / 4 / 2 / 2 instructions per count)
        run<4>("v_cndmask_b32 vop2 vcc", out, clk); run<51>("v_cndmask e64 vcc", out, clk); run<13>("v_cndmask e64 sgpr", out, clk);
        run<58>("v_cndmask vop2 0,v,vcc (new dst)", out, clk); run<59>("v_cndmask vop2 v,v,vcc (new dst)", out, clk);
        run<52>("vop2 cndmask + v_add (x2)", out, clk); run<53>("vop2 cndmask + 3 f32 (x4)", out, clk);
        run<54>("v_subbrev_co_u32 vcc", out, clk); run<9>("v_cmp_gt_f32 -> vcc", out, clk); run<55>("v_cmp_gt_f32_e64 -> sgpr", out, clk);
        run<56>("s_and vcc + vop2 cndmask (x1 valu)", out, clk); run<57>("s_and sgpr + e64 cndmask (x1 valu)", out, clk);
        return 0;
    }
    run<0>("v_fma_f32", out, clk); run<5>("v_add_f32", out, clk); run<12>("v_max3_f32", out, clk); run<3>("v_pk_fma_f32", out, clk);
    run<1>("v_fma_f64", out, clk); run<6>("v_mul_f64", out, clk); run<8>("v_cvt_f64_f32", out, clk);
    run<2>("v_add_u32", out, clk); run<7>("v_lshl_add_u32", out, clk); run<10>("v_mad_i32_i24", out, clk); run<11>("v_med3_i32", out, clk);
    run<4>("v_cndmask_b32", out, clk); run<13>("v_cndmask e64", out, clk); run<9>("v_cmp_gt_f32", out, clk);
    run<14>("v_fmac_f32", out, clk); run<15>("v_mul_f32", out, clk); run<16>("v_sub_f32", out, clk); run<17>("v_and_or_b32", out, clk);
    run<18>("v_min_i32", out, clk); run<19>("v_fmac_f64", out, clk); run<20>("v_add_f64", out, clk); run<21>("v_pk_add_f32", out, clk);
    run<22>("v_pk_mul_f32", out, clk); run<23>("v_cvt_f32_f64", out, clk); run<24>("v_mov_dpp", out, clk); run<25>("v_floor_f32", out, clk);
    run<26>("v_cvt_i32_f32", out, clk);
    run<27>("v_min_f32", out, clk); run<28>("v_max_f32", out, clk); run<29>("v_and_b32", out, clk); run<30>("v_or_b32", out, clk);
    run<31>("v_sub_u32", out, clk); run<32>("v_lshlrev_b32", out, clk); run<33>("v_mov_b32", out, clk); run<34>("v_min_u32", out, clk);
    run<35>("v_xor_b32", out, clk); run<36>("v_fma_f32 acc", out, clk); run<38>("v_add_f32 |abs|", out, clk); run<39>("v_mul_f32 |abs|", out, clk);
    run<41>("v_cmp_lt_i32", out, clk); run<42>("v_sub_f32", out, clk);
    // round 5: the transcendental unit (the gradient's rsq + rcp per pixel, the descriptor weight's ldexp / conversions)
    run<44>("v_rcp_f32", out, clk); run<45>("v_rsq_f32", out, clk); run<46>("v_sqrt_f32", out, clk); run<47>("v_exp_f32", out, clk);
    run<48>("v_rcp_f64", out, clk); run<49>("v_ldexp_f64", out, clk); run<50>("v_cvt_i32_f64", out, clk);
    return 0;
}
