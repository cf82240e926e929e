// lds_atomic.hip -- what `ds_add_f32` does on this device when several lanes of ONE wave instruction add to the same LDS word
// (a wave's private histogram): the ORDER in which the conflicting lanes are applied, the rounding (IEEE add? denormals?), and
// what it costs beside the exec-masked read-add-write rounds the descriptor kernel uses (nm_describe.hip). Diagnostic only.
//   (1) order, read off directly: lane i adds (i + 1) with ds_add_rtn_f32; the returned pre-values are prefix sums and give the
//       serialisation order of every group of lanes that share an address;
//   (2) order, through rounding: random floats over 40 binades, no-return ds_add_f32; final words against the fp32 sum in
//       ascending / descending lane order, bit for bit;
//   (3) special values: denormal + denormal, big + denormal, -0 + +0, inf, NaN;
//   (4) rate: 16 one-wave workgroups per CU (9 792 B of LDS each, like frame_desc_kernel), per iteration the 8 votes of a
//       64-sample pass into a [144 + 9 rows][16 partials] histogram -- (a) 4 exec-masked rounds of 4 ds_read2_b32 + 8 v_add_f32 +
//       4 ds_write2_b32, (b) 8 ds_add_f32 with all 64 lanes -- with F filler fma per vote pass.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lds_off(const void *p) { return (uint32_t)(uintptr_t)p; }

// one wave; trial t: addr[t][lane] (word index < 64), val[t][lane]; out_pre[t][lane] = returned pre-value (rtn form) or nothing;
// out_word[t][64] = final words
__global__ __launch_bounds__(64) void order_kernel(const int *addr, const float *val, float *out_pre, float *out_word, int trials, int rtn)
{
    __shared__ float w[64];
    const int lane = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        w[lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
        const uint32_t a = lds_off(w + addr[t * 64 + lane]);
        const float v = val[t * 64 + lane];
        float pre = 0.f;
        if (rtn) asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pre) : "v"(a), "v"(v) : "memory");
        else asm volatile("ds_add_f32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(a), "v"(v) : "memory");
        __builtin_amdgcn_wave_barrier();
        out_pre[t * 64 + lane] = pre;
        out_word[t * 64 + lane] = w[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int PITCH = 16, ROWS = 144, DUMMY = ROWS * PITCH, LDSF = DUMMY + 9 * PITCH;

template <int MODE, int FILL>
__global__ __launch_bounds__(64) void rate_kernel(float *out, int iters, unsigned seed, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) float part[LDSF];
    const int lane = threadIdx.x, tx = lane & 15, tyg = lane >> 4;
    for (int i = lane; i < LDSF; i += 64) part[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    unsigned s = seed + 977u * blockIdx.x + 31u * (unsigned)tx;            // the four rows of a column share most of the cell
    float f0 = 1.0f + 1e-7f * lane, f1 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int cell = (s >> 8) % 9 + ((s >> 20) & 1) * tyg % 2;         // 0..9: cell base (binx, biny) flattened; rows mostly agree
        const int slot = ((s >> 12) + tyg * (s >> 16)) & 7;
        float *base = part + (cell * 9 + slot) * PITCH + tx;
        float *loc[4] = {base, base + 9 * PITCH, base + 36 * PITCH, base + 45 * PITCH};
        float wt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wt[k] = f0 * (float)(k + 1);
#pragma unroll
        for (int k = 0; k < FILL; ++k) { f0 = __builtin_fmaf(f0, 1.0000001f, f1); f1 = __builtin_fmaf(f1, 0.9999999f, 1e-9f); }
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (tyg == k) {
                    float o[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { o[2 * c] = loc[c][0]; o[2 * c + 1] = loc[c][PITCH]; }
#pragma unroll
                    for (int c = 0; c < 4; ++c) { loc[c][0] = o[2 * c] + wt[2 * c]; loc[c][PITCH] = o[2 * c + 1] + wt[2 * c + 1]; }
                }
                asm volatile("" ::: "memory");
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t a = lds_off(loc[c]);
                asm volatile("ds_add_f32 %0, %1" : : "v"(a), "v"(wt[2 * c]) : "memory");
                asm volatile("ds_add_f32 %0, %1 offset:64" : : "v"(a), "v"(wt[2 * c + 1]) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_wave_barrier();
    float acc = f0 + f1;
    for (int i = lane; i < LDSF; i += 64) acc += part[i];
    out[blockIdx.x * 64 + lane] = acc;
    if (blockIdx.x == 0 && lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int FILL>
static void run_rate(const char *name, float *out, unsigned long long *clk)
{
    const int iters = 4000, blocks = 256 * 16;
    hipLaunchKernelGGL((rate_kernel<MODE, FILL>), dim3(blocks), dim3(64), 0, 0, out, iters, 12345u, clk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rate_kernel<MODE, FILL>), dim3(blocks), dim3(64), 0, 0, out, iters, 12345u + r, clk);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    printf("%-44s %8.1f us per launch = %7.1f cycles per vote pass per wave-slot (4 waves per SIMD), clock %.2f GHz\n", name,
           ms / 5 * 1e3, ms / 5 * 1e-3 * ghz * 1e9 / iters, ghz);
}

int main()
{
    const int T = 4096;
    std::vector<int> addr(T * 64);
    std::vector<float> val(T * 64), pre(T * 64), word(T * 64);
    int *d_addr; float *d_val, *d_pre, *d_word;
    CK(hipMalloc(&d_addr, T * 64 * 4)); CK(hipMalloc(&d_val, T * 64 * 4)); CK(hipMalloc(&d_pre, T * 64 * 4)); CK(hipMalloc(&d_word, T * 64 * 4));
    srand(7);
    // (1) order through returned pre-values
    for (int t = 0; t < T; ++t) {
        const int K = 1 + t % 9;                               // 1..9 distinct words: 7- to 64-way conflicts
        for (int l = 0; l < 64; ++l) { addr[t * 64 + l] = (rand() % K) * 7 % 64; val[t * 64 + l] = (float)(l + 1); }
    }
    CK(hipMemcpy(d_addr, addr.data(), T * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), T * 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_val, d_pre, d_word, T, 1);
    CK(hipMemcpy(pre.data(), d_pre, T * 64 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(word.data(), d_word, T * 64 * 4, hipMemcpyDeviceToHost));
    long asc = 0, desc = 0, other = 0, badsum = 0;
    int shown = 0;
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < 64; ++a) {
            std::vector<int> g;
            for (int l = 0; l < 64; ++l) if (addr[t * 64 + l] == a) g.push_back(l);
            if (g.size() < 2) continue;
            float up = 0.f, dn = 0.f; bool is_asc = true, is_desc = true;
            for (size_t i = 0; i < g.size(); ++i) { if (pre[t * 64 + g[i]] != up) is_asc = false; up += (float)(g[i] + 1); }
            for (size_t i = g.size(); i-- > 0;) { if (pre[t * 64 + g[i]] != dn) is_desc = false; dn += (float)(g[i] + 1); }
            if (word[t * 64 + a] != up) ++badsum;
            if (is_asc) ++asc; else if (is_desc) ++desc; else {
                ++other;
                if (shown < 4) { ++shown; printf("  group (trial %d, word %d), lane:pre =", t, a); for (int l : g) printf(" %d:%g", l, pre[t * 64 + l]); printf("\n"); }
            }
        }
    printf("(1) ds_add_rtn_f32, conflicting groups: applied in ASCENDING lane order %ld, descending %ld, other %ld; wrong totals %ld\n", asc, desc, other, badsum);
    // (2) order through rounding, no-return form
    for (int t = 0; t < T; ++t) {
        const int K = 1 + t % 9;
        for (int l = 0; l < 64; ++l) {
            addr[t * 64 + l] = (rand() % K) * 5 % 64;
            val[t * 64 + l] = ldexpf((float)rand() / RAND_MAX + 0.5f, rand() % 40 - 20) * ((t & 1) && (rand() & 1) ? -1.f : 1.f);
        }
    }
    CK(hipMemcpy(d_addr, addr.data(), T * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), T * 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_val, d_pre, d_word, T, 0);
    CK(hipMemcpy(word.data(), d_word, T * 64 * 4, hipMemcpyDeviceToHost));
    long eq_asc = 0, eq_desc = 0, words = 0, differ = 0;
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < 64; ++a) {
            volatile float up = 0.f, dn = 0.f; int n = 0;
            for (int l = 0; l < 64; ++l) if (addr[t * 64 + l] == a) { up = up + val[t * 64 + l]; ++n; }
            for (int l = 63; l >= 0; --l) if (addr[t * 64 + l] == a) dn = dn + val[t * 64 + l];
            if (n < 2) continue;
            ++words;
            float u = up, d = dn, g = word[t * 64 + a];
            if (memcmp(&u, &d, 4)) ++differ;
            if (!memcmp(&g, &u, 4)) ++eq_asc;
            if (!memcmp(&g, &d, 4)) ++eq_desc;
        }
    printf("(2) ds_add_f32 (no return), %ld conflicting words (%ld where the two orders differ): equal to the ascending-lane fp32 sum %ld, "
           "to the descending %ld\n", words, differ, eq_asc, eq_desc);
    // (3) special values: lanes 0 and 1 add to word 0 etc.
    {
        const float cases[][2] = {{1e-40f, 1e-40f}, {1.0f, 1e-40f}, {-0.0f, 0.0f}, {-0.0f, -0.0f}, {INFINITY, 1.f}, {INFINITY, -INFINITY},
                                  {NAN, 1.f}, {1.17549435e-38f, -1.1754942e-38f}, {3e38f, 3e38f}, {1.0f, 5.9604645e-8f}, {1.0f, 5.9604652e-8f}};
        const int nc = sizeof(cases) / sizeof(cases[0]);
        for (int t = 0; t < nc; ++t)
            for (int l = 0; l < 64; ++l) { addr[t * 64 + l] = l < 2 ? 0 : l; val[t * 64 + l] = l < 2 ? cases[t][l] : 0.f; }
        CK(hipMemcpy(d_addr, addr.data(), nc * 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), nc * 64 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, d_addr, d_val, d_pre, d_word, nc, 0);
        CK(hipMemcpy(word.data(), d_word, nc * 64 * 4, hipMemcpyDeviceToHost));
        printf("(3) special values (0 + a + b by ds_add_f32 | host fp32):\n");
        for (int t = 0; t < nc; ++t) {
            volatile float h = 0.f; h = h + cases[t][0]; h = h + cases[t][1];
            float hh = h; uint32_t gb, hb; memcpy(&gb, &word[t * 64], 4); memcpy(&hb, &hh, 4);
            printf("    %-14g + %-14g -> %-14g (%08x) | %-14g (%08x) %s\n", cases[t][0], cases[t][1], word[t * 64], gb, hh, hb, gb == hb ? "" : (std::isnan(hh) && std::isnan(word[t * 64]) ? "(NaN both)" : "DIFFERENT"));
        }
    }
    // (4) rates
    float *out; unsigned long long *clk;
    CK(hipMalloc(&out, 256 * 16 * 64 * 4)); CK(hipMalloc(&clk, 16));
    printf("(4) vote pass of 64 samples x 8 votes, 16 one-wave workgroups per CU:\n");
    run_rate<0, 0>("4 masked rounds read2/add/write2, no filler", out, clk);
    run_rate<1, 0>("8 x ds_add_f32 all lanes, no filler", out, clk);
    run_rate<0, 80>("4 masked rounds, 160 filler fma", out, clk);
    run_rate<1, 80>("8 x ds_add_f32, 160 filler fma", out, clk);
    run_rate<0, 40>("4 masked rounds, 80 filler fma", out, clk);
    run_rate<1, 40>("8 x ds_add_f32, 80 filler fma", out, clk);
    return 0;
}
