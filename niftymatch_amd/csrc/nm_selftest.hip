// nm_selftest.hip -- on-device self-test of the matrix-pipe premises the matcher's proofs rest on.
// The matcher's MFMA screens only SELECT candidates; what makes the result exact for every input is a bound
// |screen value - exact distance| <= E that match_finalize_kernel evaluates per row (nm_match.hip, screen_err_coeff). The
// part of E that covers the fp32 accumulation INSIDE the matrix instructions is not arithmetic one can derive from IEEE
// rules: it is a model of the hardware (H-bf16: each 8-product half is aligned to its largest product and truncated below
// 2^-24 of it, the halves and C are then added and rounded to nearest). This file measures that model on the device the
// library runs on, for both instructions the screens issue, so that the premise is a TEST (tests/test_gpu_match.py) and
// not a reading of one probe run (protects the scan of kernels/match.cu:83-117, whose results the finalize pass must equal):
//   instruction 0 = v_mfma_f32_32x32x16_bf16 (bf16x3 screen, second pass of the two-stage screen, every norm k-slot)
//   instruction 1 = v_mfma_f32_32x32x16_f16  (coarse pass of the two-stage screen, the default)
// Part 1, one instruction D = C + sum_{k<16} a_k b_k: operand/accumulator layout (integer operands: every sum exact), the
// directed cases that tell the candidate models apart, and n random instructions with
//     rel_u   = max |D - exact| / (u (|C| + sum |a_k b_k|)),                         u = 2^-24
//     model   = max |D - exact| / (u |exact| + 7 u (pmax_lo + pmax_hi)),             pmax = largest |product| of a half
// (model <= 1 means the hardware is inside H-bf16; the kernels' constants assume 2).
// Part 2, the chain exactly as the kernels issue it -- one bf16 norm k-slot instruction with C = 0, then 8 f16 instructions
// (coarse pass) or 24 bf16 instructions in (hi.hi, hi.lo, lo.hi) order per k-step (bf16x3) into the same accumulator -- on
// adversarial row families (constant vectors, aligned rounding residuals, near-duplicates with massive cancellation, 24
// binades of magnitudes, fp16-SUBNORMAL elements alone and mixed with normal ones: a pipe that flushed them would be off by
// half of (sqrt na + sqrt nb)^2 on the duplicates), against a compensated binary64 evaluation of the SAME operand images:
//     chain   = max |value - exact(images)| / (sqrt na + sqrt nb)^2
// which is the quantity screen_err_coeff budgets (the representation error of the images is bounded separately, from
// measured residual norms, and is not part of this premise).
#include "nm_common.hpp"
#include "../../include/nm_abi.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

template <int INSTR>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c)
{
    if (INSTR == 1)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned rng(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(s >> 33);
}

// 16-bit operand encodings (exact conversions only: the values are made from their bit fields)
template <int INSTR>
__device__ __forceinline__ float op_value(unsigned bits)
{
    if (INSTR == 1) return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
    return __uint_as_float(bits << 16);
}
template <int INSTR>
__device__ __forceinline__ unsigned op_bits_exact(float x)          // x must be representable
{
    if (INSTR == 1) return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x);
    return __float_as_uint(x) >> 16;
}
template <int INSTR>
__device__ __forceinline__ unsigned op_random(unsigned long long &s)   // 14 binades around 1, full mantissa, random sign
{
    const unsigned a = rng(s), e = rng(s) % 14u;
    if (INSTR == 1) return (a & 0x83FFu) | ((9u + e) << 10);           // fp16: exponent field 9..22 = 2^-6 .. 2^7
    return (a & 0x807Fu) | ((120u + e) << 7);                          // bf16: exponent field 120..133
}

__device__ __forceinline__ void atomic_max_pos(float *p, float v)    // v >= 0: the bit patterns order like the values
{
    atomicMax(reinterpret_cast<int *>(p), __float_as_int(v));
}

// Outputs of one accumulator register e of lane l: row (e & 3) + 8 (e >> 2) + 4 (l / 32) of A, column l % 32 of B.
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// One wave per block. sA / sB: the 32 x 16 operand matrices as binary32 values (row r, k), C in registers.
template <int INSTR>
__device__ void run_one(const unsigned (*bA)[16], const unsigned (*bB)[16], const f32x16 &c, f32x16 &d)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    u32x4 fa, fb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        fa[j] = bA[r][8 * h + 2 * j] | (bA[r][8 * h + 2 * j + 1] << 16);
        fb[j] = bB[r][8 * h + 2 * j] | (bB[r][8 * h + 2 * j + 1] << 16);
    }
    d = mfma16<INSTR>(fa, fb, c);
}

template <int INSTR>
__global__ __launch_bounds__(64) void selftest_instr_kernel(float *__restrict__ out, int iters)
{
    __shared__ float sA[32][17], sB[32][17];
    __shared__ unsigned bA[32][16], bB[32][16];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned long long s = 0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * 64 + lane + 1) + (INSTR ? 77 : 0);
    const double U = 5.9604644775390625e-08;     // 2^-24

    auto set_all = [&](int k, float av, float bv) {          // every row / column the same value at k-slot k (lane r < 32 writes)
        if (h == 0) { sA[r][k] = av; sB[r][k] = bv; bA[r][k] = op_bits_exact<INSTR>(av); bB[r][k] = op_bits_exact<INSTR>(bv); }
    };
    auto exact_of = [&](int m, int n, float cv, double &mag, double &pa, double &pb) {
        double e = (double)cv;
        mag = fabs(e); pa = 0.0; pb = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double p = (double)sA[m][k] * (double)sB[n][k];
            e += p; mag += fabs(p);
            if (k < 8) pa = fmax(pa, fabs(p)); else pb = fmax(pb, fabs(p));
        }
        return e;
    };

    if (blockIdx.x == 0) {
        f32x16 c, d;
        // (L) layout: integers in [-8, 8), C in [-32, 32): every partial sum is exact in binary32
        for (int k = 8 * h; k < 8 * h + 8; ++k) {
            const float av = (float)((int)(rng(s) % 16u) - 8), bv = (float)((int)(rng(s) % 16u) - 8);
            sA[r][k] = av; sB[r][k] = bv; bA[r][k] = op_bits_exact<INSTR>(av); bB[r][k] = op_bits_exact<INSTR>(bv);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = (float)((int)(rng(s) % 64u) - 32);
        __syncthreads();
        run_one<INSTR>(bA, bB, c, d);
        int bad = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            double mag, pa, pb;
            if ((double)d[e] != exact_of(acc_row(e, h), r, c[e], mag, pa, pb)) ++bad;
        }
        if (bad) atomicAdd(&out[0], (float)bad);
        __syncthreads();
        // directed cases (all rows / columns equal, so every output is the same number; lane 0 reports register 0)
        auto directed = [&](int which, float cval) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = cval;
            run_one<INSTR>(bA, bB, c, d);
            if (lane == 0) out[which] = (which == 4) ? (float)((double)d[0] - 16777216.0) : (float)(((double)d[0] - 1.0) / 1.1920928955078125e-07);
            __syncthreads();
        };
        for (int k = 0; k < 16; ++k) set_all(k, 0x1p-13f, 0x1p-12f);
        directed(1, 1.0f);                                               // (1) C = 1 + 16 x 2^-25
        for (int k = 0; k < 16; ++k) set_all(k, 0.f, 0.f);
        set_all(0, 0x1p-12f * (1.f + 0x1p-6f), 0x1p-12f);
        directed(2, 1.0f);                                               // (2) C = 1 + 2^-24 (1 + 2^-6)
        for (int k = 0; k < 16; ++k) set_all(k, 0x1p-13f, 0x1p-12f);
        set_all(0, 1.f, 1.f);
        directed(3, 0.0f);                                               // (3) 1 @k=0 + 15 x 2^-25
        for (int k = 0; k < 16; ++k) set_all(k, 1.f, 1.f);
        directed(4, 16777216.0f);                                        // (4) C = 2^24 + 16 x 1
        // (6) same-half truncation: 1 @k=0 + (255/128)^2 2^-25 @k=1: RN of the exact sum is 1 + 2^-23 (the small product is
        // 0.992 ulp); H-bf16 cuts it to 2^-24 first and the tie goes to even: 1
        for (int k = 0; k < 16; ++k) set_all(k, 0.f, 0.f);
        set_all(0, 1.f, 1.f);
        if (INSTR == 1) set_all(1, (255.f / 128.f) * 0x1p-11f, (255.f / 128.f) * 0x1p-14f);   // fp16: both factors normal
        else set_all(1, 255.f / 128.f, (255.f / 128.f) * 0x1p-25f);
        directed(11, 0.0f);
    }
    __syncthreads();

    float worst_rel = 0.f, worst_model = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int k = 8 * h; k < 8 * h + 8; ++k) {
            const unsigned a = op_random<INSTR>(s), b = op_random<INSTR>(s);
            bA[r][k] = a; bB[r][k] = b; sA[r][k] = op_value<INSTR>(a); sB[r][k] = op_value<INSTR>(b);
        }
        f32x16 c, d;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            // C: 16 random mantissa bits over 14 binades around the products' sums, or exactly 0 (the chains start from 0)
            const unsigned a = rng(s), ee = rng(s) % 16u;
            c[e] = (ee >= 14u) ? 0.f : __uint_as_float(((a & 0x1u) << 31) | ((122u + ee) << 23) | ((a >> 8) << 7 & 0x7FFF80u));
        }
        __syncthreads();
        run_one<INSTR>(bA, bB, c, d);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            double mag, pa, pb;
            const double ex = exact_of(acc_row(e, h), r, c[e], mag, pa, pb);
            const double err = fabs((double)d[e] - ex);
            worst_rel = fmaxf(worst_rel, (float)(err / (U * mag)));
            worst_model = fmaxf(worst_model, (float)(err / (U * fabs(ex) + 7.0 * U * (pa + pb) + 1e-300)));
        }
        __syncthreads();
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        worst_rel = fmaxf(worst_rel, __shfl_xor(worst_rel, dd));
        worst_model = fmaxf(worst_model, __shfl_xor(worst_model, dd));
    }
    if (lane == 0) {
        atomic_max_pos(&out[5], worst_rel);
        atomic_max_pos(&out[6], worst_model);
        atomicAdd(&out[7], (float)iters);
    }
}

// ---- part 2: the chains as the kernels issue them ----
constexpr int DIM = 128;
constexpr int NFAM = 8;

__device__ __forceinline__ unsigned bf16_rne_bits(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_three_bits(float n, unsigned &h, unsigned &m, unsigned &l)
{
    const unsigned uh = __float_as_uint(n) & 0xFFFF0000u;
    const float r1 = n - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = uh >> 16; m = um >> 16; l = __float_as_uint(r2) >> 16;
}

// element k of row `row` (0..31 = queries a, 32..63 = candidates b) of family `fam`; partner = the same element of the
// query row with the same index (candidates of the duplicate families are built from it)
__device__ float family_value(int fam, int row, int k, unsigned long long &s, float partner, float row_c)
{
    const unsigned a = rng(s);
    const float u01 = (float)(a & 0xFFFFFFu) * 0x1p-24f;
    const bool cand = row >= 32;
    switch (fam) {
        case 0: return u01;                                                          // Uniform[0, 1)
        case 1: return cand ? partner + 1e-3f * (u01 - 0.5f) : u01;                  // near-duplicates on the diagonal
        case 2: return row_c;                                                        // constant vectors
        case 3: return ldexpf(1.f + 0x1p-11f - 0x1p-23f, (row & 7) - 3);   // just below an fp16 midpoint, one sign
        case 4: return ((a >> 30) & 1u ? -1.f : 1.f) * ldexpf(1.f + u01, (int)(rng(s) % 24u) - 20);   // 24 binades
        case 5: return cand ? partner : ldexpf((float)(1u + (a & 0xFFu)), -24);      // all fp16-subnormal (|2a| < 2^-14), duplicates
        case 6: {                                                                    // normal and subnormal elements mixed
            const float v = (k & 1) ? ldexpf((float)(1u + (a & 0xFFu)), -24) : u01 * 0x1p-10f;
            return cand ? partner + ((k & 1) ? 0.f : 1e-7f * (u01 - 0.5f)) : v;
        }
        default: return 2.0e3f * u01 * (((a >> 30) & 1u) ? -1.f : 1.f);              // squared norms up to ~1.7e8: the top of the coarse pass's domain
    }
}

template <int INSTR>
__global__ __launch_bounds__(64) void selftest_chain_kernel(float *__restrict__ out, int iters)
{
    // images: INSTR 1: [row][k] fp16 of (-2 a) / b; INSTR 0: [row][k] bf16 hi, then lo
    __shared__ unsigned short img[64][DIM + 2], img_lo[INSTR == 0 ? 64 : 1][DIM + 2];
    __shared__ float s_norm[64], s_partner[32][DIM + 1];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned long long s = 0xD1B54A32D192ED03ull * ((unsigned long long)blockIdx.x * 64 + lane + 1) + (INSTR ? 991 : 0);
    float worst = 0.f, worst_sub = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int fam = (blockIdx.x + it) % NFAM;
        // one row per lane: lanes 0..31 the queries, then (after a barrier: they read their partner) lanes 32..63 the candidates
        for (int pass = 0; pass < 2; ++pass) {
            if (h == pass) {
                const float row_c = ldexpf(1.f + (float)(rng(s) & 0xFFFFu) * 0x1p-16f, (int)(rng(s) % 7u) - 3);
                double nrm = 0.0;
                for (int k = 0; k < DIM; ++k) {
                    const float x = family_value(fam, lane, k, s, pass ? s_partner[r][k] : 0.f, row_c);
                    if (!pass) s_partner[r][k] = x;
                    nrm += (double)x * (double)x;
                    const float sx = pass ? x : -2.0f * x;
                    if (INSTR == 1) {
                        img[lane][k] = __builtin_bit_cast(unsigned short, (_Float16)sx);
                    } else {
                        const unsigned hi = bf16_rne_bits(sx);
                        img[lane][k] = (unsigned short)hi;
                        img_lo[lane][k] = (unsigned short)bf16_rne_bits(sx - __uint_as_float(hi << 16));
                    }
                }
                s_norm[lane] = (float)nrm;
            }
            __syncthreads();
        }
        // the chain: norm k-slots (candidate: nb_h, nb_m, nb_l, 1, 1, 1, 0, 0; query, lanes of k = 0..7 only: 1, 1, 1, na_h, na_m,
        // na_l, 0, 0), then the k-steps t = 0..7 (k = 16 t + 8 h .. + 7), as f16_slots / f16_kstep / bf16_kstep issue them
        unsigned nh, nm, nl;
        bf16_three_bits(s_norm[32 + r], nh, nm, nl);
        const u32x4 cslot = {nh | (nm << 16), nl | (0x3F80u << 16), 0x3F80u | (0x3F80u << 16), 0u};
        u32x4 qslot = {0u, 0u, 0u, 0u};
        if (h == 0) {
            bf16_three_bits(s_norm[r], nh, nm, nl);
            qslot = (u32x4){0x3F80u | (0x3F80u << 16), 0x3F80u | (nh << 16), nm | (nl << 16), 0u};
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = mfma16<0>(cslot, qslot, acc);
        auto frag = [&](const unsigned short (*im)[DIM + 2], int row, int t) {
            u32x4 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = (unsigned)im[row][16 * t + 8 * h + 2 * j] | ((unsigned)im[row][16 * t + 8 * h + 2 * j + 1] << 16);
            return f;
        };
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (INSTR == 1) {
                acc = mfma16<1>(frag(img, 32 + r, t), frag(img, r, t), acc);
            } else {
                const u32x4 ch = frag(img, 32 + r, t), cl = frag(img_lo, 32 + r, t), qh = frag(img, r, t), ql = frag(img_lo, r, t);
                acc = mfma16<0>(ch, qh, acc);
                acc = mfma16<0>(ch, ql, acc);
                acc = mfma16<0>(cl, qh, acc);
            }
        }
        // exact value of the same operand images: n^b + n^a + sum of the products, Neumaier-compensated in binary64
#pragma unroll 1
        for (int e = 0; e < 16; ++e) {
            const int m = 32 + acc_row(e, h), n = r;              // candidate row m, query row n
            double sum = (double)s_norm[m], comp = 0.0;
            auto add = [&](double p) {
                const double t = sum + p;
                comp += (fabs(sum) >= fabs(p)) ? (sum - t) + p : (p - t) + sum;
                sum = t;
            };
            add((double)s_norm[n]);
            for (int k = 0; k < DIM; ++k) {
                const double bh = (double)op_value<INSTR>(img[m][k]), ah = (double)op_value<INSTR>(img[n][k]);
                add(bh * ah);
                if (INSTR == 0) {
                    const double bl = (double)op_value<0>(img_lo[m][k]), al = (double)op_value<0>(img_lo[n][k]);
                    add(bh * al);
                    add(bl * ah);
                }
            }
            const double exact = sum + comp;
            const double sn = sqrt((double)s_norm[m]) + sqrt((double)s_norm[n]);
            const float metric = (float)(fabs((double)acc[e] - exact) / (sn * sn));
            worst = fmaxf(worst, metric);
            if (fam == 5 || fam == 6) worst_sub = fmaxf(worst_sub, metric);
        }
        __syncthreads();
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        worst = fmaxf(worst, __shfl_xor(worst, dd));
        worst_sub = fmaxf(worst_sub, __shfl_xor(worst_sub, dd));
    }
    if (lane == 0) {
        atomic_max_pos(&out[8], worst);
        atomic_max_pos(&out[9], worst_sub);
        atomicAdd(&out[10], (float)iters);
    }
}

// ---- the fp32 instruction: v_mfma_f32_32x32x2_f32 (the matcher's fp32 screen and the materialised distance pass) ----
// Premise of both bounds: one instruction D = C + a0 b0 + a1 b1 behaves like two fused steps -- exact
// products, at most two roundings -- so that a chain of n instructions is a 2n-step fma chain.
//   out[0] = max |D - exact| / (u (|C| + |a0 b0| + |a1 b1|)) over random instructions (<= 2: two roundings; <= 1: one)
//   out[1] / out[2] = instructions whose result equals fma(a1, b1, fma(a0, b0, C)) / the correctly rounded exact sum
//   out[3] = the distance pass's form -- two chains (k = 0..63 behind the norm pair, k = 64..127 from 0) and one add -- on
//            row families (uniform, constant, near-duplicates, mixed binades): max |v - exact of the same operands| /
//            (sqrt na + sqrt nb)^2, to be held against DIST_C (nm_match.hip; 5.15e-6 incl. the norms' own 8 roundings)
//   out[4] = the fp32 screen's form (one 130-step chain), same ratio, against nm_sift_match_accum_budget(0)
//   out[5] = random instructions run
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void selftest_f32_kernel(float *__restrict__ out, int iters)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned long long s = 0xD1B54A32D192ED03ull * ((unsigned long long)blockIdx.x * 64 + lane + 1);
    const double U = 5.9604644775390625e-08;
    __shared__ float sA[32][2], sB[32][2];
    auto rnd_f = [&](int binades) {
        const unsigned a = rng(s), e = rng(s) % (unsigned)binades;
        return __uint_as_float((a & 0x807FFFFFu) | ((120u + e) << 23));
    };
    float worst = 0.f;
    unsigned n_chain = 0, n_rn = 0, n_all = 0;
    for (int it = 0; it < iters; ++it) {
        const float av = rnd_f(12), bv = rnd_f(12);           // this lane's A operand (row r, k = h) and B operand (column r, k = h)
        sA[r][h] = av; sB[r][h] = bv;
        f32x16s c, d;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = (rng(s) & 3u) ? rnd_f(14) : 0.0f;
        __syncthreads();
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = acc_row(e, h);
            const double p0 = (double)sA[m][0] * (double)sB[r][0], p1 = (double)sA[m][1] * (double)sB[r][1];
            const double ex = ((double)c[e] + p0) + p1;       // (three binary64 terms: good to 2^-52 of their magnitude)
            const double mag = fabs((double)c[e]) + fabs(p0) + fabs(p1);
            const float rel = (float)(fabs((double)d[e] - ex) / (U * mag));
            worst = fmaxf(worst, rel);
            const float chain = __builtin_fmaf(sA[m][1], sB[r][1], __builtin_fmaf(sA[m][0], sB[r][0], c[e]));
            n_chain += (__float_as_uint(chain) == __float_as_uint(d[e])) ? 1u : 0u;
            n_rn += (__float_as_uint((float)ex) == __float_as_uint(d[e])) ? 1u : 0u;
            ++n_all;
        }
        __syncthreads();
    }
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        worst = fmaxf(worst, __shfl_xor(worst, dlt));
        n_chain += __shfl_xor(n_chain, dlt); n_rn += __shfl_xor(n_rn, dlt); n_all += __shfl_xor(n_all, dlt);
    }
    if (lane == 0) {
        atomic_max_pos(out + 0, worst);
        atomicAdd(out + 1, (float)n_chain); atomicAdd(out + 2, (float)n_rn); atomicAdd(out + 5, (float)n_all);
    }
}

// 32 x 32 pairs of 128-element rows per block: the two accumulation forms against binary64 on the same operands
__global__ __launch_bounds__(64) void selftest_f32_chain_kernel(float *__restrict__ out, int iters)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    unsigned long long s = 0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * 64 + lane + 7);
    __shared__ float sX[32][129], sY[32][129];
    __shared__ float nX[32], nY[32];
    float worst2 = 0.f, worst1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int fam = (blockIdx.x + it) % 5;
        const float scale = __uint_as_float((unsigned)(127 - 20 + (int)(rng(s) % 40u)) << 23);      // 2^-20 .. 2^19
        for (int k = h; k < 128; k += 2) {                    // lane (r, h) fills half of row r of both sets
            const float u01 = (float)(rng(s) & 0xFFFFFF) * (1.0f / 16777216.0f), v01 = (float)(rng(s) & 0xFFFFFF) * (1.0f / 16777216.0f);
            float x, y;
            if (fam == 0) { x = u01; y = v01; }                                      // uniform, all-positive
            else if (fam == 1) { x = 0.75f; y = 0.75f + 0x1p-20f * (float)(r & 3); } // constant rows: every product rounds alike
            else if (fam == 2) { x = u01 - 0.5f; y = x * (1.0f + 0x1p-12f * (v01 - 0.5f)); }   // near-duplicates: massive cancellation
            else if (fam == 3) { x = (u01 - 0.5f) * __uint_as_float((unsigned)(127 - 12 + (k % 24)) << 23); y = (v01 - 0.5f) * __uint_as_float((unsigned)(127 - 12 + ((k * 7) % 24)) << 23); }
            else { x = u01 - 0.5f; y = v01 - 0.5f; }                                 // centred
            sX[r][k] = x * scale; sY[r][k] = y * scale;
        }
        __syncthreads();
        if (h == 0) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < 128; ++k) { a = __builtin_fmaf(sX[r][k], sX[r][k], a); b = __builtin_fmaf(sY[r][k], sY[r][k], b); }
            nX[r] = a; nY[r] = b;
        }
        __syncthreads();
        // rows of the result = X (A operand), columns = Y (B operand, scaled by -2), as distance_mfma_kernel issues it
        f32x16s a0, b0, c1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0[e] = 0.f; b0[e] = 0.f; c1[e] = 0.f; }
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 1.0f : nX[r], h ? nY[r] : 1.0f, a0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 1.0f : nX[r], h ? nY[r] : 1.0f, c1, 0, 0, 0);
        for (int k = 0; k < 64; k += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[r][k + h], -2.0f * sY[r][k + h], a0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[r][k + h], -2.0f * sY[r][k + h], c1, 0, 0, 0);
        }
        for (int k = 64; k < 128; k += 2) {
            b0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[r][k + h], -2.0f * sY[r][k + h], b0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[r][k + h], -2.0f * sY[r][k + h], c1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = acc_row(e, h);
            double ex = (double)nX[m] + (double)nY[r], comp = 0.0;
            for (int k = 0; k < 128; ++k) {                   // Kahan in binary64: exact to far below a binary32 ulp
                const double t = -2.0 * (double)sX[m][k] * (double)sY[r][k] - comp, nsum = ex + t;
                comp = (nsum - ex) - t; ex = nsum;
            }
            const double sq = sqrt((double)nX[m]) + sqrt((double)nY[r]);
            const double den = sq * sq;
            if (den > 0.0) {
                worst2 = fmaxf(worst2, (float)(fabs((double)(a0[e] + b0[e]) - ex) / den));
                worst1 = fmaxf(worst1, (float)(fabs((double)c1[e] - ex) / den));
            }
        }
        __syncthreads();
    }
    for (int dlt = 32; dlt >= 1; dlt >>= 1) { worst2 = fmaxf(worst2, __shfl_xor(worst2, dlt)); worst1 = fmaxf(worst1, __shfl_xor(worst1, dlt)); }
    if (lane == 0) { atomic_max_pos(out + 3, worst2); atomic_max_pos(out + 4, worst1); }
}

}  // namespace

extern "C" int nm_selftest_mfma_f32(int n_random, int n_chains, float *d_out, void *stream)
{
    if (!d_out || n_random < 0 || n_chains < 0) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    NM_RETURN_IF(hipMemsetAsync(d_out, 0, 8 * sizeof(float), st));
    const int blocks1 = 1024, it1 = (n_random / 1024 + blocks1 - 1) / blocks1;      // 1024 results per wave-instruction
    if (it1) hipLaunchKernelGGL(selftest_f32_kernel, dim3(blocks1), dim3(64), 0, st, d_out, it1);
    NM_LAUNCH_CHECK();
    const int blocks2 = 512, it2 = (n_chains + blocks2 - 1) / blocks2;
    if (it2) hipLaunchKernelGGL(selftest_f32_chain_kernel, dim3(blocks2), dim3(64), 0, st, d_out, it2);
    NM_LAUNCH_CHECK();
    return 0;
}

extern "C" int nm_selftest_mfma_model(int instruction, int n_random, int n_chains, float *d_out, void *stream)
{
    if (!d_out || (instruction != 0 && instruction != 1) || n_random < 0 || n_chains < 0) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    NM_RETURN_IF(hipMemsetAsync(d_out, 0, NM_SELFTEST_MFMA_OUTPUTS * sizeof(float), st));
    const int blocks1 = 1024, it1 = (n_random + blocks1 - 1) / blocks1;
    const int blocks2 = 512, it2 = (n_chains + blocks2 - 1) / blocks2;
    if (instruction == 1) {
        hipLaunchKernelGGL(selftest_instr_kernel<1>, dim3(blocks1), dim3(64), 0, st, d_out, it1);
        NM_LAUNCH_CHECK();
        if (it2) hipLaunchKernelGGL(selftest_chain_kernel<1>, dim3(blocks2), dim3(64), 0, st, d_out, it2);
    } else {
        hipLaunchKernelGGL(selftest_instr_kernel<0>, dim3(blocks1), dim3(64), 0, st, d_out, it1);
        NM_LAUNCH_CHECK();
        if (it2) hipLaunchKernelGGL(selftest_chain_kernel<0>, dim3(blocks2), dim3(64), 0, st, d_out, it2);
    }
    NM_LAUNCH_CHECK();
    return 0;
}
