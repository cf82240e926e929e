// nm_fpspec.hpp -- device-side floating-point spec of the SIFT kernels (gfx950).
//
// The reference's kernels call CUDA's device libm (atan2f, expf, sinf, cosf, exp, pow), which is defined only up
// to a few ulp. These kernels instead execute ONE fixed sequence of IEEE-754 binary32/binary64 operations per
// function (Cephes-style reductions, explicit fma), so that results are reproducible run to run and identical to a
// host evaluation of the same sequence. Translation units that include this header are compiled with
// -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt; every fused multiply-add is written out.
//
// Call sites in the reference: kernels/cudamath.cu:51-52 (atan2f), kernels/orientation.cu:56 (expf),
// kernels/descriptor.cu:90-91 (sinf, cosf), kernels/descriptor.cu:108 (exp), kernels/keypoint.cu:174 (pow(2,y)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmfp {

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ float pow2i_f(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }
__device__ __forceinline__ double pow2i_d(int n) { return __longlong_as_double((long long)((uint64_t)(n + 1023) << 52)); }

constexpr double TWO_PI_D = 6.283185307179586476925286766559;
constexpr double INV_TWO_PI_D = 1.0 / 6.283185307179586476925286766559;    // RN64(1 / TWO_PI_D)

// (float)(a / b) without the fp64 divide, bit-identical to the plain expression. r must be RN64(1/b).
// y = a*r lies within 2.5 double ulps of RN64(a/b); the two narrow to the same float unless a float rounding boundary
// (low 29 mantissa bits = 0x10000000) sits within a few double ulps of y -- then (p ~ 2^-24), and for results too small
// to be normal floats, the exact quotient is computed.
__device__ __forceinline__ float div_to_f32(double a, double b, double r)
{
    const double y = a * r;
    const unsigned long long u = (unsigned long long)__double_as_longlong(y);
    const uint32_t ulo = (uint32_t)u, uhi = (uint32_t)(u >> 32);
    // Two instructions per test (v_lshl_add_u32 + an unsigned compare): the shifts drop the bits that do not take part.
    //   near: the low 29 mantissa bits lie in [0x0FFFFFF0, 0x10000010] (8 x the difference, modulo 2^32, is <= 0x100 iff
    //         the difference is in [0, 0x20]);
    //   tiny: 0 < |y| < 2^-120 (biased exponent below 903), read off the sign-less high word. A double denormal whose
    //         high mantissa bits are zero is not caught: its float -- and that of the exact quotient -- is a zero of the
    //         same sign either way.
    const bool near = ((ulo << 3) - (0x0FFFFFF0u << 3)) <= (0x20u << 3);
    const bool tiny = ((uhi << 1) - 2u) < (((1023u - 120u) << 21) - 2u);
    if (__builtin_expect(near | tiny, 0)) return (float)(a / b);
    return (float)y;
}
// The same for a caller whose lanes all take the division together (a WAVE-UNIFORM branch: no exec-masked block in the
// caller's instruction stream). Lanes that are not near a boundary get the same float from either expression.
__device__ __forceinline__ float div_to_f32_wave(double a, double b, double r)
{
    const double y = a * r;
    const unsigned long long u = (unsigned long long)__double_as_longlong(y);
    const uint32_t ulo = (uint32_t)u, uhi = (uint32_t)(u >> 32);
    const bool near = ((ulo << 3) - (0x0FFFFFF0u << 3)) <= (0x20u << 3);
    const bool tiny = ((uhi << 1) - 2u) < (((1023u - 120u) << 21) - 2u);
    if (__builtin_expect(__any(near | tiny), 0)) return (float)(a / b);
    return (float)y;
}
// Three quotients of one wave at once (the descriptor's nx, ny, nt): (float)(a1 / b), (float)(a2 / b), (float)(a3 / b3) with
// r = RN64(1 / b), r3 = RN64(1 / b3). Same `near` test per value as div_to_f32; the three `tiny` tests (0 < |y| < 2^-120, three
// instructions each on the binary64 bits) become ONE on the narrowed results: |y| < 2^-120 implies |(float)y| <= 2^-120
// (rounding is monotonic), so  min(|f1|, |f2|, |f3|) <= 2^-120  is a superset of "some y is tiny" -- it also takes exact zeros and
// the first 2^-25 of the binade above, for which the exact quotients it then computes are equally right. One branch for all three.
__device__ __forceinline__ void div3_to_f32_wave(double a1, double a2, double b, double r, double a3, double b3, double r3,
                                                 float &f1, float &f2, float &f3)
{
    const double y1 = a1 * r, y2 = a2 * r, y3 = a3 * r3;
    const uint32_t l1 = (uint32_t)(unsigned long long)__double_as_longlong(y1);
    const uint32_t l2 = (uint32_t)(unsigned long long)__double_as_longlong(y2);
    const uint32_t l3 = (uint32_t)(unsigned long long)__double_as_longlong(y3);
    const bool near = (((l1 << 3) - (0x0FFFFFF0u << 3)) <= (0x20u << 3)) | (((l2 << 3) - (0x0FFFFFF0u << 3)) <= (0x20u << 3)) |
                      (((l3 << 3) - (0x0FFFFFF0u << 3)) <= (0x20u << 3));
    f1 = (float)y1; f2 = (float)y2; f3 = (float)y3;
    const bool tiny = __builtin_fminf(__builtin_fminf(__builtin_fabsf(f1), __builtin_fabsf(f2)), __builtin_fabsf(f3)) <= 0x1p-120f;
    if (__builtin_expect(__any(near | tiny), 0)) { f1 = (float)(a1 / b); f2 = (float)(a2 / b); f3 = (float)(a3 / b3); }
}
constexpr float TWO_PI_F = (float)(2 * 3.14159265358979323846);

// (float)((double)x / 3.0) in three binary32 instructions: q = RN(x R) with R = RN32(1/3), one exact-residual correction. Equal to
// the binary64 expression for EVERY binary32 x except -0 (gives +0), the infinities and NaN (give NaN), which third_f32_exact
// names (one v_cmp_class) so that a caller can send them through div_to_f32: checked over all 2^32 inputs by nm_selftest_orient.
__device__ __forceinline__ bool third_f32_exact(float x) { return !__builtin_amdgcn_classf(x, 0x227); }   // not: NaN, +-inf, -0
__device__ __forceinline__ float third_f32(float x)
{
    const float R = 0x1.555556p-2f;
    const float q = x * R;
    return fma32(fma32(-3.0f, q, x), R, q);
}

// num / den for one divisor and many numerators: the instruction sequence of the IEEE expansion (rcp, one refinement, two
// quotient corrections) with the divisor's part done once, and without the expansion's range scaling and special-case fix-up.
// The same binary32 as `num / den` whenever neither is needed: den in [2^-20, 2^20] (div_by_domain), num = 0 or in [2^-100, 2^7)
// -- the quotient and both exact residuals then stay in the normal range. Checked against `/` on that domain by nm_selftest_orient.
struct DivBy {
    float nden, rc;
};
__device__ __forceinline__ bool div_by_domain(float den) { return den >= 0x1p-20f && den <= 0x1p20f; }
__device__ __forceinline__ DivBy div_by(float den)
{
    DivBy d;
    const float r0 = __builtin_amdgcn_rcpf(den);
    d.nden = -den;
    d.rc = fma32(fma32(d.nden, r0, 1.0f), r0, r0);
    return d;
}
__device__ __forceinline__ float div_by(const DivBy &d, float num)
{
    float q = num * d.rc;
    q = fma32(fma32(d.nden, q, num), d.rc, q);
    return fma32(fma32(d.nden, q, num), d.rc, q);
}

// atan(ay/ax), ax > 0, ay >= 0: one IEEE division (range chosen by products, see oracle/nmo_math.h)
__device__ __forceinline__ float atanf_q1(float ay, float ax)
{
    float hi, lo, num, den;
    if (ay > 2.414213562373095f * ax)       { hi = 1.57079637050628662109375f;  lo = -4.37113900018624283e-8f; num = -ax; den = ay; }
    else if (ay > 0.4142135623730950f * ax) { hi = 0.785398185253143310546875f; lo = -2.18556950009312142e-8f; num = ay - ax; den = ay + ax; }
    else                                    { hi = 0.0f; lo = 0.0f; num = ay; den = ax; }
    const float z = num / den;
    const float zz = z * z;
    float p = fma32(-0.06459416449069977f, zz, 0.10746313631534576f);
    p = fma32(p, zz, -0.14264234900474548f);
    p = fma32(p, zz, 0.1999955028295517f);
    p = fma32(p, zz, -0.3333333134651184f);
    p = p * zz;
    p = fma32(p, z, z);
    return hi + (p + lo);
}

__device__ __forceinline__ float atan2f_spec(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float r;
    if (ax == 0.0f) r = (ay == 0.0f) ? 0.0f : 1.57079637050628662109375f;
    else            r = atanf_q1(ay, ax);
    if (x < 0.0f) r = (3.1415927410125732421875f - r) + -8.74227800037248566e-8f;
    if (y < 0.0f) r = -r;
    return r;
}

__device__ __forceinline__ float mod_2pi_f(float x)      // kernels/cudamath.h:82-87
{
    while (x > TWO_PI_F) x -= TWO_PI_F;
    while (x < 0.0f) x += TWO_PI_F;
    return x;
}

// CLAMP: which of the two clamps are compiled (1: upper, 2: lower); a caller that knows x <= 88 / x >= -87 (or NaN, for which
// neither fires) drops them.
template <int CLAMP = 3>
__device__ __forceinline__ float expf_spec(float x)
{
    if ((CLAMP & 1) && x > 88.0f) x = 88.0f;
    if ((CLAMP & 2) && x < -87.0f) x = -87.0f;
    const float z = __builtin_floorf(fma32(1.44269504088896341f, x, 0.5f));
    const int n = (int)z;
    float r = fma32(z, -0.693359375f, x);
    r = fma32(z, 2.12194440e-4f, r);
    const float rr = r * r;
    float p = fma32(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fma32(p, r, 8.3334519073e-3f);
    p = fma32(p, r, 4.1665795894e-2f);
    p = fma32(p, r, 1.6666665459e-1f);
    p = fma32(p, r, 5.0000001201e-1f);
    p = fma32(p, rr, r);
    p = p + 1.0f;
    return p * pow2i_f(n);
}

__device__ __forceinline__ void sincos_reduce(float ax, int &j, float &r)
{
    int jj = (int)(1.27323954473516f * ax);
    float y = (float)jj;
    if (jj & 1) { jj += 1; y += 1.0f; }
    r = fma32(y, -0.78515625f, ax);
    r = fma32(y, -2.4187564849853515625e-4f, r);
    r = fma32(y, -3.77489497744594108e-8f, r);
    j = jj & 7;
}
__device__ __forceinline__ float sinpoly(float r)
{
    const float z = r * r;
    float p = fma32(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = fma32(p, z, -1.6666654611e-1f);
    p = p * z;
    return fma32(p, r, r);
}
__device__ __forceinline__ float cospoly(float r)
{
    const float z = r * r;
    float p = fma32(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fma32(p, z, 4.166664568298827e-2f);
    p = p * z;
    p = p * z;
    p = fma32(-0.5f, z, p);
    return p + 1.0f;
}
__device__ __forceinline__ float sinf_spec(float x)
{
    bool neg = x < 0.0f;
    int j; float r;
    sincos_reduce(__builtin_fabsf(x), j, r);
    if (j > 3) { neg = !neg; j -= 4; }
    const float y = (j == 1 || j == 2) ? cospoly(r) : sinpoly(r);
    return neg ? -y : y;
}
__device__ __forceinline__ float cosf_spec(float x)
{
    bool neg = false;
    int j; float r;
    sincos_reduce(__builtin_fabsf(x), j, r);
    if (j > 3) { neg = !neg; j -= 4; }
    if (j > 1) neg = !neg;
    const float y = (j == 1 || j == 2) ? sinpoly(r) : cospoly(r);
    return neg ? -y : y;
}

// exp_spec for an argument the caller has already brought into [-700, 700]: the same operation sequence without the two
// clamps (which would be no-ops).
__device__ __forceinline__ double exp_spec_in_range(double x)
{
    const double px = __builtin_floor(fma64(1.4426950408889634073599, x, 0.5));
    const int n = (int)px;
    x = fma64(px, -6.93145751953125e-1, x);
    x = fma64(px, -1.42860682030941723212e-6, x);
    const double xx = x * x;
    double p = fma64(1.26177193074810590878e-4, xx, 3.02994407707441961300e-2);
    p = fma64(p, xx, 9.99999999999999999910e-1);
    p = p * x;
    double q = fma64(3.00198505138664455042e-6, xx, 2.52448340349684104192e-3);
    q = fma64(q, xx, 2.27265548208155028766e-1);
    q = fma64(q, xx, 2.00000000000000000009e0);
    double r = p / (q - p);
    r = fma64(2.0, r, 1.0);
    return r * pow2i_d(n);
}

// ---- (float)exp_spec(t / 8) for a float 0 <= t <= 12.875: the descriptor's window weight on every sample that can vote ----
// exp_spec_in_range is ~34 binary64 instructions with a division; its result is only used narrowed to binary32. The fast form
//   t / 8 = n ln 2 + r,  n = rint(t / 8 log2 e) in {0, 1, 2},  |r| <= 0.3466,   y = 2^n (1 + r + r^2/2! + ... + r^10/10!)
// (the reduction with exp_spec's own two-piece ln 2) is within 2^-41.5 (relative) of exp(t / 8): the Taylor remainder
// r^11 / 11! e^|r| < 2^-41.6 of the value, the roundings of the reduction, the coefficients and the 11 operations < 2^-49
// together. exp_spec is within a few 2^-53 of it. Two binary64 values that close narrow to the SAME binary32 unless a binary32
// rounding boundary (low 29 mantissa bits = 0x10000000) lies between them, which needs the low 29 bits of y within 2^11.5 of
// that pattern: `near` reports a window of +-2^13 (probability 2^-15 per sample), and the caller then evaluates exp_spec
// itself. No table, no memory access. Checked against exp_spec for EVERY float of the range by nm_selftest_expw.
constexpr float EXPW_TMAX = 12.875f;
__device__ __forceinline__ float expw_fast(float t, bool &near)
{
    const double x = (double)t * 0.125;
    const double nd = __builtin_rint(x * 1.4426950408889634073599);
    double r = fma64(nd, -6.93145751953125e-1, x);
    r = fma64(nd, -1.42860682030941723212e-6, r);
    // Horner steps p = fma(p, r, c) with the coefficient as a SCALAR operand (the compiler keeps such constants in vector
    // registers and copies one into the accumulator ahead of every v_fmac_f64: ten extra moves per sample)
    auto step = [](double pp, double rr, double c) {
        double o;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(pp), "v"(rr), "s"(c));
        return o;
    };
    double p = step(1.0 / 3628800.0, r, 1.0 / 362880.0);
    p = step(p, r, 1.0 / 40320.0);
    p = step(p, r, 1.0 / 5040.0);
    p = step(p, r, 1.0 / 720.0);
    p = step(p, r, 1.0 / 120.0);
    p = step(p, r, 1.0 / 24.0);
    p = step(p, r, 1.0 / 6.0);
    p = fma64(p, r, 0.5);
    p = fma64(p, r, 1.0);
    p = fma64(p, r, 1.0);
    const double y = __builtin_ldexp(p, (int)nd);
    const uint32_t ulo = (uint32_t)(unsigned long long)__double_as_longlong(y);
    near = ((ulo << 3) - ((0x10000000u - 8192u) << 3)) <= (16384u << 3);
    return (float)y;
}

__device__ __forceinline__ double exp_spec(double x)
{
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    const double px = __builtin_floor(fma64(1.4426950408889634073599, x, 0.5));
    const int n = (int)px;
    x = fma64(px, -6.93145751953125e-1, x);
    x = fma64(px, -1.42860682030941723212e-6, x);
    const double xx = x * x;
    double p = fma64(1.26177193074810590878e-4, xx, 3.02994407707441961300e-2);
    p = fma64(p, xx, 9.99999999999999999910e-1);
    p = p * x;
    double q = fma64(3.00198505138664455042e-6, xx, 2.52448340349684104192e-3);
    q = fma64(q, xx, 2.27265548208155028766e-1);
    q = fma64(q, xx, 2.00000000000000000009e0);
    double r = p / (q - p);
    r = fma64(2.0, r, 1.0);
    return r * pow2i_d(n);
}

__device__ __forceinline__ double exp2_spec(double y)
{
    const double n = __builtin_floor(y + 0.5);
    const double f = y - n;
    const double r = exp_spec(f * 0.693147180559945309417);
    return r * pow2i_d((int)n);
}

}  // namespace nmfp
