/*
 * oracle/nmo_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Not linked into the product.
 *
 * Deterministic restatement of the device libm calls the reference's CUDA kernels make:
 *   atan2f   kernels/cudamath.cu:51-52     (gradient angle)
 *   expf     kernels/orientation.cu:56     (orientation window weight; `exp(float)` resolves to the float overload)
 *   sinf/cosf kernels/descriptor.cu:90-91  (`sin(float)`/`cos(float)` -> float overloads, widened to double)
 *   exp      kernels/descriptor.cu:108     (double)
 *   pow(2.0, y) kernels/keypoint.cu:174    (double)
 *
 * The reference's values come from CUDA 7's device libm, which is not pinned, not vendored and documented only
 * by ulp bounds (atan2f 2 ulp, expf 2 ulp, sinf/cosf 2 ulp, exp/pow 1-2 ulp). Any implementation inside those
 * bounds is an equally valid restatement. We fix ONE: the classic Cephes single/double kernels written as an
 * explicit sequence of IEEE-754 operations (+,-,*,/,fma, floor, int conversion), so that the HIP kernels can
 * execute the identical sequence and agree with this oracle bit for bit. tests/test_oracle_math.py pins every
 * function here against glibc within a stated ulp bound.
 *
 * Build rule: compile with -ffp-contract=off; every fused multiply-add is written explicitly.
 */
#ifndef NMO_MATH_H
#define NMO_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

static inline float nmo_pow2i_f(int n)            /* exact 2^n, n in [-126,127] */
{
    uint32_t bits = (uint32_t)(n + 127) << 23;
    float f; std::memcpy(&f, &bits, 4); return f;
}
static inline double nmo_pow2i_d(int n)           /* exact 2^n, n in [-1022,1023] */
{
    uint64_t bits = (uint64_t)(n + 1023) << 52;
    double d; std::memcpy(&d, &bits, 8); return d;
}

/* atan(ay/ax) for ax > 0, ay >= 0: Cephes atanf range reduction with ONE division. The range is chosen by products
 * (ay > tan(3pi/8) ax, ay > tan(pi/8) ax), the reduced argument z = num/den is formed from ax, ay directly, and the
 * pi/2, pi/4 offsets are added as hi+lo pairs to stay near 2 ulp. */
static inline float nmo_atanf_q1(float ay, float ax)
{
    float hi, lo, num, den;
    if (ay > 2.414213562373095f * ax)      { hi = 1.57079637050628662109375f; lo = -4.37113900018624283e-8f; num = -ax; den = ay; }
    else if (ay > 0.4142135623730950f * ax){ hi = 0.785398185253143310546875f; lo = -2.18556950009312142e-8f; num = ay - ax; den = ay + ax; }
    else                                   { hi = 0.0f; lo = 0.0f; num = ay; den = ax; }
    const float z = num / den;
    const float zz = z * z;
    float p = std::fmaf(-0.06459416449069977f, zz, 0.10746313631534576f);      /* degree-4 minimax in z^2 on */
    p = std::fmaf(p, zz, -0.14264234900474548f);                              /* [0, tan^2(pi/8)], fitted for */
    p = std::fmaf(p, zz, 0.1999955028295517f);                                /* this project: 3e-9 relative  */
    p = std::fmaf(p, zz, -0.3333333134651184f);
    p = p * zz;
    p = std::fmaf(p, z, z);
    return hi + (p + lo);
}

/* atan2f(y, x) in [-pi, pi]. */
static inline float nmo_atan2f(float y, float x)
{
    const float ax = std::fabs(x), ay = std::fabs(y);
    float r;
    if (ax == 0.0f) r = (ay == 0.0f) ? 0.0f : 1.57079637050628662109375f;
    else            r = nmo_atanf_q1(ay, ax);
    if (x < 0.0f) r = (3.1415927410125732421875f - r) + -8.74227800037248566e-8f;
    if (y < 0.0f) r = -r;
    return r;
}

/* expf(x), x clamped to [-87, 88] (Cephes expf). */
static inline float nmo_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float z = std::floor(std::fmaf(1.44269504088896341f, x, 0.5f));
    const int n = (int)z;
    float r = std::fmaf(z, -0.693359375f, x);
    r = std::fmaf(z, 2.12194440e-4f, r);
    const float rr = r * r;
    float p = std::fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = std::fmaf(p, r, 8.3334519073e-3f);
    p = std::fmaf(p, r, 4.1665795894e-2f);
    p = std::fmaf(p, r, 1.6666665459e-1f);
    p = std::fmaf(p, r, 5.0000001201e-1f);
    p = std::fmaf(p, rr, r);
    p = p + 1.0f;
    return p * nmo_pow2i_f(n);
}

/* Shared octant reduction for sinf/cosf (Cephes; |x| < 8192). */
static inline void nmo_sincos_reduce(float ax, int *j_out, float *r_out)
{
    int j = (int)(1.27323954473516f * ax);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    float r = std::fmaf(y, -0.78515625f, ax);
    r = std::fmaf(y, -2.4187564849853515625e-4f, r);
    r = std::fmaf(y, -3.77489497744594108e-8f, r);
    *j_out = j & 7; *r_out = r;
}
static inline float nmo_sinpoly(float r)
{
    const float z = r * r;
    float p = std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = std::fmaf(p, z, -1.6666654611e-1f);
    p = p * z;
    return std::fmaf(p, r, r);
}
static inline float nmo_cospoly(float r)
{
    const float z = r * r;
    float p = std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = std::fmaf(p, z, 4.166664568298827e-2f);
    p = p * z;
    p = p * z;
    p = std::fmaf(-0.5f, z, p);
    return p + 1.0f;
}
static inline float nmo_sinf(float x)
{
    bool neg = x < 0.0f;
    int j; float r;
    nmo_sincos_reduce(std::fabs(x), &j, &r);
    if (j > 3) { neg = !neg; j -= 4; }
    const float y = (j == 1 || j == 2) ? nmo_cospoly(r) : nmo_sinpoly(r);
    return neg ? -y : y;
}
static inline float nmo_cosf(float x)
{
    bool neg = false;
    int j; float r;
    nmo_sincos_reduce(std::fabs(x), &j, &r);
    if (j > 3) { neg = !neg; j -= 4; }
    if (j > 1) neg = !neg;
    const float y = (j == 1 || j == 2) ? nmo_sinpoly(r) : nmo_cospoly(r);
    return neg ? -y : y;
}

/* exp(x) double, x clamped to [-700, 700] (Cephes exp: Pade form). */
static inline double nmo_exp(double x)
{
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    const double px = std::floor(std::fma(1.4426950408889634073599, x, 0.5));
    const int n = (int)px;
    x = std::fma(px, -6.93145751953125e-1, x);
    x = std::fma(px, -1.42860682030941723212e-6, x);
    const double xx = x * x;
    double p = std::fma(1.26177193074810590878e-4, xx, 3.02994407707441961300e-2);
    p = std::fma(p, xx, 9.99999999999999999910e-1);
    p = p * x;
    double q = std::fma(3.00198505138664455042e-6, xx, 2.52448340349684104192e-3);
    q = std::fma(q, xx, 2.27265548208155028766e-1);
    q = std::fma(q, xx, 2.00000000000000000009e0);
    double r = p / (q - p);
    r = std::fma(2.0, r, 1.0);
    return r * nmo_pow2i_d(n);
}

/* 2^y double (the reference's pow(2.0, y)), |y| < 1000. */
static inline double nmo_exp2(double y)
{
    const double n = std::floor(y + 0.5);
    const double f = y - n;
    const double r = nmo_exp(f * 0.693147180559945309417);
    return r * nmo_pow2i_d((int)n);
}

#endif
