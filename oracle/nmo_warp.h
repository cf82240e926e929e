/*
 * oracle/nmo_warp.h -- TEST INFRASTRUCTURE ONLY. CPU restatement of the texture-sampled warps of the reference:
 * kernels/undistort.cu:7-46, kernels/resample.cu:7-233, with the texture unit of utils/cudatex2D.cu:13-19 (border
 * addressing, linear filter, unnormalised coordinates, normalised-float reads of 8-bit texels) written out in software.
 *
 * PARITY UNPINNED, and more so than elsewhere: CUDA documents the linear filter only as
 *     tex(x,y) = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1],
 *     i = floor(x-0.5), a = frac(x-0.5) "stored in 9-bit fixed point with 8 bits of fractional value",
 * not the rounding of a, nor the order of the sum. THIS spec: a is rounded to the nearest 1/256 (half up), the four
 * weights are the exact products, the sum runs left to right in binary32 without contraction, 8-bit texels are c/255
 * (IEEE division), coordinates that cannot touch a texel (including NaN/inf) return 0.
 */
#ifndef NMO_WARP_H
#define NMO_WARP_H
#include <cmath>
#include <cstddef>

enum { NMO_TEX_U8N = 0, NMO_TEX_U8X4N = 1, NMO_TEX_F32 = 2 };

struct nmo_tex { const void *data; int w, h, fmt; };

static inline float nmo_texel(const nmo_tex &t, int i, int j, int ch)
{
    if (i < 0 || i >= t.w || j < 0 || j >= t.h) return 0.f;                       /* cudaAddressModeBorder */
    const size_t p = (size_t)j * t.w + i;
    if (t.fmt == NMO_TEX_F32) return ((const float *)t.data)[p];
    if (t.fmt == NMO_TEX_U8N) return (float)((const unsigned char *)t.data)[p] / 255.0f;
    return (float)((const unsigned char *)t.data)[4 * p + ch] / 255.0f;
}

static inline bool nmo_tex_setup(const nmo_tex &t, float x, float y, int &i, int &j, float w[4])
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    if (!(xb >= -1.0f && xb < (float)t.w && yb >= -1.0f && yb < (float)t.h)) return false;
    const float fi = std::floor(xb), fj = std::floor(yb);
    const float a = std::floor((xb - fi) * 256.0f + 0.5f) * 0.00390625f;
    const float b = std::floor((yb - fj) * 256.0f + 0.5f) * 0.00390625f;
    i = (int)fi; j = (int)fj;
    w[0] = (1.0f - a) * (1.0f - b); w[1] = a * (1.0f - b); w[2] = (1.0f - a) * b; w[3] = a * b;
    return true;
}

static inline float nmo_tex2d(const nmo_tex &t, float x, float y, int ch)
{
    int i, j; float w[4];
    if (!nmo_tex_setup(t, x, y, i, j, w)) return 0.f;
    return ((w[0] * nmo_texel(t, i, j, ch) + w[1] * nmo_texel(t, i + 1, j, ch)) + w[2] * nmo_texel(t, i, j + 1, ch)) +
           w[3] * nmo_texel(t, i + 1, j + 1, ch);
}

/* x' = (m0 x + m1 y + m2) / (m6 x + m7 y + m8) -- resample.cu:29-33,161-166,192-197; contraction as nvcc's:
 * fma(m0, x, m1*y) + m2 */
static inline void nmo_project(const float m[9], float x, float y, float &xp, float &yp)
{
    const float a = std::fmaf(m[0], x, m[1] * y) + m[2];
    const float b = std::fmaf(m[3], x, m[4] * y) + m[5];
    const float s = std::fmaf(m[6], x, m[7] * y) + m[8];
    xp = a / s; yp = b / s;
}

/* adjugate / determinant of resample.cu:131-147 (thread 0 of each block; float). a*b - c*d := fma(a, b, -(c*d)). */
static inline void nmo_invert3x3(const float t[9], float inv[9])
{
    const float c0 = std::fmaf(t[4], t[8], -(t[7] * t[5]));
    const float c1 = std::fmaf(t[3], t[8], -(t[5] * t[6]));
    const float c2 = std::fmaf(t[3], t[7], -(t[4] * t[6]));
    const float det = std::fmaf(t[2], c2, std::fmaf(t[0], c0, -(t[1] * c1)));
    const float invdet = 1.0f / det;
    inv[0] = c0 * invdet;
    inv[1] = std::fmaf(t[2], t[7], -(t[1] * t[8])) * invdet;
    inv[2] = std::fmaf(t[1], t[5], -(t[2] * t[4])) * invdet;
    inv[3] = std::fmaf(t[5], t[6], -(t[3] * t[8])) * invdet;
    inv[4] = std::fmaf(t[0], t[8], -(t[2] * t[6])) * invdet;
    inv[5] = std::fmaf(t[3], t[2], -(t[0] * t[5])) * invdet;
    inv[6] = std::fmaf(t[3], t[7], -(t[6] * t[4])) * invdet;
    inv[7] = std::fmaf(t[6], t[1], -(t[0] * t[7])) * invdet;
    inv[8] = std::fmaf(t[0], t[4], -(t[3] * t[1])) * invdet;
}

/* undistort.cu:27-44 for one position; powf(x,2) := x*x, powf(x,3) := (x*x)*x */
static inline void nmo_undistort_point(float x, float y, float fx, float fy, float cx, float cy, float k1, float k2,
                                       float k3, float &u, float &v)
{
    u = (x - cx) / fx;
    v = (y - cy) / fy;
    const float r2 = std::fmaf(u, u, v * v);
    const float r4 = r2 * r2, r6 = r4 * r2;
    const float poly = std::fmaf(k3, r6, std::fmaf(k2, r4, std::fmaf(k1, r2, 1.0f)));
    u = std::fmaf(u * poly, fx, cx);                 /* `u *= fx; u += cx;` contracts */
    v = std::fmaf(v * poly, fy, cy);
}
#endif
