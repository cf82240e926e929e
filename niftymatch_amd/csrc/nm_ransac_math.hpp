// nm_ransac_math.hpp -- per-hypothesis model fits and the inlier test of the RANSAC stage for gfx950.
// Replaces compute_translation / compute_similarity_transform / compute_homography_2 / eval_transformation of
// kernels/ransac.cu:61-427. The null vector of the design matrix comes from an independent Hestenes one-sided Jacobi
// (the reference's kernels/svd.cu is a GPL port of GSL and is not reproduced). Operation sequence identical to the
// CPU oracle's (explicit fma, IEEE divide/sqrt, -ffp-contract=off), so both agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#define NMR_SQ_SUM(a, b) __builtin_fmaf((a), (a), (b) * (b))

/* One-sided Jacobi on A (M x N, row-major, leading dimension N), accumulating V (N x N). Returns in `nullv` (N values)
 * the column of V that belongs to the smallest singular value. */
__device__ __forceinline__ void nmr_jacobi_null_vector(float *A, int M, int N, float *V, float *nullv)
{
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0f : 0.0f;
    const float tol = 10.0f * (float)M * 1.1920928955078125e-07f;
    const int max_sweeps = (5 * N > 12) ? 5 * N : 12;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        int rotations = 0;
        for (int j = 0; j < N - 1; ++j)
            for (int k = j + 1; k < N; ++k) {
                float a = 0.f, b = 0.f, p = 0.f;
                for (int r = 0; r < M; ++r) {
                    const float x = A[r * N + j], y = A[r * N + k];
                    a = __builtin_fmaf(x, x, a); b = __builtin_fmaf(y, y, b); p = __builtin_fmaf(x, y, p);
                }
                if (a == 0.f || b == 0.f) continue;
                if (__builtin_fabsf(p) <= tol * __builtin_sqrtf(a * b)) continue;
                ++rotations;
                const float zeta = (b - a) / (2.0f * p);
                const float t = ((zeta >= 0.f) ? 1.0f : -1.0f) / (__builtin_fabsf(zeta) + __builtin_sqrtf(__builtin_fmaf(zeta, zeta, 1.0f)));
                const float c = 1.0f / __builtin_sqrtf(__builtin_fmaf(t, t, 1.0f));
                const float s = c * t;
                for (int r = 0; r < M; ++r) {
                    const float x = A[r * N + j], y = A[r * N + k];
                    A[r * N + j] = __builtin_fmaf(c, x, -(s * y));
                    A[r * N + k] = __builtin_fmaf(s, x, c * y);
                }
                for (int r = 0; r < N; ++r) {
                    const float x = V[r * N + j], y = V[r * N + k];
                    V[r * N + j] = __builtin_fmaf(c, x, -(s * y));
                    V[r * N + k] = __builtin_fmaf(s, x, c * y);
                }
            }
        if (rotations == 0) break;
    }
    int best = 0;
    float best_norm = 0.f;
    for (int j = 0; j < N; ++j) {
        float a = 0.f;
        for (int r = 0; r < M; ++r) a = __builtin_fmaf(A[r * N + j], A[r * N + j], a);
        if (j == 0 || a < best_norm) { best_norm = a; best = j; }
    }
    for (int r = 0; r < N; ++r) nullv[r] = V[r * N + best];
}

/* inv(dst_transform) * H * src_transform, expanded as in ransac.cu:196-214 (same grouping, contractions explicit) */
__device__ __forceinline__ void nmr_denormalise(const float H[9], float s1, float s2, float tx1, float ty1, float tx2, float ty2,
                                   float out[9])
{
    const float w = __builtin_fmaf(-(s1 * tx1), H[6], __builtin_fmaf(-(s1 * ty1), H[7], H[8]));
    out[0] = __builtin_fmaf(s1 * tx2, H[6], (s1 * H[0]) / s2);
    out[1] = __builtin_fmaf(s1 * tx2, H[7], (s1 * H[1]) / s2);
    out[2] = __builtin_fmaf(tx2, w, __builtin_fmaf(-(s1 * tx1), H[0], __builtin_fmaf(-(s1 * ty1), H[1], H[2])) / s2);
    out[3] = __builtin_fmaf(s1 * ty2, H[6], (s1 * H[3]) / s2);
    out[4] = __builtin_fmaf(s1 * ty2, H[7], (s1 * H[4]) / s2);
    out[5] = __builtin_fmaf(ty2, w, __builtin_fmaf(-(s1 * tx1), H[3], __builtin_fmaf(-(s1 * ty1), H[4], H[5])) / s2);
    out[6] = s1 * H[6];
    out[7] = s1 * H[7];
    out[8] = w;
}

__device__ __forceinline__ void nmr_fit_translation(const float sx[1], const float sy[1], const float dx[1], const float dy[1], float H[9])
{
    H[0] = H[4] = H[8] = 1.f;
    H[1] = H[3] = H[6] = H[7] = 0.f;
    H[2] = dx[0] - sx[0];
    H[5] = dy[0] - sy[0];
}

__device__ __forceinline__ void nmr_fit_similarity(const float sx[2], const float sy[2], const float dx[2], const float dy[2], float H[9])
{
    const float smx = (sx[0] + sx[1]) * 0.5f, smy = (sy[0] + sy[1]) * 0.5f;
    const float dmx = (dx[0] + dx[1]) * 0.5f, dmy = (dy[0] + dy[1]) * 0.5f;
    float sv = 0.f, dv = 0.f;
    for (int i = 0; i < 2; ++i) {
        sv += NMR_SQ_SUM(sx[i] - smx, sy[i] - smy);
        dv += NMR_SQ_SUM(dx[i] - dmx, dy[i] - dmy);
    }
    sv = (float)((double)sv * 0.5); dv = (float)((double)dv * 0.5);            /* `*= 0.5` with a double literal */
    const float r2 = __builtin_sqrtf(2.0f);
    const float s1 = r2 / __builtin_sqrtf(sv), s2 = r2 / __builtin_sqrtf(dv);
    float X[4 * 5], V[5 * 5], nv[5];
    for (int i = 0; i < 2; ++i) {
        const float ax = (sx[i] - smx) * s1, ay = (sy[i] - smy) * s1;
        const float bx = (dx[i] - dmx) * s2, by = (dy[i] - dmy) * s2;
        float *r0 = X + (2 * i) * 5, *r1 = X + (2 * i + 1) * 5;
        r0[0] = ax; r0[1] = 1.f; r0[2] = -ay; r0[3] = 0.f; r0[4] = bx;
        r1[0] = ay; r1[1] = 0.f; r1[2] = ax;  r1[3] = 1.f; r1[4] = by;
    }
    nmr_jacobi_null_vector(X, 4, 5, V, nv);
    const float a0 = -nv[0] / nv[4], a1 = -nv[1] / nv[4], b0 = -nv[2] / nv[4], b1 = -nv[3] / nv[4];
    const float Hn[9] = {a0, -b0, a1, b0, a0, b1, 0.f, 0.f, 1.f};
    nmr_denormalise(Hn, s1, s2, smx, smy, dmx, dmy, H);
}

__device__ __forceinline__ void nmr_fit_homography(const float sx[4], const float sy[4], const float dx[4], const float dy[4], float H[9])
{
    const float smx = (((sx[0] + sx[1]) + sx[2]) + sx[3]) * 0.25f, smy = (((sy[0] + sy[1]) + sy[2]) + sy[3]) * 0.25f;
    const float dmx = (((dx[0] + dx[1]) + dx[2]) + dx[3]) * 0.25f, dmy = (((dy[0] + dy[1]) + dy[2]) + dy[3]) * 0.25f;
    float sv = 0.f, dv = 0.f;
    for (int i = 0; i < 4; ++i) {
        sv += NMR_SQ_SUM(sx[i] - smx, sy[i] - smy);
        dv += NMR_SQ_SUM(dx[i] - dmx, dy[i] - dmy);
    }
    sv *= 0.25f; dv *= 0.25f;
    const float r2 = __builtin_sqrtf(2.0f);
    const float s1 = r2 / __builtin_sqrtf(sv), s2 = r2 / __builtin_sqrtf(dv);
    float X[9 * 9], V[9 * 9], nv[9];
    float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    for (int i = 0; i < 4; ++i) {
        ax = (sx[i] - smx) * s1; ay = (sy[i] - smy) * s1;
        bx = (dx[i] - dmx) * s2; by = (dy[i] - dmy) * s2;
        float *r0 = X + (2 * i) * 9, *r1 = X + (2 * i + 1) * 9;
        r0[0] = 0.f; r0[1] = 0.f; r0[2] = 0.f; r0[3] = -ax; r0[4] = -ay; r0[5] = -1.f; r0[6] = by * ax; r0[7] = by * ay; r0[8] = by;
        r1[0] = ax;  r1[1] = ay;  r1[2] = 1.f; r1[3] = 0.f; r1[4] = 0.f; r1[5] = 0.f;  r1[6] = -bx * ax; r1[7] = -bx * ay; r1[8] = -bx;
    }
    float *r8 = X + 8 * 9;                                      /* third equation of the last point (ransac.cu:161-178) */
    r8[0] = -by * ax; r8[1] = -by * ay; r8[2] = -by; r8[3] = bx * ax; r8[4] = bx * ay; r8[5] = bx; r8[6] = 0.f; r8[7] = 0.f; r8[8] = 0.f;
    nmr_jacobi_null_vector(X, 9, 9, V, nv);
    float Hn[9];
    for (int i = 0; i < 8; ++i) Hn[i] = nv[i] / nv[8];
    Hn[8] = 1.f;
    nmr_denormalise(Hn, s1, s2, smx, smy, dmx, dmy, H);
}

/* eval_transformation's per-point test -- ransac.cu:68-78 */
__device__ __forceinline__ bool nmr_is_inlier(const float H[9], float sx, float sy, float dx, float dy, float thr)
{
    float x = __builtin_fmaf(H[0], sx, H[1] * sy) + H[2];
    float y = __builtin_fmaf(H[3], sx, H[4] * sy) + H[5];
    const float z = __builtin_fmaf(H[6], sx, H[7] * sy) + H[8];
    x /= z; y /= z;
    const float ex = dx - x, ey = dy - y;
    return __builtin_fmaf(ex, ex, ey * ey) < thr;
}

