/*
 * oracle/nm_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the SIFT detect/describe + brute-force L2 match path of gift-surg/NiftyMatch
 * (reference tree at /root/reference, cited below as path:line relative to src/gpu/). It is the parity checker
 * for the HIP library and the timed CPU baseline of bench.py. Nothing in the product links, imports or calls it;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path (its tests live in a
 * private external repo, CONTRIBUTING.md:7,17) and is CUDA-only, so it cannot be compiled or run here. This file
 * is pinned by hand-derived known-answer tests (tests/test_oracle_kat.py) and by the Q-table of SURVEY.md 8(a),
 * which fixes every place where the reference's behaviour is undefined (races, atomics order, stale memory).
 * The ONE exception: nmo_sift_params (SiftParams, siftparams.h:30-51) is pinned by the reference itself -- that header
 * compiles with g++, oracle/siftparams_dump.cpp dumps its fields for 12 geometries into tests/golden/siftparams_ref.json
 * (make -C oracle ref golden) and tests/test_siftparams_pinned.py compares bit for bit.
 *
 * Floating-point contract (shared with the HIP kernels, see DESIGN.md "fp spec"):
 *   - compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf()/fma() below. The places
 *     where nvcc (-fmad=true default) would contract a*b+c are written as the fma LLVM's DAG combiner produces:
 *     (a*b + c*d) -> fma(a,b,c*d); (x - y*z) -> fma(-y,z,x).
 *   - mixed float/double expressions keep the reference's C++ promotion rules literally.
 *   - device libm calls go through oracle/nmo_math.h (one fixed implementation inside CUDA's documented ulp bounds).
 *   - atomics with undefined order get ONE fixed order, stated at each site.
 */
#include "nmo_math.h"
#include "nmo_ransac.h"
#include "nmo_warp.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NMO_API extern "C" __attribute__((visibility("default")))

static const double NMO_2PI_D = 6.283185307179586476925286766559;   /* 2*M_PI */
static const float  NMO_2PI_F = (float)(2 * 3.14159265358979323846);

/* ---------------------------------------------------------------------------------------------------------- */
/* threading control (bench.py's cpu_baseline reports the thread count it used)                                */
NMO_API int nmo_set_threads(int n)
{
#ifdef _OPENMP
    /* n > 0: that many threads; n <= 0: back to the default, one per processor OpenMP sees (omp_get_num_procs honours the
       affinity mask / cgroup the process runs under), so a caller that lowered the count can restore it */
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------------------- */
/* SiftParams -- sift/siftparams.h:30-51                                                                       */
struct nmo_params {
    int width, height, num_octaves, num_dog_levels, level_max, level_min;
    float sigma_d_0, sigma_k, sigma_0, sigma_n, base_smooth, peak_threshold, edge_threshold;
    float sigmas[8];
    int num_sigmas;
};

NMO_API void nmo_sift_params(int width, int height, nmo_params *p)
{
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height;
    p->num_dog_levels = 3; p->sigma_n = 0.5f; p->peak_threshold = 0.f; p->edge_threshold = 10.f;
    p->level_max = p->num_dog_levels + 1;
    p->level_min = -1;
    /* siftparams.h:36 -- MINIMUM_OCTAVE_SIZE 32 */
    p->num_octaves = (int)std::floor(std::log(std::min(width, height) * 2.0 / 32) / std::log(2.0));
    if (p->num_octaves <= 0) p->num_octaves = 1;
    /* :39  std::pow(float,float) -> float */
    p->sigma_k = std::pow(2.0f, 1.0f / p->num_dog_levels);
    p->sigma_0 = 1.6f * p->sigma_k;
    /* :41  float * double sqrt -> narrowed */
    p->sigma_d_0 = (float)((double)p->sigma_0 * std::sqrt(1.0 - 1.0 / (double)(p->sigma_k * p->sigma_k)));
    /* :43  std::pow(float,int) promotes to double */
    float sa = (float)((double)p->sigma_0 * std::pow((double)p->sigma_k, (double)p->level_min));
    float sb = p->sigma_n;
    if (sa > sb) p->base_smooth = std::sqrt(sa * sa - sb * sb);
    p->num_sigmas = 0;
    for (int i = p->level_min + 1; i <= p->level_max; ++i)
        p->sigmas[p->num_sigmas++] = (float)((double)p->sigma_d_0 * std::pow((double)p->sigma_k, (double)i));
}

/* PyramidData::create_kernel_for_sigma -- sift/pyramidata.cu:105-123. taps must hold 2*ceil(4*sigma)+1 floats. */
NMO_API int nmo_create_kernel_for_sigma(float sigma, float *taps)
{
    const int r = (int)(std::ceil(sigma * 4));
    const int len = 2 * r + 1;
    float sum = 0.f;
    for (int j = 0; j < len; ++j) {
        float val = ((float)j - r) / sigma;
        val = (float)std::exp(-0.5 * (val * val));      /* host double exp, as the reference */
        if (taps) taps[j] = val;
        sum += val;
    }
    if (taps) for (int j = 0; j < len; ++j) taps[j] = taps[j] / sum;
    return r;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* convolve -- kernels/convolution.cu:16-159. Zero padding (Q2), taps k=-r..r, sum = fma(x, w[r-k], sum) from 0,
 * rows into `buffer`, then columns of `buffer` into `result`. Q3: every pixel written exactly once.           */
NMO_API void nmo_convolve(float *result, const float *image, float *buffer, int width, int height,
                          const float *kernel, int r)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        const float *row = image + (size_t)y * width;
        float *out = buffer + (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            float sum = 0.f;
            for (int k = -r; k <= r; ++k) {
                const int xx = x + k;
                const float v = (xx >= 0 && xx < width) ? row[xx] : 0.f;
                sum = std::fmaf(v, kernel[r - k], sum);
            }
            out[x] = sum;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        float *out = result + (size_t)y * width;
        for (int x = 0; x < width; ++x) {
            float sum = 0.f;
            for (int k = -r; k <= r; ++k) {
                const int yy = y + k;
                const float v = (yy >= 0 && yy < height) ? buffer[(size_t)yy * width + x] : 0.f;
                sum = std::fmaf(v, kernel[r - k], sum);
            }
            out[x] = sum;
        }
    }
}

/* downsample_by_2 -- kernels/downsample.cu:6-17 */
NMO_API void nmo_downsample2(float *result, int rw, int rh, const float *source, int sw, int sh)
{
    (void)sh;
    for (int y = 0; y < rh; ++y)
        for (int x = 0; x < rw; ++x)
            result[(size_t)y * rw + x] = source[(size_t)(y * 2) * sw + (x * 2)];
}

/* subtract -- kernels/cudamath.cu:26-35 : C = A - B */
NMO_API void nmo_subtract(const float *A, const float *B, float *C, int width, int height)
{
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) C[i] = A[i] - B[i];
}

static inline float nmo_mod_2pi_f(float x)          /* kernels/cudamath.h:82-87 */
{
    while (x > NMO_2PI_F) x -= NMO_2PI_F;
    while (x < 0.0f) x += NMO_2PI_F;
    return x;
}

/* gradient -- kernels/cudamath.cu:38-54. grad is float2 (mag, angle). Border pixels = (0,0) (Q4). Q5 types. */
NMO_API void nmo_gradient(const float *src, float *grad, int width, int height)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            float *g2 = grad + 2 * ((size_t)y * width + x);
            if (x < 1 || x >= width - 1 || y < 1 || y >= height - 1) { g2[0] = 0.f; g2[1] = 0.f; continue; }
            const float nx = src[(size_t)y * width + x + 1], px = src[(size_t)y * width + x - 1];
            const float ny = src[(size_t)(y + 1) * width + x], py = src[(size_t)(y - 1) * width + x];
            const float dx = nx - px, dy = ny - py;
            const float g = (float)(0.5 * (double)std::sqrt(std::fmaf(dx, dx, dy * dy)));
            float r = 0.0f;
            if (g != 0.0f) r = nmo_mod_2pi_f((float)((double)nmo_atan2f(dy, dx) + NMO_2PI_D));
            g2[0] = g; g2[1] = r;
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* keypoints -- kernels/keypoint.cu:19-201. Planes are plain row-major arrays: the reference's texture fetches
 * at (x+0.5, y+0.5) with linear filtering are exact texel loads (gpu/utils/cudatex2D.cu:15-19).               */
struct nmo_f4 { float x, y, z, w; };

template <bool GT>
static inline bool nmo_is_extremum(const float *cur, const float *dn, const float *up, int x, int y, int w)
{
    const float cv = cur[(size_t)y * w + x];
    const float *planes[3] = {cur, dn, up};
    for (int p = 0; p < 3; ++p)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (p == 0 && dx == 0 && dy == 0) continue;
                const float pv = planes[p][(size_t)(y + dy) * w + (x + dx)];
                if (GT ? !(cv > pv) : !(cv < pv)) return false;
            }
    return true;
}

/* keypoint.cu:108-180 (Q7). Returns true and fills out[4] when the candidate is accepted. */
static inline bool nmo_refine(const float *cur, const float *dn, const float *up, int x, int y, int w,
                              float peak, float edge, float xper, float sigma0, int num_dogs, int level,
                              float *out)
{
#define C_(dx, dy) cur[(size_t)(y + (dy)) * w + (x + (dx))]
#define D_(dx, dy) dn[(size_t)(y + (dy)) * w + (x + (dx))]
#define U_(dx, dy) up[(size_t)(y + (dy)) * w + (x + (dx))]
    const float c = C_(0, 0);
    const float fx = (float)(0.5 * (double)(C_(1, 0) - C_(-1, 0)));
    const float fy = (float)(0.5 * (double)(C_(0, 1) - C_(0, -1)));
    const float fs = (float)(0.5 * (double)(U_(0, 0) - D_(0, 0)));
    const float fxx = (float)((double)(C_(1, 0) + C_(-1, 0)) - 2.0 * (double)c);
    const float fyy = (float)((double)(C_(0, 1) + C_(0, -1)) - 2.0 * (double)c);
    const float fss = (float)((double)(U_(0, 0) + D_(0, 0)) - 2.0 * (double)c);
    const float fxy = (float)(0.25 * (double)(((C_(1, 1) + C_(-1, -1)) - C_(-1, 1)) - C_(1, -1)));
    const float fxs = (float)(0.25 * (double)(((U_(1, 0) + D_(-1, 0)) - U_(-1, 0)) - D_(1, 0)));
    const float fys = (float)(0.25 * (double)(((U_(0, 1) + D_(0, -1)) - U_(0, -1)) - D_(0, 1)));
#undef C_
#undef D_
#undef U_
    nmo_f4 A0 = fxx > 0 ? nmo_f4{fxx, fxy, fxs, -fx} : nmo_f4{-fxx, -fxy, -fxs, fx};
    nmo_f4 A1 = fxy > 0 ? nmo_f4{fxy, fyy, fys, -fy} : nmo_f4{-fxy, -fyy, -fys, fy};
    nmo_f4 A2 = fxs > 0 ? nmo_f4{fxs, fys, fss, -fs} : nmo_f4{-fxs, -fys, -fss, fs};
    nmo_f4 t;
    const float max_a = std::fmax(std::fmax(A0.x, A1.x), A2.x);
    if (!((double)max_a >= 1e-10)) return false;
    if (max_a == A1.x)      { t = A1; A1 = A0; A0 = t; }
    else if (max_a == A2.x) { t = A2; A2 = A0; A0 = t; }
    A0.y /= A0.x; A0.z /= A0.x; A0.w /= A0.x;
    A1.y = std::fmaf(-A1.x, A0.y, A1.y); A1.z = std::fmaf(-A1.x, A0.z, A1.z); A1.w = std::fmaf(-A1.x, A0.w, A1.w);
    A2.y = std::fmaf(-A2.x, A0.y, A2.y); A2.z = std::fmaf(-A2.x, A0.z, A2.z); A2.w = std::fmaf(-A2.x, A0.w, A2.w);
    if (std::fabs(A2.y) > std::fabs(A1.y)) { t = A2; A2 = A1; A1 = t; }
    if (!((double)std::fabs(A1.y) >= 1e-10)) return false;
    A1.z /= A1.y; A1.w /= A1.y;
    A2.z = std::fmaf(-A2.y, A1.z, A2.z); A2.w = std::fmaf(-A2.y, A1.w, A2.w);
    if (!((double)std::fabs(A2.z) >= 1e-10)) return false;
    const float ds = A2.w / A2.z;
    const float dy = std::fmaf(-ds, A1.z, A1.w);
    const float dx = std::fmaf(-dy, A0.y, std::fmaf(-ds, A0.z, A0.w));
    const float tt = std::fmaf(ds, fs, std::fmaf(dx, fx, dy * fy));
    const float v = (float)((double)c + 0.5 * (double)tt);
    const float tr = fxx + fyy;
    const float s = (tr * tr) / std::fmaf(fxx, fyy, -(fxy * fxy));
    const float ethr = ((edge + 1) * (edge + 1)) / edge;
    if ((std::fabs(v) > peak) && s < ethr && std::fabs(dx) < 1 && std::fabs(dy) < 1 && std::fabs(ds) < 1) {
        out[0] = ((float)x + dx) * xper;
        out[1] = ((float)y + dy) * xper;
        out[2] = (float)(((double)sigma0 * nmo_exp2((double)((float)level + ds) / (double)num_dogs)) * (double)xper);
        out[3] = (float)level;
        return true;
    }
    return false;
}

/* CUDA bilinear fetch of the full-resolution mask at texture coordinate (u,v), border addressing, unnormalised
 * (cudatex2D.cu:15-19, keypoint.cu:214). The hardware keeps 8 fractional bits; for xper = 2^o the fractional
 * part is 0 (o = 0) or 0.5 (o >= 1), both exact in 1.8 fixed point.                                           */
static inline float nmo_mask_fetch(const float *mask, int mw, int mh, float u, float v)
{
    const float xb = u - 0.5f, yb = v - 0.5f;
    const float fi = std::floor(xb), fj = std::floor(yb);
    const float a = xb - fi, b = yb - fj;
    const int i = (int)fi, j = (int)fj;
    auto T = [&](int ii, int jj) -> float {
        return (ii >= 0 && ii < mw && jj >= 0 && jj < mh) ? mask[(size_t)jj * mw + ii] : 0.f;
    };
    return (1 - a) * (1 - b) * T(i, j) + a * (1 - b) * T(i + 1, j) + (1 - a) * b * T(i, j + 1) + a * b * T(i + 1, j + 1);
}

/* find_keypoints -- keypoint.cu:183-251. `result` is the dense W*H float4 map the CALLER pre-filled with -1
 * (siftfunctions.cu:120-121); only accepted pixels are written. mask == NULL -> unmasked overload.          */
NMO_API void nmo_find_keypoints(const float *cur, const float *dn, const float *up, const float *mask, int mask_w,
                                int mask_h, int width, int height, float peak, float edge, float xper,
                                float sigma0, int num_dogs, int level, float *result)
{
#pragma omp parallel for schedule(static)
    for (int y = 1; y <= height - 2; ++y) {
        for (int x = 1; x <= width - 2; ++x) {
            if (mask && nmo_mask_fetch(mask, mask_w, mask_h, ((float)x + 0.5f) * xper, ((float)y + 0.5f) * xper) < 1.f)
                continue;
            const float c = cur[(size_t)y * width + x];
            const float thr = 0.8f * peak;
            if ((c <= thr && nmo_is_extremum<false>(cur, dn, up, x, y, width)) ||
                (c >= thr && nmo_is_extremum<true>(cur, dn, up, x, y, width)))
                nmo_refine(cur, dn, up, x, y, width, peak, edge, xper, sigma0, num_dogs, level,
                           result + 4 * ((size_t)y * width + x));
        }
    }
}

/* PyramidData::gpu_collate_keypoints_for_level -- sift/pyramidata.cu:84-91: stable copy_if(w >= 0) (Q8). */
NMO_API int nmo_compact_keypoints(const float *dense, int num_pixels, float *out)
{
    int n = 0;
    for (int i = 0; i < num_pixels; ++i)
        if (dense[4 * (size_t)i + 3] >= 0) { std::memcpy(out + 4 * (size_t)n, dense + 4 * (size_t)i, 16); ++n; }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* orientations -- kernels/orientation.cu:11-129 (semantics), :132-216 (intent of the racy parts), Q10/Q11.
 * FIXED ORDER (the reference's shared-memory atomicAdd order is undefined): the clipped window (at most 21 x 21)
 * is cut into 63 strips: strip p = rx + 21 * (ry / 7), rx = cx - xmin, ry = cy - ymin. Each strip sums its (at most
 * 7) votes per bin in increasing cy; a bin is the sum of its 63 strip partials taken in increasing p, starting
 * from the first. Smoothing is the race-free circular 3-tap mean of :181-192. `result` is float2 per keypoint and
 * must be pre-filled with (-1,-1) by the caller (pyramidata.cu:90).                                           */
/* The raw (unsmoothed) 36-bin histogram of one keypoint in the FIXED ORDER above. Optionally also the ORDER-FREE envelope data
 * of the same votes: h64[i] = binary64 sum of the float votes of bin i (exact to 1e-16 relative, whatever the order), nv[i] =
 * their number -- any order the reference's shared-memory atomicAdd may take lies within (nv[i] - 1) u of h64[i]
 * (tests/test_oracle_order_envelope.py).                                                                          */
static inline void nmo_orient_raw_hist(const float *kp, const float *grad, int ow, int oh, float gauss_factor, float xper,
                                       float hist[36], double *h64, int *nv)
{
    const int NBINS = 36;
    const float x = kp[0] / xper, y = kp[1] / xper, s = kp[2] / xper;
    const int xi = (int)((double)x + 0.5), yi = (int)((double)y + 0.5);
    const float sigma_w = gauss_factor * s;
    int W = std::max((int)std::floor(3 * sigma_w), 1);
    W = std::min(22 / 2 - 1, W);                              /* blockDim (22,22) -> 10 (orientation.cu:29-30) */
    const long grad_index = ((long)kp[3] * oh + yi) * ow + xi; /* Q10: integer arithmetic */
    const float *g = grad + 2 * grad_index;
    float part[63][NBINS];
    for (int p = 0; p < 63; ++p) for (int i = 0; i < NBINS; ++i) part[p][i] = 0.f;
    if (h64) for (int i = 0; i < NBINS; ++i) { h64[i] = 0.0; nv[i] = 0; }
    const int xmin = std::max(-W, -xi), xmax = std::min(W, ow - 1 - xi);
    const int ymin = std::max(-W, -yi), ymax = std::min(W, oh - 1 - yi);
    const float denom = (2 * sigma_w) * sigma_w;
    for (int cy = ymin; cy <= ymax; ++cy)
        for (int cx = xmin; cx <= xmax; ++cx) {
            const float dx = (float)(cx + xi) - x, dy = (float)(cy + yi) - y;
            const float r2 = std::fmaf(dx, dx, dy * dy);
            if (!((double)r2 < (double)(W * W) + 0.6)) continue;
            const float wgt = nmo_expf(r2 / denom);                               /* exp(+...) per Q11 */
            const float *gp = g + 2 * ((long)cy * ow + cx);
            const float q = (float)((double)(36.0f * gp[1]) / NMO_2PI_D);
            const int bin = (int)std::floor(q);
            const float vote = gp[0] * wgt;
            part[(cx - xmin) + 21 * ((cy - ymin) / 7)][bin % NBINS] += vote;
            if (h64) { h64[bin % NBINS] += (double)vote; nv[bin % NBINS] += 1; }
        }
    for (int i = 0; i < NBINS; ++i) {
        float h = part[0][i];
        for (int p = 1; p < 63; ++p) h += part[p][i];
        hist[i] = h;
    }
}

NMO_API void nmo_detect_orientations(const float *key_pts, const float *grad, int num_pts, int ow, int oh,
                                     float gauss_factor, float xper, float *result)
{
    const int NBINS = 36;
#pragma omp parallel for schedule(dynamic, 16)
    for (int pt = 0; pt < num_pts; ++pt) {
        const float *kp = key_pts + 4 * (size_t)pt;
        if (kp[3] < 0) continue;
        float hist[NBINS];
        nmo_orient_raw_hist(kp, grad, ow, oh, gauss_factor, xper, hist, nullptr, nullptr);
        for (int iter = 0; iter < 6; ++iter) {
            float prev = hist[NBINS - 1];
            const float first = hist[0];
            int i;
            for (i = 0; i < NBINS - 1; ++i) {
                const float newh = (float)((double)((prev + hist[i]) + hist[(i + 1) % NBINS]) / 3.0);
                prev = hist[i];
                hist[i] = newh;
            }
            hist[i] = (float)((double)((prev + hist[i]) + first) / 3.0);
        }
        float maxh = 0.f;
        for (int i = 0; i < NBINS; ++i) maxh = std::fmax(maxh, hist[i]);
        const float threshold = (float)((double)maxh * 0.8);
        int nangles = 0;
        for (int i = 0; i < NBINS && nangles < 2; ++i) {
            const float h0 = hist[i], hm = hist[(i - 1 + NBINS) % NBINS], hp = hist[(i + 1 + NBINS) % NBINS];
            if (h0 > threshold && h0 > hm && h0 > hp) {
                const float di = (float)((-0.5 * (double)(hp - hm)) / (double)((hp + hm) - 2 * h0));
                const float th = (float)((NMO_2PI_D * ((double)((float)i + di) + 0.5)) / (double)NBINS);
                result[2 * (size_t)pt + nangles] = th;
                ++nangles;
            }
        }
    }
}

/* Envelope of the orientation histograms (test infrastructure for the summation-order freedom, Q11): per keypoint the raw
 * 36 bins as the oracle sums them (hist32), the order-free binary64 sums of the same votes (hist64) and the votes per bin. */
NMO_API void nmo_orientation_envelope(const float *key_pts, const float *grad, int num_pts, int ow, int oh, float gauss_factor,
                                      float xper, float *hist32, double *hist64, int *nvotes)
{
    for (int pt = 0; pt < num_pts; ++pt) {
        const float *kp = key_pts + 4 * (size_t)pt;
        if (kp[3] < 0) {
            for (int i = 0; i < 36; ++i) { hist32[36 * (size_t)pt + i] = 0.f; hist64[36 * (size_t)pt + i] = 0.0; nvotes[36 * (size_t)pt + i] = 0; }
            continue;
        }
        nmo_orient_raw_hist(kp, grad, ow, oh, gauss_factor, xper, hist32 + 36 * (size_t)pt, hist64 + 36 * (size_t)pt,
                            nvotes + 36 * (size_t)pt);
    }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* descriptors -- kernels/descriptor.cu:32-145, Q12. Only the DIAGONAL 16x16 chunks of the window vote (cx and
 * cy advance together, :142-143); exp(+...) window; no normalisation; first orientation only.
 * FIXED ORDER (the reference's global atomicAdd order is undefined): within a chunk, the samples of column
 * tx = cx - chunk origin (0..15) vote into partial histogram tx in increasing cy, each sample's 8 votes in
 * (dbinx, dbiny, dbint) order; chunks in increasing order. bin = balanced pairwise tree over the 16 partials:
 * stride 1,2,4,8: v[i] += v[i+stride].                                                                      */
/* One window sample of the descriptor (descriptor.cu:102-117), with the reference's float / double promotions: everything
 * a vote needs except the 8 trilinear factors. Shared by the fixed-order oracle below and by the order-free envelope.   */
struct nmo_desc_sample { float mod, win, rbinx, rbiny, rbint; int binx, biny, bint; };
static inline nmo_desc_sample nmo_desc_sample_at(const float *gptr, int ow, int cx, int cy, int xi, int yi, float x, float y,
                                                 float angle0, double st0, double ct0, float SBP)
{
    nmo_desc_sample r;
    r.mod = gptr[2 * ((long)cy * ow + cx)];
    const float ang = gptr[2 * ((long)cy * ow + cx) + 1];
    const float theta = nmo_mod_2pi_f(ang - angle0);
    const float dx = (float)(xi + cx) - x, dy = (float)(yi + cy) - y;
    const float nx = (float)(std::fma(ct0, (double)dx, st0 * (double)dy) / (double)SBP);
    const float ny = (float)(std::fma(-st0, (double)dx, ct0 * (double)dy) / (double)SBP);
    const float nt = (float)((double)(8.0f * theta) / NMO_2PI_D);
    r.win = (float)nmo_exp((double)std::fmaf(nx, nx, ny * ny) / 8.0);
    r.binx = (int)std::floor((double)nx - 0.5);
    r.biny = (int)std::floor((double)ny - 0.5);
    r.bint = (int)std::floor(nt);
    r.rbinx = (float)((double)nx - ((double)r.binx + 0.5));
    r.rbiny = (float)((double)ny - ((double)r.biny + 0.5));
    r.rbint = nt - (float)r.bint;
    return r;
}

NMO_API void nmo_compute_sift_descriptors(const float *key_pts, const float *orients, const float *grad,
                                          int num_pts, int ow, int oh, int num_dogs, float xper, float *desc,
                                          float *xp, float *yp)
{
    const int NBO = 8, NBP = 4;
    const int binto = 1, binyo = NBO * NBP, binxo = NBO;
#pragma omp parallel for schedule(dynamic, 8)
    for (int pt = 0; pt < num_pts; ++pt) {
        const float *kp = key_pts + 4 * (size_t)pt;
        const float x = kp[0] / xper, y = kp[1] / xper, s = kp[2] / xper;
        const int xi = (int)((double)x + 0.5), yi = (int)((double)y + 0.5), si = (int)kp[3];
        if (xi < 0 || xi >= ow || yi < 0 || yi >= oh || si < 0 || si >= num_dogs) continue;
        const float SBP = (float)((double)(3 * s) + 1.e-07);
        const int W = (int)std::floor(std::sqrt(2.0) * (double)SBP * (NBP + 1) / 2.0 + 0.5);
        const int xmin = std::max(-W, -xi), xmax = std::min(W, ow - 1 - xi);
        const int ymin = std::max(-W, -yi), ymax = std::min(W, oh - 1 - yi);
        const int max_dims = std::max(xmax - xmin, ymax - ymin);
        const int chunks = (int)std::ceil((max_dims + 1.f) / 16);
        xp[pt] = kp[0]; yp[pt] = kp[1];
        const float *gptr = grad + 2 * (((long)si * oh + yi) * ow + xi);
        const float angle0 = orients[2 * (size_t)pt];
        const double st0 = (double)nmo_sinf(angle0), ct0 = (double)nmo_cosf(angle0);
        /* 16 partial histograms (one per window column modulo 16), each 16 cells x NINE temporal slots: a sample's two
         * temporal votes go to slots (bint & 7) and (bint & 7) + 1, so slot 8 collects what wraps round to orientation bin 0
         * (descriptor.cu:136: (bint + dbint) % NBO); it is added to bin 0 after the partials have been combined. The reference
         * adds with atomicAdd in an undefined order (descriptor.cu:137): this order is the spec (DESIGN.md "fp spec"). */
        std::vector<float> part(16 * 144, 0.f);           /* part[tx*144 + 9 * cell + slot], cell = (binx+2) + 4 (biny+2) */
        for (int c = 0; c < chunks; ++c)
            for (int q = 0; q < 4; ++q)
                for (int L = 0; L < 64; ++L) {
                    const int p = 64 * q + L;
                    const int cx = (p & 15) + xmin + 16 * c, cy = (p >> 4) + ymin + 16 * c;
                    if (!(cx <= xmax && cy <= ymax)) continue;
                    const nmo_desc_sample sm = nmo_desc_sample_at(gptr, ow, cx, cy, xi, yi, x, y, angle0, st0, ct0, SBP);
                    const float mod = sm.mod, win = sm.win, rbinx = sm.rbinx, rbiny = sm.rbiny, rbint = sm.rbint;
                    const int binx = sm.binx, biny = sm.biny, bint = sm.bint;
                    for (int dbx = 0; dbx < 2; ++dbx)
                        for (int dby = 0; dby < 2; ++dby)
                            for (int dbt = 0; dbt < 2; ++dbt) {
                                if (binx + dbx >= -(NBP / 2) && binx + dbx < (NBP / 2) &&
                                    biny + dby >= -(NBP / 2) && biny + dby < (NBP / 2)) {
                                    const float wt = win * mod * std::fabs((1.f - dbx) - rbinx) *
                                                     std::fabs((1.f - dby) - rbiny) * std::fabs((1.f - dbt) - rbint);
                                    const int cell = (binx + dbx + NBP / 2) + NBP * (biny + dby + NBP / 2);
                                    part[(p & 15) * 144 + 9 * cell + (bint & (NBO - 1)) + dbt] += wt;
                                }
                            }
                }
        for (int stride = 1; stride < 16; stride *= 2)
            for (int i = 0; i < 16; i += 2 * stride)
                for (int b = 0; b < 144; ++b) part[i * 144 + b] += part[(i + stride) * 144 + b];
        /* descriptor element (binx + 2) * binxo + (biny + 2) * binyo + t (descriptor.cu:81,136); bin 0 = slot 0 + slot 8 */
        for (int cell = 0; cell < 16; ++cell)
            for (int t = 0; t < NBO; ++t) {
                const int e = (cell & 3) * binxo + (cell >> 2) * binyo + t * binto;
                desc[128 * (size_t)pt + e] = (t == 0) ? part[9 * cell] + part[9 * cell + 8] : part[9 * cell + t];
            }
    }
}

/* Envelope of the descriptors (test infrastructure for the summation-order freedom, Q12): a LITERAL walk of the reference's
 * kernel -- 16 x 16 threads, cx and cy advancing together per chunk (descriptor.cu:86-87,142-143), location
 * (binx + dbinx) binxo + (biny + dbiny) binyo + ((bint + dbint) binto) % NBO relative to the centre (:81,136) -- that adds every
 * float vote into a binary64 accumulator: sum64[128 pt + e] is the order-free value of descriptor element e (exact to 1e-16
 * relative), nvotes its number of votes. Any order of the reference's global atomicAdd (:137), the oracle's nine-slot order
 * included, lies within (nvotes - 1) u of it. Returns the number of samples whose bint is outside [0, NBO] (must be 0: the
 * nine-slot layout's `bint & 7` equals the reference's `% NBO` only for those).                                          */
NMO_API int nmo_descriptor_envelope(const float *key_pts, const float *orients, const float *grad, int num_pts, int ow, int oh,
                                    int num_dogs, float xper, double *sum64, int *nvotes)
{
    const int NBO = 8, NBP = 4;
    const int binto = 1, binyo = NBO * NBP, binxo = NBO;
    int bad = 0;
    for (int pt = 0; pt < num_pts; ++pt) {
        double *d64 = sum64 + 128 * (size_t)pt;
        int *nv = nvotes + 128 * (size_t)pt;
        for (int e = 0; e < 128; ++e) { d64[e] = 0.0; nv[e] = 0; }
        const float *kp = key_pts + 4 * (size_t)pt;
        const float x = kp[0] / xper, y = kp[1] / xper, s = kp[2] / xper;
        const int xi = (int)((double)x + 0.5), yi = (int)((double)y + 0.5), si = (int)kp[3];
        if (xi < 0 || xi >= ow || yi < 0 || yi >= oh || si < 0 || si >= num_dogs) continue;
        const float SBP = (float)((double)(3 * s) + 1.e-07);
        const int W = (int)std::floor(std::sqrt(2.0) * (double)SBP * (NBP + 1) / 2.0 + 0.5);
        const int xmin = std::max(-W, -xi), xmax = std::min(W, ow - 1 - xi);
        const int ymin = std::max(-W, -yi), ymax = std::min(W, oh - 1 - yi);
        const int max_dims = std::max(xmax - xmin, ymax - ymin);
        const int chunks = (int)std::ceil((max_dims + 1.f) / 16);
        const float *gptr = grad + 2 * (((long)si * oh + yi) * ow + xi);
        const float angle0 = orients[2 * (size_t)pt];
        const double st0 = (double)nmo_sinf(angle0), ct0 = (double)nmo_cosf(angle0);
        const int centre = (NBP / 2) * binyo + (NBP / 2) * binxo;
        for (int ty = 0; ty < 16; ++ty)
            for (int tx = 0; tx < 16; ++tx) {
                int cx = tx + xmin, cy = ty + ymin;
                for (int i = 0; i < chunks; ++i, cx += 16, cy += 16) {
                    if (!(cx <= xmax && cy <= ymax)) continue;
                    const nmo_desc_sample sm = nmo_desc_sample_at(gptr, ow, cx, cy, xi, yi, x, y, angle0, st0, ct0, SBP);
                    if (sm.bint < 0 || sm.bint > NBO) { ++bad; continue; }
                    for (int dbinx = 0; dbinx < 2; ++dbinx)
                        for (int dbiny = 0; dbiny < 2; ++dbiny)
                            for (int dbint = 0; dbint < 2; ++dbint)
                                if (sm.binx + dbinx >= -(NBP / 2) && sm.binx + dbinx < (NBP / 2) &&
                                    sm.biny + dbiny >= -(NBP / 2) && sm.biny + dbiny < (NBP / 2)) {
                                    const float wt = sm.win * sm.mod * std::fabs((1.f - dbinx) - sm.rbinx) *
                                                     std::fabs((1.f - dbiny) - sm.rbiny) * std::fabs((1.f - dbint) - sm.rbint);
                                    const int loc = (sm.binx + dbinx) * binxo + (sm.biny + dbiny) * binyo +
                                                    ((sm.bint + dbint) * binto) % NBO;
                                    d64[centre + loc] += (double)wt;
                                    nv[centre + loc] += 1;
                                }
                }
            }
    }
    return bad;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* matcher -- kernels/transpose.cu:9-30, kernels/match.cu:14-117, sift/siftfunctions.cu:15-40               */
NMO_API void nmo_transpose(float *out, const float *in, int width, int height)
{
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) out[(size_t)x * height + y] = in[(size_t)y * width + x];
}

/* compute_brute_force_distance: At is dim x size_A, B is size_B x dim, D is size_B x size_A (match.cu:36-47). */
NMO_API void nmo_bf_distance(const float *At, int size_A, const float *B, int size_B, int dim, float *D)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < size_B; ++j)
        for (int i = 0; i < size_A; ++i) {
            float acc = 0.0f;
            for (int k = 0; k < dim; ++k) {
                const float t = At[(size_t)k * size_A + i] - B[(size_t)j * dim + k];
                acc = std::fmaf(t, t, acc);
            }
            D[(size_t)j * size_A + i] = acc;
        }
}

/* get_sift_matches -- match.cu:83-117 (Q14). result[i] untouched when min2 <= 0. */
NMO_API void nmo_get_sift_matches(const float *distance, int rows, int cols, int buffer_width, int *result,
                                  float ambiguity)
{
    for (int i = 0; i < rows; ++i) {
        const float *row = distance + (size_t)i * buffer_width;
        float min1 = row[0];
        float min2 = (float)0x7f800000;       /* int -> float conversion, 2139095040.0f, NOT +inf (match.cu:91) */
        int idx = 0;
        for (int j = 1; j < cols; ++j) {
            const float cur = row[j];
            if (cur < min1) { min2 = min1; idx = j; min1 = cur; }
            else if (cur < min2) min2 = cur;
        }
        if (min2 > 0) {
            const float a = min1 / min2;
            result[i] = (a < ambiguity) ? idx : -1;
        }
    }
}

/* compute_sift_matches -- siftfunctions.cu:15-40. distance is nA x nB (may be NULL: extension, skip the store).
 * Also returns per-row (min1, idx, min2) when the pointers are non-NULL (used to check the multi-GPU merge). */
NMO_API void nmo_sift_matches(const float *A, int nA, const float *B, int nB, float *distance, int *result,
                              float ambiguity, float *min1_out, int *idx_out, float *min2_out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; ++i) {
        float min1 = 0.f, min2 = (float)0x7f800000; int idx = 0;
        for (int j = 0; j < nB; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 128; ++k) {
                const float t = A[(size_t)i * 128 + k] - B[(size_t)j * 128 + k];
                acc = std::fmaf(t, t, acc);
            }
            if (distance) distance[(size_t)i * nB + j] = acc;
            if (j == 0) { min1 = acc; idx = 0; }
            else if (acc < min1) { min2 = min1; idx = j; min1 = acc; }
            else if (acc < min2) min2 = acc;
        }
        if (min1_out) min1_out[i] = min1;
        if (idx_out) idx_out[i] = idx;
        if (min2_out) min2_out[i] = min2;
        if (min2 > 0) {
            const float a = min1 / min2;
            result[i] = (a < ambiguity) ? idx : -1;
        }
    }
}

/* Multi-GPU building blocks (no reference counterpart: the reference is single-GPU). What a rank holding the candidate rows
 * [index_offset, index_offset + nB) reports per query, such that nmo_sift_match_merge over the shards in ascending order
 * reproduces the reference's scan over the whole set (match.cu:91-105) for EVERY input, including distances above
 * 2139095040 and non-finite ones:
 *   - the scan's comparisons skip a NaN distance at any candidate but the very first (`current < x` is false), while a
 *     NaN at global candidate 0 stays in min_1_distance for good; so a shard leaves NaN distances out, except that the
 *     shard holding global candidate 0 (index_offset == 0) reports min1 = NaN, idx = 0 when that distance is NaN;
 *   - when the scan replaces its minimum it OVERWRITES min_2_distance with the previous minimum (match.cu:97), so for a
 *     row whose minimum is not at candidate 0 min2 is the true second smallest, also above 2139095040; only a row whose
 *     minimum sits at candidate 0 keeps the initial 2139095040.0f as an upper bound. A shard therefore reports min2 =
 *     the smallest of its OTHER non-NaN distances, +inf when there is none, unclamped; the merge applies the clamp.
 * A shard with no distance below +inf reports (+inf, idx 0 if it holds global candidate 0 else -1, min2 as above).   */
NMO_API void nmo_sift_match_shard(const float *A, int nA, const float *B, int nB, int index_offset, float *min1_out,
                                  int *idx_out, float *min2_out)
{
    const float inf = std::numeric_limits<float>::infinity();
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; ++i) {
        float m1 = inf, m2 = inf; int idx = -1;
        bool first_nan = false;
        for (int j = 0; j < nB; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 128; ++k) {
                const float t = A[(size_t)i * 128 + k] - B[(size_t)j * 128 + k];
                acc = std::fmaf(t, t, acc);
            }
            if (j == 0 && index_offset == 0 && acc != acc) first_nan = true;
            if (acc < m1) { m2 = m1; m1 = acc; idx = j; }        /* strict <: lowest index wins; NaN and +inf fall through */
            else if (acc < m2) m2 = acc;
        }
        if (first_nan) { min1_out[i] = std::numeric_limits<float>::quiet_NaN(); idx_out[i] = 0; min2_out[i] = m1; continue; }
        min1_out[i] = m1;
        idx_out[i] = idx >= 0 ? idx + index_offset : (index_offset == 0 && nB > 0 ? 0 : -1);
        min2_out[i] = m2;
    }
}

/* Merge of shard-major (n_shards x nA) triples in ascending shard order, then the ratio test of match.cu:107-116.
 * A NaN min1 (global candidate 0) survives every comparison, as in the scan. */
NMO_API void nmo_sift_match_merge(const float *min1, const int *idx1, const float *min2, int n_shards, int nA, int *result,
                                  float ambiguity)
{
    for (int i = 0; i < nA; ++i) {
        float m1 = std::numeric_limits<float>::infinity(), m2 = m1; int idx = -1;      /* the neutral (empty shard) triple */
        for (int g = 0; g < n_shards; ++g) {
            const float a1 = min1[(size_t)g * nA + i], a2 = min2[(size_t)g * nA + i];
            /* a NaN minimum can only come from the shard holding global candidate 0; every shard before it is empty */
            if (a1 != a1) { m1 = a1; idx = 0; m2 = a2; }
            else if (a1 < m1) { m2 = (m1 < a2) ? m1 : a2; m1 = a1; idx = idx1[(size_t)g * nA + i]; }
            else if (a1 < m2) m2 = a1;
        }
        /* the minimum never left candidate 0 (idx -1: no distance below +inf anywhere; the ratio test gives -1 either way) */
        if (idx <= 0 && (float)0x7f800000 < m2) m2 = (float)0x7f800000;
        if (m2 > 0) {
            const float a = m1 / m2;
            result[i] = (a < ambiguity) ? idx : -1;
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* Per-frame driver: the client loop implied by the data-structure contracts, calling the stage
 * functions above in the reference's orchestration order (sift/siftfunctions.cu:42-181).
 * Outputs (all optional except desc/x/y): kpts_out = compacted float4 list in output order, orient_out = float2.
 * Returns the number of descriptors written (<= capacity, Q13).                                             */
/* The same client loop with the reference's run-time knobs: SiftParams::_peak_threshold / _edge_threshold are public
 * fields passed per call (sift/siftparams.h:97-98, siftfunctions.cu:123-125), and compute_keypoints_with_mask
 * (siftfunctions.cu:65-98) takes a full-resolution mask texture sampled at ((x+0.5) xper, (y+0.5) xper)
 * (keypoint.cu:204-224). mask: width x height floats or NULL.                                                 */
NMO_API int nmo_sift_detect_describe_ex(const float *gray, int width, int height, int capacity, float peak_threshold,
                                        float edge_threshold, const float *mask, float *desc, float *xs, float *ys,
                                        float *kpts_out, float *orient_out, int *counts_out);

NMO_API int nmo_sift_detect_describe(const float *gray, int width, int height, int capacity, float *desc,
                                     float *xs, float *ys, float *kpts_out, float *orient_out,
                                     int *counts_out /* [num_octaves*3] accepted per (octave, level) or NULL */)
{
    nmo_params P; nmo_sift_params(width, height, &P);
    return nmo_sift_detect_describe_ex(gray, width, height, capacity, P.peak_threshold, P.edge_threshold, nullptr, desc, xs,
                                       ys, kpts_out, orient_out, counts_out);
}

/* optional envelope outputs of the driver (nmo_sift_detect_describe_envelope), indexed by output item */
struct nmo_envelope_out { double *desc64; int *desc_nv; float *ohist32; double *ohist64; int *ohist_nv; int bad_bint; };

static int nmo_detect_describe_impl(const float *gray, int width, int height, int capacity, float peak_threshold,
                                    float edge_threshold, const float *mask, float *desc, float *xs, float *ys,
                                    float *kpts_out, float *orient_out, int *counts_out, nmo_envelope_out *env)
{
    nmo_params P; nmo_sift_params(width, height, &P);
    P.peak_threshold = peak_threshold; P.edge_threshold = edge_threshold;
    const size_t npix = (size_t)width * height;
    const int nlev = P.level_max - P.level_min + 1;          /* 6 (pyramidata.cu:28) */
    const int ndog = P.level_max - P.level_min;              /* 5 */
    std::vector<std::vector<float>> oct(nlev, std::vector<float>(npix, 0.f)), dog(ndog, std::vector<float>(npix, 0.f));
    std::vector<float> buffer(npix, 0.f), grad(2 * npix * ndog, 0.f);
    std::vector<float> keymap(4 * npix), coll[3], orient[3];
    std::vector<float> base_k(2 * (int)std::ceil(P.base_smooth * 4) + 1);
    const int base_r = nmo_create_kernel_for_sigma(P.base_smooth, base_k.data());
    std::vector<std::vector<float>> taps(P.num_sigmas); std::vector<int> radii(P.num_sigmas);
    for (int i = 0; i < P.num_sigmas; ++i) {
        taps[i].resize(2 * (int)std::ceil(P.sigmas[i] * 4) + 1);
        radii[i] = nmo_create_kernel_for_sigma(P.sigmas[i], taps[i].data());
    }
    int num_items = 0;
    nmo_convolve(oct[0].data(), gray, buffer.data(), width, height, base_k.data(), base_r);
    for (int o = 0; o < P.num_octaves; ++o) {
        const int ow = width >> o, oh = height >> o;
        const float xper = (float)std::pow(2.0, o);
        if (o > 0) nmo_downsample2(oct[0].data(), ow, oh, oct[3].data(), width >> (o - 1), height >> (o - 1));
        for (int i = 1; i < nlev; ++i)
            nmo_convolve(oct[i].data(), oct[i - 1].data(), buffer.data(), ow, oh, taps[i - 1].data(), radii[i - 1]);
        for (int i = 0; i < ndog; ++i) nmo_subtract(oct[i + 1].data(), oct[i].data(), dog[i].data(), ow, oh);
        for (int i = P.level_min + 1; i <= P.level_max - 2; ++i)
            nmo_gradient(oct[i + 1].data(), grad.data() + 2 * (size_t)i * ow * oh, ow, oh);
        int cnt[3] = {0, 0, 0};
        for (int i = 1; i < ndog - 1; ++i) {
            for (size_t k = 0; k < 4 * npix; ++k) keymap[k] = -1.0f;
            nmo_find_keypoints(dog[i].data(), dog[i - 1].data(), dog[i + 1].data(), mask, width, height, ow, oh,
                               P.peak_threshold, P.edge_threshold, xper, P.sigma_0, P.num_dog_levels, i - 1,
                               keymap.data());
            coll[i - 1].assign(4 * (size_t)ow * oh, -1.f);
            cnt[i - 1] = nmo_compact_keypoints(keymap.data(), ow * oh, coll[i - 1].data());
        }
        /* compute_orientations / compute_descriptors: an empty level ends the octave (Q9). */
        int live = 0;
        for (int l = 0; l < P.num_dog_levels; ++l) { if (cnt[l] == 0) break; ++live; }
        for (int l = 0; l < P.num_dog_levels; ++l) {
            int n = (l < live) ? cnt[l] : 0;
            if (counts_out) counts_out[o * 3 + l] = n;
            if (n == 0) continue;
            orient[l].assign(2 * (size_t)n, -1.f);
            nmo_detect_orientations(coll[l].data(), grad.data(), n, ow, oh, 1.5f, xper, orient[l].data());
        }
        for (int l = 0; l < live; ++l) {
            int n = cnt[l];
            if (n + num_items > capacity) n = capacity - num_items;
            if (n > 0) {
                nmo_compute_sift_descriptors(coll[l].data(), orient[l].data(), grad.data(), n, ow, oh,
                                             P.num_dog_levels, xper, desc + 128 * (size_t)num_items,
                                             xs + num_items, ys + num_items);
                if (env) {
                    env->bad_bint += nmo_descriptor_envelope(coll[l].data(), orient[l].data(), grad.data(), n, ow, oh,
                                                             P.num_dog_levels, xper, env->desc64 + 128 * (size_t)num_items,
                                                             env->desc_nv + 128 * (size_t)num_items);
                    nmo_orientation_envelope(coll[l].data(), grad.data(), n, ow, oh, 1.5f, xper,
                                             env->ohist32 + 36 * (size_t)num_items, env->ohist64 + 36 * (size_t)num_items,
                                             env->ohist_nv + 36 * (size_t)num_items);
                }
                if (kpts_out) std::memcpy(kpts_out + 4 * (size_t)num_items, coll[l].data(), 16 * (size_t)n);
                if (orient_out) std::memcpy(orient_out + 2 * (size_t)num_items, orient[l].data(), 8 * (size_t)n);
                num_items += n;
            }
        }
    }
    return num_items;
}

NMO_API int nmo_sift_detect_describe_ex(const float *gray, int width, int height, int capacity, float peak_threshold,
                                        float edge_threshold, const float *mask, float *desc, float *xs, float *ys,
                                        float *kpts_out, float *orient_out, int *counts_out)
{
    return nmo_detect_describe_impl(gray, width, height, capacity, peak_threshold, edge_threshold, mask, desc, xs, ys, kpts_out,
                                    orient_out, counts_out, nullptr);
}

/* The same client loop, additionally reporting the ORDER-FREE envelope of both histogram stages for every output item (see
 * nmo_descriptor_envelope / nmo_orientation_envelope): desc64 / desc_nv 128 per item, ohist32 / ohist64 / ohist_nv 36 per item.
 * *bad_bint receives the number of descriptor samples whose temporal bin left [0, 8] (must be 0).                       */
NMO_API int nmo_sift_detect_describe_envelope(const float *gray, int width, int height, int capacity, float *desc, float *xs,
                                              float *ys, float *kpts_out, float *orient_out, double *desc64, int *desc_nv,
                                              float *ohist32, double *ohist64, int *ohist_nv, int *bad_bint)
{
    nmo_params P; nmo_sift_params(width, height, &P);
    nmo_envelope_out env{desc64, desc_nv, ohist32, ohist64, ohist_nv, 0};
    const int n = nmo_detect_describe_impl(gray, width, height, capacity, P.peak_threshold, P.edge_threshold, nullptr, desc, xs,
                                           ys, kpts_out, orient_out, nullptr, &env);
    if (bad_bint) *bad_bint = env.bad_bint;
    return n;
}

/* Stage-level pyramid only (Gaussian levels + DoG + gradients of one octave), for stage parity tests and the
 * cpu_baseline "pyramid" figure. levels: 6 planes, dogs: 5 planes, each ow*oh, contiguous.                  */
NMO_API void nmo_octave_pyramid(const float *level0, int ow, int oh, int width, int height, float *levels,
                                float *dogs, float *grad3 /* 3 planes float2 or NULL */)
{
    nmo_params P; nmo_sift_params(width, height, &P);
    const size_t n = (size_t)ow * oh;
    std::vector<float> buffer(n);
    std::memcpy(levels, level0, n * 4);
    for (int i = 0; i < P.num_sigmas; ++i) {
        std::vector<float> t(2 * (int)std::ceil(P.sigmas[i] * 4) + 1);
        const int r = nmo_create_kernel_for_sigma(P.sigmas[i], t.data());
        nmo_convolve(levels + (i + 1) * n, levels + i * n, buffer.data(), ow, oh, t.data(), r);
    }
    for (int i = 0; i < 5; ++i) nmo_subtract(levels + (i + 1) * n, levels + i * n, dogs + i * n, ow, oh);
    if (grad3) for (int l = 0; l < 3; ++l) nmo_gradient(levels + (l + 1) * n, grad3 + 2 * l * n, ow, oh);
}

/* ---------------------------------------------------------------------------------------------------------- */
/* "next" rows: element-wise stages either side of the path                                   */
/* grayscale -- kernels/bgra_2_gray.cu:9-18: 0.07*B + 0.72*G + 0.21*R in double, narrowed; contraction explicit. */
NMO_API void nmo_grayscale(const unsigned char *bgra, float *out, int width, int height)
{
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        const double b = bgra[4 * i], g = bgra[4 * i + 1], r = bgra[4 * i + 2];
        out[i] = (float)std::fma(0.21, r, std::fma(0.07, b, 0.72 * g));
    }
}
/* extract_channel / put_channel / set_alpha_to_const -- bgra_2_gray.cu:36-113 */
NMO_API void nmo_extract_channel(const unsigned char *bgra, float *out, int width, int height, int channel)
{
    if (channel < 0 || channel > 3) return;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) out[i] = (float)bgra[4 * i + channel];
}
NMO_API void nmo_put_channel(unsigned char *bgra, const float *in, int width, int height, int channel)
{
    if (channel < 0 || channel > 3) return;
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) bgra[4 * i + channel] = (channel == 3) ? 255 : (unsigned char)in[i];
}
NMO_API void nmo_set_alpha(unsigned char *bgra, int width, int height, unsigned char val)
{
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) bgra[4 * i + 3] = val;
}
/* cast<float, unsigned char> -- kernels/cast.cu:8-21 (inputs in [0, 256)) */
NMO_API void nmo_cast_f32_u8(const float *src, size_t cols, size_t rows, unsigned char *dst, unsigned char max_val)
{
    for (size_t i = 0; i < cols * rows; ++i)
        dst[i] = (max_val != 0 && src[i] >= (float)max_val) ? max_val : (unsigned char)src[i];
}
/* downsample_by_2<uchar4> -- kernels/downsample.cu:6-17,32 */
NMO_API void nmo_downsample2_u8x4(unsigned char *result, int rw, int rh, const unsigned char *source, int sw, int sh)
{
    (void)sh;
    for (int y = 0; y < rh; ++y)
        for (int x = 0; x < rw; ++x)
            std::memcpy(result + 4 * ((size_t)y * rw + x), source + 4 * ((size_t)(y * 2) * sw + (x * 2)), 4);
}
/* align_points -- kernels/ransac.cu:29-48 */
NMO_API void nmo_align_points(const float *sx, const float *sy, const float *dx, const float *dy, float *csx, float *csy,
                              float *cdx, float *cdy, const int *matches, int n)
{
    for (int i = 0; i < n; ++i) {
        const int m = matches[i];
        if (m != -1) { csx[i] = sx[i]; csy[i] = sy[i]; cdx[i] = dx[m]; cdy[i] = dy[m]; }
        else { csx[i] = -1; csy[i] = -1; cdx[i] = -1; cdy[i] = -1; }
    }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* N3/N4 warps: undistortion map, texture resampling, perspective warp, mosaicking blend (see nmo_warp.h).      */
/* cuda_undistort -- kernels/undistort.cu:7-46,48-64 */
NMO_API void nmo_undistort_map(const float *x, const float *y, size_t cols, size_t rows, const float *camera_matrix,
                               const float *distortion_coeffs, float *u, float *v)
{
    const float k1 = distortion_coeffs[0], k2 = distortion_coeffs[1], k3 = distortion_coeffs[2];
    const float fx = camera_matrix[0], fy = camera_matrix[1], cx = camera_matrix[2], cy = camera_matrix[3];
    for (size_t p = 0; p < cols * rows; ++p) nmo_undistort_point(x[p], y[p], fx, fy, cx, cy, k1, k2, k3, u[p], v[p]);
}
/* resample_undistort -- resample.cu:100-113,234-248: out = tex(x+0.5, y+0.5) * 255.9999f */
NMO_API void nmo_resample_undistort(const void *tex, int tw, int th, int fmt, const float *x, const float *y,
                                    size_t cols, size_t rows, float *out)
{
    const nmo_tex t{tex, tw, th, fmt};
    for (size_t p = 0; p < cols * rows; ++p) out[p] = nmo_tex2d(t, x[p] + 0.5f, y[p] + 0.5f, 0) * 255.9999f;
}
/* resample_mask -- resample.cu:69-82,207-216 */
NMO_API void nmo_resample_mask(unsigned char *result, const void *tex, int tw, int th, int fmt, int cols, int rows,
                               const float *x, const float *y, float threshold)
{
    const nmo_tex t{tex, tw, th, fmt};
    for (size_t p = 0; p < (size_t)cols * rows; ++p) {
        const float r = nmo_tex2d(t, x[p] + 0.5f, y[p] + 0.5f, 0);
        result[p] = (r <= threshold) ? 0 : (unsigned char)(r * 255.999f);
    }
}
/* resample_perspective_transform -- resample.cu:84-98,116-204: fills x_pos / y_pos, then samples the uchar4 texture */
NMO_API void nmo_resample_perspective(unsigned char *result, const unsigned char *tex, int tw, int th, int cols, int rows,
                                      float *x_pos, float *y_pos, const float *mat3x3, int inverse)
{
    float m[9];
    if (inverse) nmo_invert3x3(mat3x3, m);
    else for (int k = 0; k < 9; ++k) m[k] = mat3x3[k];
    const nmo_tex t{tex, tw, th, NMO_TEX_U8X4N};
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const size_t p = (size_t)y * cols + x;
            nmo_project(m, (float)x, (float)y, x_pos[p], y_pos[p]);
            for (int c = 0; c < 4; ++c)
                result[4 * p + c] = (unsigned char)(nmo_tex2d(t, x_pos[p] + 0.5f, y_pos[p] + 0.5f, c) * 255.9999f);
        }
}
/* transform_blend -- resample.cu:7-66,218-232. frame (uchar4), frame_mask and frame_wts are fw x fh textures. */
NMO_API void nmo_transform_blend(unsigned char *canvas, int cw, int ch, const unsigned char *frame, int fw, int fh, int nw,
                                 int nh, const float *mat3x3, int tx, int ty, const void *mask, int mask_fmt,
                                 float *canvas_wts, const void *wts, int wts_fmt)
{
    const nmo_tex tf{frame, fw, fh, NMO_TEX_U8X4N}, tm{mask, fw, fh, mask_fmt}, tw_{wts, fw, fh, wts_fmt};
    for (int y = 0; y < nh; ++y)
        for (int x = 0; x < nw; ++x) {
            const int px = x + tx, py = y + ty;
            if (px < 0 || px >= cw || py < 0 || py >= ch) continue;
            float xp, yp;
            nmo_project(mat3x3, (float)x, (float)y, xp, yp);
            if (xp >= (float)fw || yp >= (float)fh) continue;
            const float u = xp + 0.5f, v = yp + 0.5f;
            if (nmo_tex2d(tm, u, v, 0) <= 0.5f) continue;
            const float nwt = nmo_tex2d(tw_, u, v, 0);
            const size_t idx = (size_t)py * cw + px;
            float res[3];
            for (int c = 0; c < 3; ++c) res[c] = nmo_tex2d(tf, u, v, c);
            if (canvas_wts[idx] == 0) {
                for (int c = 0; c < 3; ++c) canvas[4 * idx + c] = (unsigned char)(res[c] * 255.9999f);
                canvas[4 * idx + 3] = 255;
                canvas_wts[idx] = nwt;
            } else {
                const float cwt = canvas_wts[idx], sum = cwt + nwt;
                for (int c = 0; c < 3; ++c)
                    canvas[4 * idx + c] =
                        (unsigned char)(std::fmaf(res[c] * nwt, 255.9999f, (float)canvas[4 * idx + c] * cwt) / sum);
                canvas[4 * idx + 3] = 255;
                canvas_wts[idx] = cwt + nwt;
            }
        }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* RANSAC hypotheses -- kernels/ransac.cu:430-521 (one hypothesis per thread), :523-694 (host: best = first maximum).
 * model 0 translation (1 sample), 1 similarity (2), 2 homography (4). rand_list holds iterations * samples point
 * indices, drawn by the caller (the reference draws them on the host from std::mt19937, :543-551). Hypotheses with
 * a repeated index are skipped: their H stays 0 and their inlier count 0, as in the reference's zero-initialised
 * device vectors. Returns the position of the first maximum of the inlier counts; H_best = that hypothesis.       */
NMO_API int nmo_ransac(int model, const float *sx, const float *sy, const float *dx, const float *dy, int n,
                       const int *rand_list, int iterations, float thr, float *H_all, int *inliers, float *H_best)
{
    const int ns = (model == 0) ? 1 : (model == 1) ? 2 : 4;
#pragma omp parallel for schedule(dynamic, 16)
    for (int it = 0; it < iterations; ++it) {
        float *H = H_all + 9 * (size_t)it;
        for (int k = 0; k < 9; ++k) H[k] = 0.f;
        inliers[it] = 0;
        const int *ri = rand_list + (size_t)it * ns;
        bool dup = false;
        for (int a = 0; a < ns; ++a)
            for (int b = a + 1; b < ns; ++b) dup = dup || (ri[a] == ri[b]);
        if (dup) continue;
        float px[4], py[4], qx[4], qy[4];
        for (int a = 0; a < ns; ++a) { px[a] = sx[ri[a]]; py[a] = sy[ri[a]]; qx[a] = dx[ri[a]]; qy[a] = dy[ri[a]]; }
        if (model == 0) nmo_fit_translation(px, py, qx, qy, H);
        else if (model == 1) nmo_fit_similarity(px, py, qx, qy, H);
        else nmo_fit_homography(px, py, qx, qy, H);
        int cnt = 0;
        for (int i = 0; i < n; ++i)
            if (sx[i] >= 0 && nmo_is_inlier(H, sx[i], sy[i], dx[i], dy[i], thr)) ++cnt;
        inliers[it] = cnt;
    }
    int pos = 0;
    for (int it = 1; it < iterations; ++it)
        if (inliers[it] > inliers[pos]) pos = it;                  /* thrust::max_element: first maximum */
    for (int k = 0; k < 9; ++k) H_best[k] = H_all[9 * (size_t)pos + k];
    return pos;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* vectorised math entry points for tests/test_oracle_math.py                                                */
NMO_API void nmo_vec_atan2f(const float *y, const float *x, float *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_atan2f(y[i], x[i]); }
NMO_API void nmo_vec_expf(const float *x, float *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_expf(x[i]); }
NMO_API void nmo_vec_sinf(const float *x, float *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_sinf(x[i]); }
NMO_API void nmo_vec_cosf(const float *x, float *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_cosf(x[i]); }
NMO_API void nmo_vec_exp(const double *x, double *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_exp(x[i]); }
NMO_API void nmo_vec_exp2(const double *x, double *o, int n) { for (int i = 0; i < n; ++i) o[i] = nmo_exp2(x[i]); }
