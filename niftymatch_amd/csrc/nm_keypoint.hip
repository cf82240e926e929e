// nm_keypoint.hip -- DoG extrema, one-shot sub-pixel refinement, ordered (raster) stream compaction for gfx950.
// Replaces kernels/keypoint.cu:19-251 and the thrust::copy_if of sift/pyramidata.cu:84-91. The reference's texture
// fetches at (x+0.5,y+0.5) are exact texel loads (utils/cudatex2D.cu:15-19), so planes are read as plain arrays.
#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "nm_keypoint.hpp"
#include "../../include/nm_abi.h"

using nmfp::fma32;

namespace {

template <bool GT>
__device__ __forceinline__ bool is_extremum(const float *__restrict__ cur, const float *__restrict__ dn,
                                            const float *__restrict__ up, int x, int y, int w, float cv)
{
    const size_t c = (size_t)y * w + x;
#define NM_CMP(v) if (GT ? !(cv > (v)) : !(cv < (v))) return false;
    NM_CMP(cur[c - 1]) NM_CMP(cur[c + 1])
    NM_CMP(cur[c - w - 1]) NM_CMP(cur[c - w]) NM_CMP(cur[c - w + 1])
    NM_CMP(cur[c + w - 1]) NM_CMP(cur[c + w]) NM_CMP(cur[c + w + 1])
    NM_CMP(dn[c]) NM_CMP(dn[c - 1]) NM_CMP(dn[c + 1])
    NM_CMP(dn[c - w - 1]) NM_CMP(dn[c - w]) NM_CMP(dn[c - w + 1])
    NM_CMP(dn[c + w - 1]) NM_CMP(dn[c + w]) NM_CMP(dn[c + w + 1])
    NM_CMP(up[c]) NM_CMP(up[c - 1]) NM_CMP(up[c + 1])
    NM_CMP(up[c - w - 1]) NM_CMP(up[c - w]) NM_CMP(up[c - w + 1])
    NM_CMP(up[c + w - 1]) NM_CMP(up[c + w]) NM_CMP(up[c + w + 1])
#undef NM_CMP
    return true;
}

// Where refine() takes its 3 x 3 x 3 neighbourhood from: three DoG planes, or four Gaussian levels whose differences they are
// (frame driver without materialised DoG planes: dn = l1 - l0, cur = l2 - l1, up = l3 - l2, the subtraction of
// cudamath.cu:26-35 / siftfunctions.cu:42-51 done on the fly).
struct DogPlanes {
    const float *__restrict__ cur, *__restrict__ dn, *__restrict__ up;
    __device__ __forceinline__ float c(size_t i) const { return cur[i]; }
    __device__ __forceinline__ float d(size_t i) const { return dn[i]; }
    __device__ __forceinline__ float u(size_t i) const { return up[i]; }
};
struct LevelPlanes {
    const float *__restrict__ l0, *__restrict__ l1, *__restrict__ l2, *__restrict__ l3;
    __device__ __forceinline__ float c(size_t i) const { return l2[i] - l1[i]; }
    __device__ __forceinline__ float d(size_t i) const { return l1[i] - l0[i]; }
    __device__ __forceinline__ float u(size_t i) const { return l3[i] - l2[i]; }
};

// keypoint.cu:108-180. The float/double mix of the reference is kept literally; a*b+c contractions are explicit.
template <typename PL>
__device__ __forceinline__ bool refine_at(const PL &pl, int x, int y, int w, float peak, float edge,
                                          float xper, float sigma0, int num_dogs, int level, float4 &out)
{
    const size_t o = (size_t)y * w + x;
#define C_(dx, dy) pl.c(o + (dy) * w + (dx))
#define D_(dx, dy) pl.d(o + (dy) * w + (dx))
#define U_(dx, dy) pl.u(o + (dy) * w + (dx))
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
    float4 A0 = fxx > 0 ? make_float4(fxx, fxy, fxs, -fx) : make_float4(-fxx, -fxy, -fxs, fx);
    float4 A1 = fxy > 0 ? make_float4(fxy, fyy, fys, -fy) : make_float4(-fxy, -fyy, -fys, fy);
    float4 A2 = fxs > 0 ? make_float4(fxs, fys, fss, -fs) : make_float4(-fxs, -fys, -fss, fs);
    float4 t;
    const float max_a = __builtin_fmaxf(__builtin_fmaxf(A0.x, A1.x), A2.x);
    if (!((double)max_a >= 1e-10)) return false;
    if (max_a == A1.x)      { t = A1; A1 = A0; A0 = t; }
    else if (max_a == A2.x) { t = A2; A2 = A0; A0 = t; }
    A0.y /= A0.x; A0.z /= A0.x; A0.w /= A0.x;
    A1.y = fma32(-A1.x, A0.y, A1.y); A1.z = fma32(-A1.x, A0.z, A1.z); A1.w = fma32(-A1.x, A0.w, A1.w);
    A2.y = fma32(-A2.x, A0.y, A2.y); A2.z = fma32(-A2.x, A0.z, A2.z); A2.w = fma32(-A2.x, A0.w, A2.w);
    if (__builtin_fabsf(A2.y) > __builtin_fabsf(A1.y)) { t = A2; A2 = A1; A1 = t; }
    if (!((double)__builtin_fabsf(A1.y) >= 1e-10)) return false;
    A1.z /= A1.y; A1.w /= A1.y;
    A2.z = fma32(-A2.y, A1.z, A2.z); A2.w = fma32(-A2.y, A1.w, A2.w);
    if (!((double)__builtin_fabsf(A2.z) >= 1e-10)) return false;
    const float ds = A2.w / A2.z;
    const float dy = fma32(-ds, A1.z, A1.w);
    const float dx = fma32(-dy, A0.y, fma32(-ds, A0.z, A0.w));
    const float tt = fma32(ds, fs, fma32(dx, fx, dy * fy));
    const float v = (float)((double)c + 0.5 * (double)tt);
    const float tr = fxx + fyy;
    const float s = (tr * tr) / fma32(fxx, fyy, -(fxy * fxy));
    const float ethr = ((edge + 1) * (edge + 1)) / edge;
    if ((__builtin_fabsf(v) > peak) && s < ethr && __builtin_fabsf(dx) < 1 && __builtin_fabsf(dy) < 1 &&
        __builtin_fabsf(ds) < 1) {
        out.x = ((float)x + dx) * xper;
        out.y = ((float)y + dy) * xper;
        out.z = (float)(((double)sigma0 * nmfp::exp2_spec((double)((float)level + ds) / (double)num_dogs)) * (double)xper);
        out.w = (float)level;
        return true;
    }
    return false;
}
__device__ __forceinline__ bool refine(const float *__restrict__ cur, const float *__restrict__ dn,
                                       const float *__restrict__ up, int x, int y, int w, float peak, float edge,
                                       float xper, float sigma0, int num_dogs, int level, float4 &out)
{
    return refine_at(DogPlanes{cur, dn, up}, x, y, w, peak, edge, xper, sigma0, num_dogs, level, out);
}

// Bilinear, border-addressed, unnormalised fetch of the full-resolution mask (utils/cudatex2D.cu:15-19).
__device__ __forceinline__ float mask_fetch(const float *__restrict__ mask, int mw, int mh, float u, float v)
{
    const float xb = u - 0.5f, yb = v - 0.5f;
    const float fi = __builtin_floorf(xb), fj = __builtin_floorf(yb);
    const float a = xb - fi, b = yb - fj;
    const int i = (int)fi, j = (int)fj;
    auto T = [&](int ii, int jj) -> float {
        return (ii >= 0 && ii < mw && jj >= 0 && jj < mh) ? mask[(size_t)jj * mw + ii] : 0.f;
    };
    return (1 - a) * (1 - b) * T(i, j) + a * (1 - b) * T(i + 1, j) + (1 - a) * b * T(i, j + 1) + a * b * T(i + 1, j + 1);
}

// One test + refinement of pixel (x,y); true when accepted.
__device__ __forceinline__ bool detect_pixel(const float *__restrict__ cur, const float *__restrict__ dn,
                                             const float *__restrict__ up, int x, int y, int w, int h, float peak,
                                             float edge, float xper, float sigma0, int num_dogs, int level,
                                             float4 &kp)
{
    if (x < 1 || x > w - 2 || y < 1 || y > h - 2) return false;
    const float c = cur[(size_t)y * w + x];
    const float thr = 0.8f * peak;
    const bool cand = (c <= thr && is_extremum<false>(cur, dn, up, x, y, w, c)) ||
                      (c >= thr && is_extremum<true>(cur, dn, up, x, y, w, c));
    if (!cand) return false;
    return refine(cur, dn, up, x, y, w, peak, edge, xper, sigma0, num_dogs, level, kp);
}

// ---- API kernels: dense float4 map ----
__global__ __launch_bounds__(256) void find_keypoints_dense_kernel(const float *__restrict__ cur,
                                                                  const float *__restrict__ dn,
                                                                  const float *__restrict__ up,
                                                                  const float *__restrict__ mask, int mw, int mh,
                                                                  int w, int h, float peak, float edge, float xper,
                                                                  float sigma0, int num_dogs, int level,
                                                                  float4 *__restrict__ result)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    if (mask) {
        if (x < 1 || x > w - 2 || y < 1 || y > h - 2) return;
        if (mask_fetch(mask, mw, mh, ((float)x + 0.5f) * xper, ((float)y + 0.5f) * xper) < 1.f) return;
    }
    float4 kp;
    if (detect_pixel(cur, dn, up, x, y, w, h, peak, edge, xper, sigma0, num_dogs, level, kp))
        result[(size_t)y * w + x] = kp;
}

// Ordered in-block compaction helper: returns this thread's rank among the block's flagged threads (block-wide,
// in thread order) and the block total. 256 threads = 4 waves.
__device__ __forceinline__ int block_rank(bool flag, int &total, int *s_wave /* [4] */)
{
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = s_wave[i];
        if (i < wave) off += c;
        total += c;
    }
    return off + rank;
}

__global__ __launch_bounds__(256) void count_valid_kernel(const float4 *__restrict__ dense, int n,
                                                         int *__restrict__ counts)
{
    __shared__ int s_wave[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool f = (i < n) && (dense[i].w >= 0);
    int total;
    block_rank(f, total, s_wave);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scatter_valid_kernel(const float4 *__restrict__ dense, int n,
                                                           const int *__restrict__ offsets,
                                                           float4 *__restrict__ out)
{
    __shared__ int s_wave[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = make_float4(-1, -1, -1, -1);
    if (i < n) v = dense[i];
    const bool f = (i < n) && (v.w >= 0);
    int total;
    const int r = block_rank(f, total, s_wave);
    if (f) out[offsets[blockIdx.x] + r] = v;
}

struct NmCompact3 { const float4 *dense[3]; float4 *out[3]; int *counts; int *offsets; int *totals; int n, nb; };

__global__ __launch_bounds__(256) void count_valid3_kernel(NmCompact3 c)
{
    __shared__ int s_wave[4];
    const int l = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool f = (i < c.n) && (c.dense[l][i].w >= 0);
    int total;
    block_rank(f, total, s_wave);
    if (threadIdx.x == 0) c.counts[l * c.nb + blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scatter_valid3_kernel(NmCompact3 c)
{
    __shared__ int s_wave[4];
    const int l = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = make_float4(-1, -1, -1, -1);
    if (i < c.n) v = c.dense[l][i];
    const bool f = (i < c.n) && (v.w >= 0);
    int total;
    const int r = block_rank(f, total, s_wave);
    if (f) c.out[l][c.offsets[l * c.nb + blockIdx.x] + r] = v;
}

// Exclusive scan of n ints by ONE workgroup of 1024 threads; returns the total to every thread.
__device__ int block_exclusive_scan_1024(const int *__restrict__ in, int *__restrict__ out, int n, int *s /* [1024+1] */)
{
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int beg = t * per, end = min(beg + per, n);
    int sum = 0;
    for (int i = beg; i < end; ++i) sum += in[i];
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan
        int v = 0;
        if (t >= d) v = s[t - d];
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    const int total = s[1023];
    int run = s[t] - sum;
    for (int i = beg; i < end; ++i) { const int c = in[i]; out[i] = run; run += c; }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void scan_counts_kernel(const int *__restrict__ counts, int *__restrict__ offsets,
                                                          int n, int *__restrict__ total_out)
{
    __shared__ int s[1024];
    const int total = block_exclusive_scan_1024(counts, offsets, n, s);
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

__global__ __launch_bounds__(1024) void scan_counts3_kernel(NmCompact3 c)
{
    __shared__ int s[1024];
    for (int l = 0; l < 3; ++l) {
        const int total = block_exclusive_scan_1024(c.counts + l * c.nb, c.offsets + l * c.nb, c.nb, s);
        if (threadIdx.x == 0) c.totals[l] = total;
    }
}

// ---- frame-driver kernels: detect 3 levels of one octave straight into per-unit staging, then scan + book-keeping,
//      then gather into the output-ordered keypoint list ----
// A unit is a 256-pixel segment of one image row (units in raster order: u = y * nseg + seg). One workgroup per unit,
// one pixel per thread. Every lane loads its own column of the 3 rows x 5 DoG planes (15 coalesced 256-B row segments
// per wave); horizontal neighbours come from the adjacent lanes by DPP wave shifts, the two segment-edge lanes fetch
// their halo explicitly. The 26-neighbour strict extremum test is branch-free: per plane, max3/min3 of each row, then
//   is_max(level l) = c > max(M9[l], M9[l+2], M8[l+1]),  M9 = max of a plane's 3x3, M8 = the 3x3 without its centre.
// Only accepted candidates (rare) run the divergent sub-pixel refinement from global memory.
__device__ __forceinline__ float dpp_from_lower(float own, float edge)   // lane i <- lane i-1, lane 0 <- edge
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper(float own, float edge)   // lane i <- lane i+1, lane 63 <- edge
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

constexpr int DET_ROWS = 5;      // image rows per workgroup: 7 rows are loaded for 5 tested. 4 was best while the kernel read
                                 // 5 DoG planes; reading 6 level planes, 5 rows are 0.6-0.7 % of the headline better (same box,
                                 // three alternating runs); the 12 sub-lists per row must fit the 64-lane scan: <= 5

template <bool DENSE, bool LEV = false, bool MASKED = DENSE>
__global__ __launch_bounds__(256) void detect_stage_kernel(NmDetectArgs a)
{
    __shared__ unsigned char s_x[DET_ROWS][3][4][64];    // candidate lanes per (row, level, wave), in lane order
    __shared__ int s_cnt[DET_ROWS * 12 + 1];              // sub-list lengths, then their exclusive scan (+ total)
    __shared__ int s_acc[DET_ROWS * 3], s_last[DET_ROWS * 3], s_wtot[4], s_pref[257];
    const int frame = blockIdx.y;
    const float *const *dog = LEV ? a.lev[frame] : a.dog[frame];       // LEV: the six Gaussian levels
    const int seg = blockIdx.x % a.nseg, yg = blockIdx.x / a.nseg;
    const int y0 = yg * DET_ROWS;
    const int x = seg * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ow = a.ow, oh = a.oh;
    const bool xin = x < ow;
    const int xc = xin ? x : ow - 1;                       // clamped column for safe addressing
    const int xe = (lane == 0) ? max(xc - 1, 0) : min(xc + 1, ow - 1);
    const bool edge_lane = (lane == 0) || (lane == 63);
    const float thr = 0.8f * a.peak;
    const float *const mask = MASKED ? (DENSE ? a.mask : a.masks[frame]) : nullptr;

    // sliding 3-row window per plane: row maxima/minima of (left, mid, right); centre row keeps mid and max/min(l, r).
    // Rows are FETCHED one iteration ahead of being absorbed into the window (raw values wait in registers), so the
    // global-load latency of row y+2 hides behind the tests of row y instead of stalling every iteration.
    float rmax[5][3], rmin[5][3], cmid[5][2], clr_max[5][2], clr_min[5][2];
    float raw_mid[2][5], raw_ev[2][5];
    auto fetch_row = [&](int yy, int buf) {
        const int yr = min(max(yy, 0), oh - 1);
        if (LEV) {                              // DoG p = level p + 1 - level p, formed here instead of read
            float lm[6], le[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const float *row = dog[p] + (size_t)yr * ow;
                lm[p] = row[xc];
                le[p] = 0.f;
                if (edge_lane) le[p] = row[xe];
            }
#pragma unroll
            for (int p = 0; p < 5; ++p) { raw_mid[buf][p] = lm[p + 1] - lm[p]; raw_ev[buf][p] = le[p + 1] - le[p]; }
            return;
        }
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float *row = dog[p] + (size_t)yr * ow;
            raw_mid[buf][p] = row[xc];
            raw_ev[buf][p] = 0.f;
            if (edge_lane) raw_ev[buf][p] = row[xe];
        }
    };
    auto absorb_row = [&](int buf, int slot, int cslot) {
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float mid = raw_mid[buf][p], ev = raw_ev[buf][p];
            const float lf = dpp_from_lower(mid, ev), rt = dpp_from_upper(mid, ev);
            rmax[p][slot] = max3f(lf, mid, rt);
            rmin[p][slot] = min3f(lf, mid, rt);
            cmid[p][cslot] = mid;
            clr_max[p][cslot] = __builtin_fmaxf(lf, rt);
            clr_min[p][cslot] = __builtin_fminf(lf, rt);
        }
    };
    // rows y0-1, y0 absorbed, y0+1 in flight before the loop; iteration j absorbs y+1 and fetches y+2
    fetch_row(y0 - 1, 0);
    fetch_row(y0, 1);
    absorb_row(0, 0, 0);
    fetch_row(y0 + 1, 0);
    absorb_row(1, 1, 1);
#pragma unroll
    for (int j = 0; j < DET_ROWS; ++j) {
        const int y = y0 + j;
        const int s_up = j % 3, s_c = (j + 1) % 3, s_dn = (j + 2) % 3;     // rows y-1, y, y+1
        const int cs = (j + 1) & 1;                                        // centre-row slot of row y
        if (j + 1 < DET_ROWS) fetch_row(y + 2, (j + 1) & 1);
        absorb_row(j & 1, s_dn, j & 1);
        bool interior = xin && x >= 1 && x <= ow - 2 && y >= 1 && y <= oh - 2;
        if (MASKED && mask && interior)      // masked detection: the bilinear border fetch of the full-resolution mask must be 1
            interior = mask_fetch(mask, a.mask_w, a.mask_h, ((float)x + 0.5f) * a.xper, ((float)y + 0.5f) * a.xper) >= 1.f;
        float m9[5], n9[5], m8[5], n8[5];
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            m9[p] = max3f(rmax[p][s_up], rmax[p][s_c], rmax[p][s_dn]);
            n9[p] = min3f(rmin[p][s_up], rmin[p][s_c], rmin[p][s_dn]);
            m8[p] = max3f(rmax[p][s_up], rmax[p][s_dn], clr_max[p][cs]);
            n8[p] = min3f(rmin[p][s_up], rmin[p][s_dn], clr_min[p][cs]);
        }
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const float cv = cmid[level + 1][cs];
            const bool is_max = cv > max3f(m9[level], m9[level + 2], m8[level + 1]);
            const bool is_min = cv < min3f(n9[level], n9[level + 2], n8[level + 1]);
            const bool cand = interior && ((cv <= thr && is_min) || (cv >= thr && is_max));
            // candidates are only LISTED here (per row, level and wave, in lane = column order); they are refined after the
            // scan by consecutive lanes. Refining in place made whole waves run the ~150-instruction refinement for the
            // one or two candidate lanes they hold, three times per row: most of the kernel's VALU work.
            const unsigned long long m = __ballot(cand);
            if (cand) s_x[j][level][wave][__popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)lane;
            if (lane == 0) s_cnt[(j * 3 + level) * 4 + wave] = __popcll(m);
            if (DENSE && !cand && xin && y < oh)     // API path: every pixel of the dense maps is written exactly once
                reinterpret_cast<float4 *>(a.dense[level])[(size_t)y * ow + x] = make_float4(-1.f, -1.f, -1.f, -1.f);
        }
    }
    __syncthreads();
    // exclusive scan of the 48 sub-list lengths -> flattened candidate order (row, level, column): s_cnt[q] = first index
    if (wave == 0) {
        const int c = lane < DET_ROWS * 12 ? s_cnt[lane] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane < DET_ROWS * 12) s_cnt[lane] = incl - c;
        if (lane == DET_ROWS * 12 - 1) s_cnt[DET_ROWS * 12] = incl;
        if (lane < DET_ROWS * 3) s_acc[lane] = 0;
    }
    __syncthreads();
    const int total = s_cnt[DET_ROWS * 12];
    for (int b0 = 0; b0 < total; b0 += 256) {           // 256 candidates per pass (a 4 x 256 pixel unit usually holds < 100)
        const int c = b0 + (int)threadIdx.x;
        bool acc = false;
        float4 kp = make_float4(-1.f, -1.f, -1.f, -1.f);
        int g = 0, px = 0, py = 0, lvl = 0;
        if (c < total) {
            int lo = 0, hi = DET_ROWS * 12;               // largest q with s_cnt[q] <= c (empty sub-lists share their successor's start)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cnt[mid] <= c) lo = mid; else hi = mid;
            }
            g = lo >> 2;                                  // (row, level) group
            const int jr = g / 3, w = lo & 3;
            lvl = g - 3 * jr;
            px = seg * 256 + w * 64 + s_x[jr][lvl][w][c - s_cnt[lo]];
            py = y0 + jr;
            if (LEV)
                acc = refine_at(LevelPlanes{dog[lvl], dog[lvl + 1], dog[lvl + 2], dog[lvl + 3]}, px, py, ow, a.peak, a.edge, a.xper,
                                a.sigma0, a.num_dogs, lvl, kp);
            else
                acc = refine(dog[lvl + 1], dog[lvl], dog[lvl + 2], px, py, ow, a.peak, a.edge, a.xper, a.sigma0, a.num_dogs, lvl, kp);
        }
        if (DENSE) {
            if (c < total) reinterpret_cast<float4 *>(a.dense[lvl])[(size_t)py * ow + px] = kp;    // kp stays -1 when rejected
            continue;
        }
        // ordered compaction of the accepted candidates inside their (row, level) group
        const unsigned long long m = __ballot(acc);
        const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int before = in_wave;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < wave) before += s_wtot[w];
        s_pref[threadIdx.x] = before;                     // accepted candidates of this pass before candidate c
        if (threadIdx.x == 255) s_pref[256] = before + (acc ? 1 : 0);
        __syncthreads();
        if (c < total) {
            const int gs = s_cnt[g * 4], ge = s_cnt[g * 4 + 4];       // the group's flattened range (s_cnt[48] = total)
            const int base = max(gs, b0) - b0;
            const int pos = s_acc[g] + before - s_pref[base];
            if (acc) {
                const int unit = py * a.nseg + seg;
                float4 *st = reinterpret_cast<float4 *>(a.staging[frame]) + (size_t)lvl * a.stage_stride + (size_t)unit * 256;
                st[pos] = kp;
            }
            if (c == min(ge, b0 + 256) - 1) s_last[g] = pos + (acc ? 1 : 0);     // accepted so far, this pass included
        }
        __syncthreads();
        if (threadIdx.x < DET_ROWS * 3) {
            const int gs = s_cnt[threadIdx.x * 4], ge = s_cnt[threadIdx.x * 4 + 4];
            if (ge > b0 && gs < b0 + 256 && ge > gs) s_acc[threadIdx.x] = s_last[threadIdx.x];
        }
        __syncthreads();
    }
    if (!DENSE && threadIdx.x < DET_ROWS * 3) {           // per-unit counts of the three levels
        const int jr = threadIdx.x / 3, lv = threadIdx.x - 3 * jr;
        const int y = y0 + jr;
        if (y < oh) a.counts[frame][lv * a.n_blocks + y * a.nseg + seg] = s_acc[threadIdx.x];
    }
}

__global__ __launch_bounds__(1024) void scan_book_kernel(NmScanArgs a)
{
    __shared__ int s[1024];
    __shared__ int totals[3];
    const int frame = blockIdx.x;
    for (int l = 0; l < 3; ++l) {
        const int tot = block_exclusive_scan_1024(a.counts[frame] + l * a.n_blocks, a.offsets[frame] + l * a.n_blocks, a.n_blocks, s);
        if (threadIdx.x == 0) totals[l] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        NmFrameBook *b = a.book[frame];
        int num_items = (a.octave == 0) ? 0 : b->num_items;
        b->oct_base[a.octave] = num_items;
        bool live = true;                         // sift/siftfunctions.cu:145,160: an empty level ends the octave
        for (int l = 0; l < 3; ++l) {
            int cnt = live ? totals[l] : 0;
            if (cnt == 0) live = false;
            int n = cnt;
            if (n + num_items > a.capacity) n = a.capacity - num_items;   // siftfunctions.cu:165-169
            if (n < 0) n = 0;
            b->lvl_count[a.octave][l] = cnt;
            b->lvl_base[a.octave][l] = num_items;
            b->lvl_n[a.octave][l] = n;
            num_items += n;
        }
        b->num_items = num_items;
        b->oct_base[a.octave + 1] = num_items;
        if (a.d_num_items[frame]) *a.d_num_items[frame] = num_items;
    }
}

// One thread per OUTPUT slot of (frame, level): the unit that holds slot `pos` is the last one whose exclusive offset is
// <= pos (binary search over the scanned offsets; empty units share their successor's offset and are skipped by the
// upper bound). A grid over the units instead would launch hundreds of thousands of empty workgroups per octave.
__global__ __launch_bounds__(256) void gather_stage_kernel(NmGatherArgs a)
{
    const int level = blockIdx.y, frame = blockIdx.z;
    const NmFrameBook *book = a.book[frame];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= book->lvl_n[a.octave][level]) return;
    const int *off = a.offsets[frame] + level * a.n_blocks;
    int lo = 0, hi = a.n_blocks;                      // invariant: off[lo] <= pos, (hi == n_blocks or off[hi] > pos)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    const float4 *st = reinterpret_cast<const float4 *>(a.staging[frame]) + (size_t)level * a.stage_stride + (size_t)lo * 256;
    reinterpret_cast<float4 *>(a.kpts[frame])[book->lvl_base[a.octave][level] + pos] = st[pos - off[lo]];
}

}  // namespace

int nm_launch_detect_octave(const NmDetectArgs &d, const NmScanArgs &s, const NmGatherArgs &g, hipStream_t stream)
{
    if (d.n_blocks <= 0 || d.n <= 0) return 0;
    const dim3 grid(d.nseg * nm_divup(d.oh, DET_ROWS), d.n);
    if (d.from_levels && d.any_mask) hipLaunchKernelGGL((detect_stage_kernel<false, true, true>), grid, dim3(256), 0, stream, d);
    else if (d.from_levels) hipLaunchKernelGGL((detect_stage_kernel<false, true, false>), grid, dim3(256), 0, stream, d);
    else if (d.any_mask) hipLaunchKernelGGL((detect_stage_kernel<false, false, true>), grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((detect_stage_kernel<false, false, false>), grid, dim3(256), 0, stream, d);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_book_kernel, dim3(s.n), dim3(1024), 0, stream, s);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gather_stage_kernel, dim3(nm_divup(min(g.capacity, g.n_blocks * 256), 256), 3, g.n), dim3(256), 0, stream, g);
    NM_LAUNCH_CHECK();
    return 0;
}

__global__ void book_counts_kernel(const NmFrameBook *book, int *counts)
{
    if (threadIdx.x < 3) counts[threadIdx.x] = book->lvl_n[0][threadIdx.x];
}

extern "C" {

// The frame driver's detection of ONE octave on caller-provided DoG planes: fused 3-level detection straight into ordered
// lists (no dense maps), with the orchestration rules of compute_orientations / compute_descriptors applied (an empty
// level ends the octave, sift/siftfunctions.cu:145,160; at most `capacity` keypoints in total, :165-169). out: capacity
// float4, the levels' raster-ordered lists back to back; d_counts: 3 device ints = keypoints kept per level.
size_t nm_find_keypoints3_compact_workspace_bytes(int width, int height)
{
    const size_t nb = (size_t)(height > 0 ? height : 1) * nm_divup(width > 0 ? width : 1, 256);
    return 3 * nb * 256 * 16 + 2 * 3 * nb * sizeof(int) + 1024;
}

int nm_find_keypoints3_compact_f32(const float *const dog[5], int width, int height, float peak_threshold,
                                   float edge_threshold, float xper, float sigma_0, int num_dogs, int capacity, float *out,
                                   int *d_counts, void *workspace, void *stream)
{
    if (width <= 0 || height <= 0 || capacity <= 0) return 0;
    if (!dog || !out || !d_counts || !workspace) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    const int nseg = nm_divup(width, 256), n_blocks = height * nseg;
    char *base = static_cast<char *>(workspace);
    NmFrameBook *book = reinterpret_cast<NmFrameBook *>(base);        // 1 KB reserved
    float *staging = reinterpret_cast<float *>(base + 1024);
    const size_t stage_stride = (size_t)n_blocks * 256;
    int *counts = reinterpret_cast<int *>(base + 1024 + 3 * stage_stride * 16);
    int *offsets = counts + 3 * (size_t)n_blocks;
    static_assert(sizeof(NmFrameBook) <= 1024, "book fits its slot");
    NmDetectArgs d{};
    NmScanArgs s{};
    NmGatherArgs g{};
    d.n = s.n = g.n = 1;
    d.ow = width; d.oh = height; d.peak = peak_threshold; d.edge = edge_threshold; d.xper = xper; d.sigma0 = sigma_0;
    d.num_dogs = num_dogs; d.stage_stride = stage_stride; d.n_blocks = n_blocks; d.nseg = nseg;
    for (int i = 0; i < 5; ++i) { if (!dog[i]) return (int)hipErrorInvalidValue; d.dog[0][i] = dog[i]; }
    d.staging[0] = staging; d.counts[0] = counts;
    s.counts[0] = counts; s.offsets[0] = offsets; s.book[0] = book; s.n_blocks = n_blocks; s.octave = 0; s.capacity = capacity;
    g.staging[0] = staging; g.counts[0] = counts; g.offsets[0] = offsets; g.book[0] = book; g.kpts[0] = out;
    g.stage_stride = stage_stride; g.n_blocks = n_blocks; g.octave = 0; g.capacity = capacity;
    const int rc = nm_launch_detect_octave(d, s, g, st);
    if (rc) return rc;
    hipLaunchKernelGGL(book_counts_kernel, dim3(1), dim3(64), 0, st, book, d_counts);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_find_keypoints_masked_f32(const float *current, const float *mask, int mask_width, int mask_height,
                                 const float *down, const float *up, int width, int height, float peak_threshold,
                                 float edge_threshold, float xper, float sigma_0, int num_dogs, int dog, float *result,
                                 void *stream)
{
    if (width <= 0 || height <= 0) return 0;
    dim3 grid(nm_divup(width, 64), nm_divup(height, 4));
    hipLaunchKernelGGL(find_keypoints_dense_kernel, grid, dim3(256), 0, nm_stream(stream), current, down, up, mask,
                       mask_width, mask_height, width, height, peak_threshold, edge_threshold, xper, sigma_0, num_dogs,
                       dog, reinterpret_cast<float4 *>(result));
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_find_keypoints_f32(const float *current, const float *down, const float *up, int width, int height,
                          float peak_threshold, float edge_threshold, float xper, float sigma_0, int num_dogs, int dog,
                          float *result, void *stream)
{
    return nm_find_keypoints_masked_f32(current, nullptr, 0, 0, down, up, width, height, peak_threshold, edge_threshold,
                                        xper, sigma_0, num_dogs, dog, result, stream);
}

// find_keypoints for the three searched DoG levels of an octave in ONE launch (the loop of compute_keypoints,
// sift/siftfunctions.cu:100-134). Every pixel of the first width*height entries of result[0..2] is written.
int nm_find_keypoints3_f32(const float *const dog[5], const float *mask, int mask_width, int mask_height, int width,
                           int height, float peak_threshold, float edge_threshold, float xper, float sigma_0, int num_dogs,
                           float *const result[3], void *stream)
{
    if (width <= 0 || height <= 0) return 0;
    if (!dog || !result) return (int)hipErrorInvalidValue;
    NmDetectArgs d{};
    d.n = 1; d.ow = width; d.oh = height; d.peak = peak_threshold; d.edge = edge_threshold; d.xper = xper;
    d.sigma0 = sigma_0; d.num_dogs = num_dogs;
    d.nseg = nm_divup(width, 256); d.n_blocks = height * d.nseg;
    for (int i = 0; i < 5; ++i) { if (!dog[i]) return (int)hipErrorInvalidValue; d.dog[0][i] = dog[i]; }
    for (int l = 0; l < 3; ++l) { if (!result[l]) return (int)hipErrorInvalidValue; d.dense[l] = result[l]; }
    d.mask = mask; d.mask_w = mask_width; d.mask_h = mask_height;
    hipLaunchKernelGGL(detect_stage_kernel<true>, dim3(d.nseg * nm_divup(height, DET_ROWS), 1), dim3(256), 0,
                       nm_stream(stream), d);
    NM_LAUNCH_CHECK();
    return 0;
}

size_t nm_compact3_workspace_bytes(int num_pixels)
{
    const size_t nb = (size_t)nm_divup(num_pixels > 0 ? num_pixels : 1, 256);
    return 6 * nb * sizeof(int);
}

// The stable compaction of gpu_collate_keypoints_for_level (sift/pyramidata.cu:84-88) for three dense maps at once:
// three launches instead of nine. d_counts: 3 ints on the device.
int nm_compact_keypoints3(const float *const dense[3], int num_pixels, float *const out[3], int *d_counts, void *workspace,
                          void *stream)
{
    hipStream_t st = nm_stream(stream);
    if (!dense || !out || !d_counts || !workspace) return (int)hipErrorInvalidValue;
    if (num_pixels <= 0) return (int)hipMemsetAsync(d_counts, 0, 3 * sizeof(int), st);
    NmCompact3 c{};
    c.n = num_pixels; c.nb = nm_divup(num_pixels, 256);
    for (int l = 0; l < 3; ++l) {
        c.dense[l] = reinterpret_cast<const float4 *>(dense[l]);
        c.out[l] = reinterpret_cast<float4 *>(out[l]);
    }
    c.counts = static_cast<int *>(workspace); c.offsets = c.counts + 3 * c.nb; c.totals = d_counts;
    hipLaunchKernelGGL(count_valid3_kernel, dim3(c.nb, 3), dim3(256), 0, st, c);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_counts3_kernel, dim3(1), dim3(1024), 0, st, c);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_valid3_kernel, dim3(c.nb, 3), dim3(256), 0, st, c);
    NM_LAUNCH_CHECK();
    return 0;
}

size_t nm_compact_workspace_bytes(int num_pixels)
{
    const size_t nb = (size_t)nm_divup(num_pixels > 0 ? num_pixels : 1, 256);
    return 2 * nb * sizeof(int);
}

int nm_compact_keypoints(const float *dense, int num_pixels, float *out, int *d_count, void *workspace, void *stream)
{
    hipStream_t st = nm_stream(stream);
    if (num_pixels <= 0) return (int)hipMemsetAsync(d_count, 0, sizeof(int), st);
    const int nb = nm_divup(num_pixels, 256);
    int *counts = static_cast<int *>(workspace), *offsets = counts + nb;
    hipLaunchKernelGGL(count_valid_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<const float4 *>(dense), num_pixels, counts);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, nb, d_count);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_valid_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<const float4 *>(dense), num_pixels,
                       offsets, reinterpret_cast<float4 *>(out));
    NM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
