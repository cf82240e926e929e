// nm_detect_dev.hpp -- device-side body of the frame driver's extrema detection + sub-pixel refinement (kernels/keypoint.cu:19-251),
// shared by detect_stage_kernel (nm_keypoint.hip: one launch per octave) and the octave-tail kernel (nm_tail.hip: a work item
// of a persistent launch). Header-only device code; everything is inline.
#pragma once
#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "nm_devmem.hpp"

namespace nmdet {

using nmfp::fma32;

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

// keypoint.cu:108-180. The reference mixes float and double; every binary64 detour below is replaced by the binary32
// expression that returns the SAME float for all inputs (the candidates are refined by a few lanes per wave while the others
// wait, so the length of this dependent chain is what a detection workgroup's tail costs: 26 % of detect_stage_kernel):
//   (float)(0.5 (double)d), (float)(0.25 (double)d)   = 0.5f d, 0.25f d: the scaling is exact, one rounding either way (also
//                                                       when the result is subnormal: both round the same real number);
//   (float)((double)s - 2.0 (double)c)                = fma(-2, c, s): when the exponents of s and 2 c differ by less than 29
//                                                       the binary64 difference is exact; beyond, the small operand is below
//                                                       2^-29 of the large one and both forms return the large one;
//   (float)((double)c + 0.5 (double)t)                = fma(0.5, t, c), by the same argument;
//   (double)a >= 1e-10                                <=> a >= 0x1.b7cdfep-34f, the smallest float that is >= 1e-10.
// The elimination keeps the reference's order, pivot rule and IEEE divisions; its row swaps are selects and its early exits
// are flags (a rejected candidate runs through divisions by zero whose results are never used).
template <typename PL>
__device__ __forceinline__ bool refine_at(const PL &pl, int x, int y, int w, float peak, float edge,
                                          float xper, float sigma0, int num_dogs, int level, float4 &out)
{
    const size_t o = (size_t)y * w + x;
#define C_(dx, dy) pl.c(o + (dy) * w + (dx))
#define D_(dx, dy) pl.d(o + (dy) * w + (dx))
#define U_(dx, dy) pl.u(o + (dy) * w + (dx))
    const float c = C_(0, 0);
    const float cxp = C_(1, 0), cxm = C_(-1, 0), cyp = C_(0, 1), cym = C_(0, -1), u0 = U_(0, 0), d0 = D_(0, 0);
    const float fx = 0.5f * (cxp - cxm);
    const float fy = 0.5f * (cyp - cym);
    const float fs = 0.5f * (u0 - d0);
    const float fxx = fma32(-2.0f, c, cxp + cxm);
    const float fyy = fma32(-2.0f, c, cyp + cym);
    const float fss = fma32(-2.0f, c, u0 + d0);
    const float fxy = 0.25f * (((C_(1, 1) + C_(-1, -1)) - C_(-1, 1)) - C_(1, -1));
    const float fxs = 0.25f * (((U_(1, 0) + D_(-1, 0)) - U_(-1, 0)) - D_(1, 0));
    const float fys = 0.25f * (((U_(0, 1) + D_(0, -1)) - U_(0, -1)) - D_(0, 1));
#undef C_
#undef D_
#undef U_
    constexpr float TINY = 0x1.b7cdfep-34f;      // bits 0x2EDBE6FF: the smallest float >= 1e-10
    auto sel4 = [](bool k, const float4 &a, const float4 &b) { return make_float4(k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z, k ? a.w : b.w); };
    const float4 R0 = fxx > 0 ? make_float4(fxx, fxy, fxs, -fx) : make_float4(-fxx, -fxy, -fxs, fx);
    const float4 R1 = fxy > 0 ? make_float4(fxy, fyy, fys, -fy) : make_float4(-fxy, -fyy, -fys, fy);
    const float4 R2 = fxs > 0 ? make_float4(fxs, fys, fss, -fs) : make_float4(-fxs, -fys, -fss, fs);
    const float max_a = __builtin_fmaxf(__builtin_fmaxf(R0.x, R1.x), R2.x);
    bool good = max_a >= TINY;
    const bool p1 = max_a == R1.x, p2 = !p1 && (max_a == R2.x);          // pivot row: 1, else 2, else 0
    float4 A0 = sel4(p1, R1, sel4(p2, R2, R0));
    float4 A1 = sel4(p1, R0, R1);
    float4 A2 = sel4(p2, R0, R2);
    A0.y /= A0.x; A0.z /= A0.x; A0.w /= A0.x;
    A1.y = fma32(-A1.x, A0.y, A1.y); A1.z = fma32(-A1.x, A0.z, A1.z); A1.w = fma32(-A1.x, A0.w, A1.w);
    A2.y = fma32(-A2.x, A0.y, A2.y); A2.z = fma32(-A2.x, A0.z, A2.z); A2.w = fma32(-A2.x, A0.w, A2.w);
    const bool sw = __builtin_fabsf(A2.y) > __builtin_fabsf(A1.y);
    const float b1y = sw ? A2.y : A1.y, b1z = sw ? A2.z : A1.z, b1w = sw ? A2.w : A1.w;
    const float b2y = sw ? A1.y : A2.y, b2z = sw ? A1.z : A2.z, b2w = sw ? A1.w : A2.w;
    good = good && (__builtin_fabsf(b1y) >= TINY);
    const float e1z = b1z / b1y, e1w = b1w / b1y;
    const float e2z = fma32(-b2y, e1z, b2z), e2w = fma32(-b2y, e1w, b2w);
    good = good && (__builtin_fabsf(e2z) >= TINY);
    const float ds = e2w / e2z;
    const float dy = fma32(-ds, e1z, e1w);
    const float dx = fma32(-dy, A0.y, fma32(-ds, A0.z, A0.w));
    const float tt = fma32(ds, fs, fma32(dx, fx, dy * fy));
    const float v = fma32(0.5f, tt, c);
    const float tr = fxx + fyy;
    const float s = (tr * tr) / fma32(fxx, fyy, -(fxy * fxy));
    const float ethr = ((edge + 1) * (edge + 1)) / edge;
    if (good && (__builtin_fabsf(v) > peak) && s < ethr && __builtin_fabsf(dx) < 1 && __builtin_fabsf(dy) < 1 &&
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

// Image rows per unit group (one workgroup, or one quarter of a tail workgroup): ROWS + 2 rows are loaded and absorbed into the
// sliding window for ROWS rows tested, so the halo share falls with ROWS (1.4 at 5, 1.1 at 20) -- and so does the number of
// workgroups. Round 5: launches with thousands of unit groups (the 64-frame calls of the frame driver) take DET_ROWS_TALL:
// octave 0 of 64 frames 1 060 -> 990 us alone, +2.5-3 % on the headline (three alternations, same box:
// 2 936/2 959/2 994 against 3 034/3 033/2 943 at 20 rows and 3 038/3 044/3 027 at 27); a single frame keeps DET_ROWS (with 20
// rows its 432 workgroups no longer fill the chip: 279 -> 582 us per frame). The octave-tail kernel keeps DET_ROWS too.
// (Until round 5 the sub-list scan below took one entry per lane, which capped the rows at 5; 7 rows alone measured 24 %
// SLOWER than 5, 12 rows 8 % slower, 16 equal, 20 7 % faster: profiles/r05_af_detect_rows.txt.)
constexpr int DET_ROWS = 5;
#ifndef NM_DET_ROWS_TALL
#define NM_DET_ROWS_TALL 20
#endif
constexpr int DET_ROWS_TALL = NM_DET_ROWS_TALL;
constexpr int DET_ROWS_TALL2 = 27;       // the second tall height (nm_launch_detect_octave picks per launch)
// A wave tests 62 columns: its lanes 0 and 63 hold the halo columns of the 3 x 3 neighbourhoods and test nothing, so every lane
// loads ONE value per plane and row and the horizontal neighbours come from the adjacent lanes alone. (Until round 5 a wave
// tested 64 columns and every lane loaded a second value -- the left neighbour on lane 0, the right one elsewhere -- of which
// two lanes used theirs: half the load instructions of a kernel that is bound by their number. The launch without those loads,
// results wrong at the wave edges, ran 733 against 972 us; profiles/r05_af_detect_rows.txt.) A unit = one image row of a
// segment = DET_SEG_W pixels; its staging slots stay 256.
constexpr int DET_WAVE_W = NM_DET_WAVE_W, DET_SEG_W = NM_DET_SEG_W;

template <int ROWS>
struct DetectSmemT {
    unsigned char s_x[ROWS][3][4][64];    // candidate lanes per (row, level, wave), in lane order
    int s_cnt[ROWS * 12 + 1];              // sub-list lengths, then their exclusive scan (+ total)
    int s_acc[ROWS * 3], s_last[ROWS * 3], s_wtot[4], s_pref[257];
};
typedef DetectSmemT<DET_ROWS> DetectSmem;

// What one unit group needs of one frame's octave (filled from NmDetectArgs by the launch wrapper, or from the tail
// kernel's per-frame tables).
struct DetectView {
    const float *pl[6];             // LEV: Gaussian levels 0..5, else DoG planes 0..4 (pl[5] unused). Only ever indexed with
                                    // compile-time constants (the refinement SELECTS its planes by the candidate's level): a
                                    // run-time index would send this copy to scratch memory
    float *staging;                 // 3 x stage_stride float4
    size_t stage_stride;
    int *counts;                    // 3 x n_blocks
    float *const *dense;            // DENSE (API path): the three dense maps, likewise a pointer into the kernel arguments
    const float *mask;              // full-resolution mask or NULL
    int mask_w, mask_h;
    int ow, oh;
    float peak, edge, xper, sigma0;
    int num_dogs, n_blocks, nseg;
};

// One unit group (DET_ROWS rows of one DET_SEG_W-pixel segment) of one frame's octave: `blk` = seg + nseg * row group. 256 threads.
// SC1: the survivors and the per-unit counts are consumed by ANOTHER workgroup of the SAME launch (the octave-tail kernel's
// scan + gather item): they are stored write-through (agent-scope atomic stores = global_store ... sc1), see nm_tail.hip.
// NQ > 1 (the tail kernel's 1024-thread workgroups): NQ unit groups run side by side, 256 threads each (quarter q =
// threadIdx.x / 256 works on sm_all[q]); the workgroup barriers inside are shared, so the refinement loop runs for the
// LONGEST candidate list of the NQ groups. !active: the quarter has no unit group (the last item of a segment): it computes on
// a clamped one and stores nothing.
template <bool DENSE, bool LEV, bool MASKED, bool SC1 = false, int NQ = 1, int ROWS = DET_ROWS>
__device__ __forceinline__ void detect_stage_body(const DetectView &a, int blk, DetectSmemT<ROWS> *sm_all, bool active = true)
{
    constexpr int DET_ROWS = ROWS;                          // (shadows the namespace constant inside this function)
    const int tid = (NQ > 1) ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
    DetectSmemT<ROWS> &sm = sm_all[(NQ > 1) ? (threadIdx.x >> 8) : 0];
    auto &s_x = sm.s_x; auto &s_cnt = sm.s_cnt; auto &s_acc = sm.s_acc; auto &s_last = sm.s_last; auto &s_wtot = sm.s_wtot;
    auto &s_pref = sm.s_pref;
    const float *pl[6];                                    // LEV: the six Gaussian levels, else the five DoG planes
#pragma unroll
    for (int p = 0; p < 6; ++p) pl[p] = a.pl[p];
    const int seg = blk % a.nseg, yg = blk / a.nseg;
    const int y0 = yg * DET_ROWS;
    const int lane = tid & 63, wave = tid >> 6;
    const int x = seg * DET_SEG_W + wave * DET_WAVE_W + lane - 1;      // lanes 0 and 63: halo columns
    const int ow = a.ow, oh = a.oh;
    const bool xin = lane >= 1 && lane <= DET_WAVE_W && x < ow;        // the lane tests a pixel of the plane
    const int xc = min(max(x, 0), ow - 1);                             // clamped column for safe addressing
    const float thr = 0.8f * a.peak;
    const float *const mask = MASKED ? a.mask : nullptr;

    // sliding 3-row window per plane: row maxima/minima of (left, mid, right); centre row keeps mid and max/min(l, r).
    // Rows are FETCHED one iteration ahead of being absorbed into the window (raw values wait in registers), so the
    // global-load latency of row y+2 hides behind the tests of row y instead of stalling every iteration.
    float rmax[5][3], rmin[5][3], cmid[5][2], clr_max[5][2], clr_min[5][2];
    float raw_mid[2][5];
    // The addresses are a uniform row base + a 32-bit lane offset (one scalar-base load form each; the 64-bit per-lane pointer
    // arithmetic of `row[xc]` was 84 VALU instructions per workgroup).
    const unsigned xb = 4u * (unsigned)xc;
    // (Measured and not kept: raw buffer loads -- plane descriptor, row as the scalar offset -- remove the last address
    // instructions too and ran 270 against 263 us; loading BOTH neighbour columns instead of shifting by DPP ran 371 us.)
    auto fetch_row = [&](int yy, int buf) {
        const int yr = min(max(yy, 0), oh - 1);
        const size_t rofs = (size_t)yr * (size_t)ow * 4;      // uniform
        if (LEV) {                              // DoG p = level p + 1 - level p, formed here instead of read
            float lm[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) lm[p] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pl[p]) + rofs + xb);
#pragma unroll
            for (int p = 0; p < 5; ++p) raw_mid[buf][p] = lm[p + 1] - lm[p];
            return;
        }
#pragma unroll
        for (int p = 0; p < 5; ++p) raw_mid[buf][p] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pl[p]) + rofs + xb);
    };
    auto absorb_row = [&](int buf, int slot, int cslot) {
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const float mid = raw_mid[buf][p];
            const float lf = dpp_from_lower(mid, mid), rt = dpp_from_upper(mid, mid);      // (lanes 0 / 63: unused)
            rmax[p][slot] = max3f(lf, mid, rt);
            rmin[p][slot] = min3f(lf, mid, rt);
            cmid[p][cslot] = mid;
            clr_max[p][cslot] = __builtin_fmaxf(lf, rt);
            clr_min[p][cslot] = __builtin_fminf(lf, rt);
        }
    };
    // rows y0-1, y0 absorbed, y0+1 in flight before the loop; iteration j absorbs y+1 and fetches y+2
    // (Round 5, measured: TWO rows in flight -- a ring of three raw buffers -- is 19 % slower alone, 20-row groups: 1 166 against
    // 977 us per 64-frame launch, and within the noise on the headline. Like loading both neighbour columns, more loads in
    // flight cost more than they hide.)
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
    // exclusive scan of the DET_ROWS x 12 sub-list lengths -> flattened candidate order (row, level, column): s_cnt[q] = first
    // index. A lane of wave 0 takes EPL consecutive entries (serial prefix inside, wave scan of the lane totals).
    if (wave == 0) {
        constexpr int NSUB = DET_ROWS * 12, EPL = (NSUB + 63) / 64;
        int c[EPL], sum = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const int q = lane * EPL + e; c[e] = q < NSUB ? s_cnt[q] : 0; sum += c[e]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        int run = incl - sum;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const int q = lane * EPL + e; if (q < NSUB) s_cnt[q] = run; run += c[e]; }
        if (lane == 63) s_cnt[NSUB] = incl;
    }
    if (tid < DET_ROWS * 3) s_acc[tid] = 0;
    __syncthreads();
    const int total = s_cnt[DET_ROWS * 12];
    int total_all = total;                                  // barriers inside the loop: every quarter runs the longest list
    if (NQ > 1) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) total_all = max(total_all, sm_all[q].s_cnt[DET_ROWS * 12]);
    }
    for (int b0 = 0; b0 < total_all; b0 += 256) {           // 256 candidates per pass (a 5 x 248 pixel group usually holds < 100)
        const int c = b0 + tid;
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
            px = seg * DET_SEG_W + w * DET_WAVE_W + s_x[jr][lvl][w][c - s_cnt[lo]] - 1;
            py = y0 + jr;
            // the candidate's planes: selected from the copies above (lvl is 0, 1 or 2) -- a table lookup with a run-time
            // index would be one more dependent memory access at the head of the refinement's chain
            auto plane = [&](int k) { return lvl == 0 ? pl[k] : (lvl == 1 ? pl[k + 1] : pl[k + 2]); };
            if (LEV)
                acc = refine_at(LevelPlanes{plane(0), plane(1), plane(2), plane(3)}, px, py, ow, a.peak, a.edge, a.xper,
                                a.sigma0, a.num_dogs, lvl, kp);
            else
                acc = refine(plane(1), plane(0), plane(2), px, py, ow, a.peak, a.edge, a.xper, a.sigma0, a.num_dogs, lvl, kp);
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
        s_pref[tid] = before;                     // accepted candidates of this pass before candidate c
        if (tid == 255) s_pref[256] = before + (acc ? 1 : 0);
        __syncthreads();
        if (c < total) {
            const int gs = s_cnt[g * 4], ge = s_cnt[g * 4 + 4];       // the group's flattened range (s_cnt[48] = total)
            const int base = max(gs, b0) - b0;
            const int pos = s_acc[g] + before - s_pref[base];
            if (acc) {
                const int unit = py * a.nseg + seg;
                float4 *st = reinterpret_cast<float4 *>(a.staging) + (size_t)lvl * a.stage_stride + (size_t)unit * 256;
                if (!active) {}
                else if (SC1) nmdev::store_f4_agent(st + pos, kp);
                else st[pos] = kp;
            }
            if (c == min(ge, b0 + 256) - 1) s_last[g] = pos + (acc ? 1 : 0);     // accepted so far, this pass included
        }
        __syncthreads();
        if (tid < DET_ROWS * 3) {
            const int gs = s_cnt[tid * 4], ge = s_cnt[tid * 4 + 4];
            if (ge > b0 && gs < b0 + 256 && ge > gs) s_acc[tid] = s_last[tid];
        }
        __syncthreads();
    }
    if (!DENSE && tid < DET_ROWS * 3) {           // per-unit counts of the three levels
        const int jr = tid / 3, lv = tid - 3 * jr;
        const int y = y0 + jr;
        if (y < oh && active) {
            int *dst = a.counts + lv * a.n_blocks + y * a.nseg + seg;
            if (SC1) nmdev::store_i32_agent(dst, s_acc[tid]); else *dst = s_acc[tid];
        }
    }
}

}  // namespace nmdet
