// nm_describe.hip -- keypoint orientation histograms and 128-D descriptors for gfx950, one wavefront per keypoint.
// Replaces kernels/orientation.cu:11-129,219-230 and kernels/descriptor.cu:32-145,243-255.
//
// The reference accumulates its histograms with shared/global float atomicAdd, whose order is undefined. Here the
// order is fixed (DESIGN.md "fp spec") and implemented without atomics:
//   orientation: the clipped window (<= 21 x 21) is cut into 63 strips (column rx, row third ry/7), one per lane; a
//                lane sums its <= 7 votes per bin in increasing cy into a private LDS histogram; lane b then adds
//                the 63 partials of bin b in strip order.
//   descriptor : the samples of column tx (0..15) of a 16x16 chunk vote into partial histogram tx in increasing cy
//                (4 rows per wave pass, issued as 4 exec-masked read-add-write rounds in row order: LDS executes a
//                wave's instructions in order), 8 votes per sample in (dbinx, dbiny, dbint) order; a partial histogram
//                has NINE temporal slots per cell (slot 8 = the votes that wrap round to orientation bin 0), so that the two
//                temporal votes of a sample are one two-word LDS access; the 16 partials of every slot are combined by a
//                balanced pairwise tree (strides 1,2,4,8) by the lane that owns the bin, bin 0 = tree(slot 0) + tree(slot 8).
#include <cstdlib>
#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "nm_describe.hpp"
#include "../../include/nm_abi.h"

using nmfp::fma32;
using nmfp::fma64;

namespace {

constexpr int ORI_MAXW = 10;                    // 22x22 block of the reference -> W <= 10 (orientation.cu:29-30)
constexpr int ORI_PITCH = 65;                   // floats per bin row: 64 strip partials + 1 (bank skew for the column sum)
constexpr int ORI_LDS = 36 * ORI_PITCH;         // floats of LDS per wave (9.4 KB)

// One wave computes the orientation(s) of one keypoint in two steps, so that a wave can request the gradient samples of
// its NEXT keypoint before it processes the current one (the gather's latency was most of the kernel's time: 64 % of the
// wave cycles parked, profiles/r02_b_describe_detect_pmc_counters.txt).
struct OriSamples {
    float x, y;
    int xi, yi, W, xmin, xmax, ymin, ymax;
    float denom;
    float2 gv[7];               // this lane's strip: column rx = lane % 21, rows 7 rg .. 7 rg + 6 (rg = lane / 21)
    bool valid;
};

__device__ __forceinline__ void orient_fetch(const float4 kp, const float2 *__restrict__ grad, int ow, int oh,
                                             float gauss_factor, float xper, OriSamples &o)
{
    const int lane = threadIdx.x & 63;
    o.valid = !(kp.w < 0);
    if (!o.valid) return;
    const float x = kp.x / xper, y = kp.y / xper, s = kp.z / xper;
    const int xi = (int)((double)x + 0.5), yi = (int)((double)y + 0.5);
    const float sigma_w = gauss_factor * s;
    int W = max((int)__builtin_floorf(3 * sigma_w), 1);
    W = min(ORI_MAXW, W);
    o.x = x; o.y = y; o.xi = xi; o.yi = yi; o.W = W;
    o.xmin = max(-W, -xi); o.xmax = min(W, ow - 1 - xi);
    o.ymin = max(-W, -yi); o.ymax = min(W, oh - 1 - yi);
    o.denom = (2 * sigma_w) * sigma_w;
    const float2 *g = grad + (((long)kp.w * oh + yi) * (long)ow + xi);
    const int rg = lane / 21, rx = lane - 21 * rg;
    const int cx = o.xmin + rx;
    const bool col_ok = (lane < 63) && (cx <= o.xmax);
#pragma unroll
    for (int j = 0; j < 7; ++j) {                // all 7 gathers in flight together
        const int cy = o.ymin + 7 * rg + j;
        o.gv[j] = make_float2(0.f, 0.f);
        if (col_ok && cy <= o.ymax) o.gv[j] = g[(long)cy * ow + cx];
    }
}

// part: ORI_LDS floats private to the wave, [bin][strip]. First half of a keypoint: zero the partial sums, cast the votes (the
// only part that reads the gathered samples).
// FAST (a wave-uniform choice of the caller, orient_votes below): the vote's arithmetic with the keypoint's part of it hoisted --
//   * `(double)r2 < r2lim` as the binary32 compare r2 < C, C the smallest binary32 >= r2lim (equivalent for every r2, NaN too);
//   * r2 / denom by nmfp::div_by (5 instructions instead of the 11 of the IEEE expansion), its domain checked per keypoint:
//     denom in [2^-20, 2^20], and r2 = 0 or >= 2^-100 because x, y are 0 or at least 2^-50 in magnitude (dx = (float)n - x is
//     then 0 or at least 2^-50: for n != 0 it is 0 or no smaller than half an ulp of n);
//   * the quotient lies in [0, 87.1] (r2 < C <= 87 denom), so expf_spec's two clamps are dropped;
//   * a row's (float)(cy + yi) as fy0 + j (|y| < 2^22: both sides exact);
//   * bin = b mod 36 as a select when every lane's b lies in [0, 36] (orientations in [0, 2 pi] as the gradient kernel writes
//     them; any other value takes `%`).
__device__ __forceinline__ float orient_r2cap(double r2lim)      // r2lim > 0
{
    const float c = (float)r2lim;
    return (double)c < r2lim ? __uint_as_float(__float_as_uint(c) + 1u) : c;      // the next binary32 up
}

template <bool FAST>
__device__ __forceinline__ void orient_votes_as(const OriSamples &o, float *part)
{
    const int lane = threadIdx.x & 63;
    const float x = o.x, y = o.y;
    const int xi = o.xi, yi = o.yi, W = o.W;
    const int xmin = o.xmin, xmax = o.xmax, ymin = o.ymin, ymax = o.ymax;
    const float denom = o.denom;
    const double r2lim = (double)(W * W) + 0.6;
    const float r2cap = orient_r2cap(r2lim);
    const nmfp::DivBy by = nmfp::div_by(denom);

    {   // zero the wave's 36 x 65 partial sums with 16-byte stores (585 float4: 9 rounds of 64 lanes + 9)
        static_assert(ORI_LDS % 4 == 0 && ORI_LDS / 4 == 9 * 64 + 9, "zeroing pattern");
        float4 *z = reinterpret_cast<float4 *>(part);
#pragma unroll
        for (int i = 0; i < 9; ++i) z[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 9) z[9 * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int rg = lane / 21, rx = lane - 21 * rg;
    const int cx = xmin + rx;
    const bool col_ok = (lane < 63) && (cx <= xmax);
    float *mine = part + lane;
    const int cy0 = ymin + 7 * rg, rows_left = ymax - cy0;
    const float dx = (float)(cx + xi) - x;
    const float fy0 = (float)(cy0 + yi);             // FAST: |y| < 2^22, so (float)(cy0 + yi + j) = fy0 + j, both exact
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const float dy = (FAST ? fy0 + (float)j : (float)(cy0 + j + yi)) - y;
        const float r2 = fma32(dx, dx, dy * dy);
        if (col_ok && j <= rows_left && (FAST ? r2 < r2cap : (double)r2 < r2lim)) {
            // Only floor(q) of q = (float)((double)(36 theta) / 2 pi) is used. A binary32 estimate q' = (36 theta) * (1 / 2 pi)
            // is within 1e-5 of q (q <= 36), so floor(q') = floor(q) whenever q' is at least 1e-4 away from an integer; the
            // few samples inside that band take the exact expression.
            const float t36 = 36.0f * o.gv[j].y;
            const float qe = t36 * 0.15915494309189535f;
            float fq = __builtin_floorf(qe);
            const float dq = qe - fq;
            if (__builtin_expect(!(dq > 1e-4f && dq < 0.9999f), 0))
                fq = __builtin_floorf(nmfp::div_to_f32((double)t36, nmfp::TWO_PI_D, nmfp::INV_TWO_PI_D));
            const int b = (int)fq;
            int bin;
            if (FAST && !__builtin_expect(__any((unsigned)b > 36u), 0)) bin = (int)min((unsigned)b, (unsigned)b - 36u);   // b - 36 wraps unless b = 36
            else bin = b % 36;
            float *const word = FAST ? mine + __mul24(bin, ORI_PITCH) : mine + bin * ORI_PITCH;
            const float sum = *word;                    // lane-private word: plain read-add-write, program order
            const float wgt = FAST ? nmfp::expf_spec<0>(nmfp::div_by(by, r2)) : nmfp::expf_spec(r2 / denom);
            *word = sum + o.gv[j].x * wgt;
        }
    }
    __builtin_amdgcn_wave_barrier();             // same-wave LDS traffic is in order; this only pins the compiler
}

__device__ __forceinline__ void orient_votes(const OriSamples &o, float *part)
{
    if (!o.valid) return;
    const float ax = __builtin_fabsf(o.x), ay = __builtin_fabsf(o.y);
    const bool fast = nmfp::div_by_domain(o.denom) && (ax == 0.f || ax >= 0x1p-50f) && (ay == 0.f || ay >= 0x1p-50f) &&
                      ay < 0x1p22f && orient_r2cap((double)(o.W * o.W) + 0.6) <= 87.0f * o.denom;
    if (__builtin_amdgcn_readfirstlane((int)fast)) orient_votes_as<true>(o, part);       // a keypoint's values are wave-uniform
    else orient_votes_as<false>(o, part);
}

// Lane exchanges of the second half as DPP operands of the instructions that consume them (no LDS traffic, no waits): the wave
// rotations by one lane of GFX9 for the circular neighbours, the row patterns for the maximum.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int l)      // l wave-uniform
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
constexpr int DPP_WAVE_ROL1 = 0x134, DPP_WAVE_ROR1 = 0x13C;       // lane i <- lane i + 1 / lane i - 1, modulo 64
constexpr int DPP_QUAD_X1 = 0xB1, DPP_QUAD_X2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

// Second half: sums, smoothing, peaks. Returns the number of peaks found (0..2); th0/th1 are valid in every lane.
// Lane b < 36 owns bin b. The six smoothing rounds and the peak test read the two circular neighbours of a bin; so that these are
// the neighbouring LANES, lanes 36..42 carry copies of bins 0..6 and lanes 57..63 copies of bins 29..35 (lane 63 -> lane 0 is the
// rotation's wrap): a copy computes what its bin's owner computes, from the same operands in the same order, and stays right as
// long as both its neighbours are -- the run of right lanes 57..63, 0..42 loses one lane at either end per round, and after six
// rounds lanes 63, 0..36 are left, which is what the peak test of lanes 0..35 reads. Lanes 43..56 hold junk that nothing reads.
__device__ __forceinline__ int orient_peaks(const OriSamples &o, float &th0, float &th1, const float *part)
{
    const int lane = threadIdx.x & 63;
    th0 = -1.f; th1 = -1.f;
    if (!o.valid) return 0;

    const int hb = lane < 36 ? lane : lane < 43 ? lane - 36 : lane >= 57 ? lane - 28 : -1;
    float h = 0.f;                                // partials in strip order
    // (reads 9 at a time. Round 6, measured: all 63 reads issued before the first add -- 106 registers, the launch has 128 -- is
    // 1 % SLOWER, five alternations: profiles/r06_zz_orient_sum63.txt)
    if (hb >= 0) {                                // h = ((row[0] + row[1]) + row[2]) + ... + row[62]
        const float *row = part + hb * ORI_PITCH;
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            float v[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) v[i] = row[9 * g + i];
#pragma unroll
            for (int i = 0; i < 9; ++i) h = (g == 0 && i == 0) ? v[0] : h + v[i];
        }
    }
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int iter = 0; iter < 6; ++iter) {        // race-free circular 3-tap mean (orientation.cu:181-192)
        const float s3 = (dpp_f32<DPP_WAVE_ROR1>(h) + h) + dpp_f32<DPP_WAVE_ROL1>(h);      // nh == (float)((double)s3 / 3.0)
        h = __builtin_expect(__all(nmfp::third_f32_exact(s3)), 1) ? nmfp::third_f32(s3) : nmfp::div_to_f32((double)s3, 3.0, 1.0 / 3.0);
    }
    const float hm = dpp_f32<DPP_WAVE_ROR1>(h), hp = dpp_f32<DPP_WAVE_ROL1>(h);
    float m = (lane < 36) ? h : 0.f;              // the maximum of 64 values, in any order
    m = __builtin_fmaxf(m, dpp_f32<DPP_QUAD_X1>(m));
    m = __builtin_fmaxf(m, dpp_f32<DPP_QUAD_X2>(m));
    m = __builtin_fmaxf(m, dpp_f32<DPP_HALF_MIRROR>(m));
    m = __builtin_fmaxf(m, dpp_f32<DPP_ROW_MIRROR>(m));          // every lane: the maximum of its row of 16
    m = __builtin_fmaxf(__builtin_fmaxf(lane_value(m, 0), lane_value(m, 16)), lane_value(m, 32));   // row 3 is all zeros, as is half of row 2
    const float threshold = (float)((double)m * 0.8);
    const bool peak = (lane < 36) && (h > threshold) && (h > hm) && (h > hp);
    unsigned long long mask = __ballot(peak);
    const float di = (float)((-0.5 * (double)(hp - hm)) / (double)((hp + hm) - 2 * h));
    const float th = nmfp::div_to_f32(nmfp::TWO_PI_D * ((double)((float)lane + di) + 0.5), 36.0, 1.0 / 36.0);
    int npk = 0;
    if (mask) { th0 = lane_value(th, __ffsll((long long)mask) - 1); mask &= mask - 1; npk = 1; }
    if (mask) { th1 = lane_value(th, __ffsll((long long)mask) - 1); npk = 2; }
    return npk;
}

__device__ __forceinline__ int orient_wave(const float4 kp, const float2 *__restrict__ grad, int ow, int oh,
                                           float gauss_factor, float xper, float &th0, float &th1, float *part)
{
    OriSamples o;
    orient_fetch(kp, grad, ow, oh, gauss_factor, xper, o);
    orient_votes(o, part);
    return orient_peaks(o, th0, th1, part);
}

// LDS of one descriptor wave (floats): [16 cells][9 temporal slots][16 partials] | [9 rows][16] landing words of rejected votes.
// The kernel is bound by the NUMBER of LDS instructions its votes issue (round 4: halving the waves per SIMD -> 1.7 x the time;
// 10 % fewer VALU instructions, conflict-free banks, the votes' latency hidden behind the next pass's math -> no gain or a
// loss; no votes at all -> -27 %; LDS float atomics -> 6 x slower), so the layout serves the votes: a sample's two temporal
// votes go to slots (bint & 7) and (bint & 7) + 1 of the same cell and partial -- 16 floats apart, ONE ds_read2_b32 and ONE
// ds_write2_b32 (round 6: ONE ds_add_f32 per vote with all 64 lanes -- the hardware applies the lanes that share a word in
// ascending lane order with IEEE adds, denormals included, so the order would have been specifiable -- runs at ~770 cycles per
// wave-instruction: the kernel 301.8 instead of 46.4 us per frame; tools/micro/lds_atomic.hip, profiles/r06_b_*) -- with slot 8
// collecting what wraps round to orientation bin 0 (added to it when the partials are combined:
// the fp spec's summation order). With a pitch of 16 floats the 16 lanes of a vote instruction (partials
// tx = 0..15 of arbitrary rows) hit 16 different banks.
constexpr int DESC_PITCH = 16;
constexpr int DESC_ROWS = 16 * 9;               // 144
constexpr int DESC_DUMMY = DESC_ROWS * DESC_PITCH;             // + slot * DESC_PITCH + tx
constexpr int DESC_LDS = DESC_DUMMY + 9 * DESC_PITCH;          // 2448 floats = 9 792 B: 16 waves per CU
static_assert(DESC_LDS * 4 <= 10240, "descriptor LDS layout");
// One wave computes one descriptor. part: DESC_LDS floats of LDS private to the wave, laid out [bin][partial].
// A descriptor is computed in two steps so that a wave can set up its NEXT keypoint (window, and the gradient samples of
// the first 16 x 16 chunk) before it processes the current one: the gather latency at the start of a keypoint was exposed.
struct DescSetup {
    bool valid;
    float kx, ky, x, y, SBP, angle0;
    int xi, yi, xmin, xmax, ymin, ymax, chunks, ow;
    const float2 *gptr;
    float2 first[4];            // chunk 0: rows tyg, tyg + 4, tyg + 8, tyg + 12 of column tx
};

__device__ __forceinline__ void desc_fetch_chunk(const DescSetup &d, int c, float2 (&dst)[4])
{
    const int lane = threadIdx.x & 63, tx = lane & 15, tyg = lane >> 4;
    // 32-bit element offsets relative to the keypoint's pixel (a level plane has < 2^31 elements): chunk c starts at
    // (xmin + 16 c, ymin + 16 c), a lane's q-th sample lies 4 q rows further down
    const int off0 = (tyg + d.ymin) * d.ow + (tx + d.xmin) + c * 16 * (d.ow + 1), row4 = 4 * d.ow;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cx = tx + d.xmin + 16 * c, cy = 4 * q + tyg + d.ymin + 16 * c;
        dst[q] = make_float2(0.f, 0.f);
        if (cx <= d.xmax && cy <= d.ymax) dst[q] = d.gptr[off0 + q * row4];
    }
}

__device__ __forceinline__ void desc_setup(const float4 kp, const float angle0, const float2 *__restrict__ grad, int ow,
                                           int oh, int num_dogs, float xper, DescSetup &d)
{
    const float x = kp.x / xper, y = kp.y / xper, s = kp.z / xper;
    const int xi = (int)((double)x + 0.5), yi = (int)((double)y + 0.5), si = (int)kp.w;
    d.valid = !(xi < 0 || xi >= ow || yi < 0 || yi >= oh || si < 0 || si >= num_dogs);
    if (!d.valid) return;
    const float SBP = (float)((double)(3 * s) + 1.e-07);
    const int W = (int)__builtin_floor(1.41421356237309514547 * (double)SBP * 5 / 2.0 + 0.5);   // sqrt(2.0) in double
    d.kx = kp.x; d.ky = kp.y; d.x = x; d.y = y; d.SBP = SBP; d.angle0 = angle0; d.xi = xi; d.yi = yi; d.ow = ow;
    d.xmin = max(-W, -xi); d.xmax = min(W, ow - 1 - xi);
    d.ymin = max(-W, -yi); d.ymax = min(W, oh - 1 - yi);
    const int max_dims = max(d.xmax - d.xmin, d.ymax - d.ymin);
    d.chunks = (int)__builtin_ceilf((max_dims + 1.f) / 16);
    d.gptr = grad + (((long)si * oh + yi) * (long)ow + xi);
    if (d.chunks > 0) desc_fetch_chunk(d, 0, d.first);
}

// One wave computes one descriptor. part: DESC_LDS floats of LDS private to the wave, laid out [bin][partial].
__device__ __forceinline__ void desc_run(const DescSetup &d, float *__restrict__ desc, float *__restrict__ xp,
                                         float *__restrict__ yp, float *part)
{
    if (!d.valid) return;
    const int lane = threadIdx.x & 63;
    const float x = d.x, y = d.y, SBP = d.SBP, angle0 = d.angle0;
    const int xi = d.xi, yi = d.yi, xmin = d.xmin, xmax = d.xmax, ymin = d.ymin, ymax = d.ymax, chunks = d.chunks;
    if (lane == 0) { *xp = d.kx; *yp = d.ky; }
    const double st0 = (double)nmfp::sinf_spec(angle0), ct0 = (double)nmfp::cosf_spec(angle0);
    const double dSBP = (double)SBP, rSBP = 1.0 / dSBP;
    const float fct = (float)ct0, fst = (float)st0, frs = 1.0f / SBP;
    const int tx = lane & 15, tyg = lane >> 4;

    {   // zero the wave's partial histograms with 16-byte stores
        float4 *z = reinterpret_cast<float4 *>(part);
        // (the landing rows behind the histogram only ever take +0 and are never read: they are not zeroed)
        static_assert(DESC_DUMMY / 4 == 9 * 64, "zeroing pattern");
#pragma unroll
        for (int i = 0; i < 9; ++i) z[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float *mine = part + 90 * DESC_PITCH + tx;    // origin at the centre cell (descriptor.cu:81: cell (2, 2) = row 9 * 10), partial tx
    // Per-lane constants of the sample grid. A lane's samples are (column tx of chunk c, row 4 q + tyg of chunk c):
    //   cx = xmin + tx + 16 c,  cy = ymin + tyg + 16 c + 4 q;  in the window <=> 16 c <= lx and 16 c + 4 q <= ly.
    // fx0 + 16 c and fy0 + (16 c + 4 q) are exact (integers below 2^24): the sample's pixel coordinate (float)(xi + cx).
    const float fx0 = (float)(xi + xmin + tx), fy0 = (float)(yi + ymin + tyg);
    const int lx = xmax - xmin - tx, ly = ymax - ymin - tyg;
    // binary32 ESTIMATE of the normalised coordinates (nx, ny), linear in (c, q): only used to skip passes, with a margin
    // (0.01) a thousand times its error
    const float ex0 = (fct * (fx0 - x) + fst * (fy0 - y)) * frs, ey0 = (fct * (fy0 - y) - fst * (fx0 - x)) * frs;
    const float sxc = 16.f * (fct + fst) * frs, syc = 16.f * (fct - fst) * frs, sxq = 4.f * fst * frs, syq = 4.f * fct * frs;

    // Gradient samples of a chunk are fetched together (4 independent loads per lane) and the next chunk's loads are
    // issued before the current chunk is processed, so the gather latency is paid once, not per sample.
    float2 cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = d.first[q];

    for (int c = 0; c < chunks; ++c) {
        if (c + 1 < chunks) desc_fetch_chunk(d, c + 1, nxt);
        // Round 6: what depends on the sample's COLUMN only is formed once per chunk. ct0, st0 are floats widened to binary64 and
        // dx, dy are floats, so the products ct0 dx, st0 dx (and st0 dy, ct0 dy below) are EXACT in binary64 (48 significant
        // bits): fma64(ct0, dx, st0 dy) = RN64(ct0 dx + st0 dy) is then the plain sum of the two exact products, and
        // fma64(-st0, dx, ct0 dy) = RN64(ct0 dy - st0 dx) their difference -- the same roundings of the same real numbers.
        const float dx = (fx0 + (float)(16 * c)) - x;
        const double pxc = ct0 * (double)dx, pxs = st0 * (double)dx;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool inwin = (16 * c <= lx) && (16 * c + 4 * q <= ly);
            {   // a sample votes only if |nx| < 2.5 and |ny| < 2.5 (bins -2..1 in x and y). Skip the whole 4 x 16 pass
                // when no lane can: the skipped votes would all have been exact +0 no-ops. fp32 estimate, 0.01 margin.
                const float exc = fma32((float)c, sxc, ex0), eyc = fma32((float)c, syc, ey0);
                const float enx = q ? fma32((float)q, sxq, exc) : exc, eny = q ? fma32((float)q, syq, eyc) : eyc;
                const bool maybe = inwin && __builtin_fabsf(enx) < 2.51f && __builtin_fabsf(eny) < 2.51f;
                if (!__any(maybe)) continue;
            }
            const float2 gq = cur[q];
            const float mod = inwin ? gq.x : 0.f, ang = gq.y;
            const float theta = nmfp::mod_2pi_f(ang - angle0);
            const float dy = (fy0 + (float)(16 * c + 4 * q)) - y;
            const double ddy = (double)dy;
            float nx, ny, nt;
            nmfp::div3_to_f32_wave(pxc + st0 * ddy, ct0 * ddy - pxs, dSBP, rSBP, (double)(8.0f * theta), nmfp::TWO_PI_D,
                                   nmfp::INV_TWO_PI_D, nx, ny, nt);
            // exp_spec clamps its argument to [-700, 700]: t / 8 > 700 <=> t > 5600 exactly (t = nx^2 + ny^2 >= 0 is a float,
            // the division by 8 is exact), so the clamp is taken on the float (one v_min_f32) and the binary64 compares /
            // selects are dropped
            // (On a sample that can vote, |nx|, |ny| <= 2.5 and t <= 12.5: the fast form of nm_fpspec.hpp, bit-identical to the
            // spec sequence unless it reports `near`. Beyond EXPW_TMAX the clamped value is meaningless and never used: every
            // vote of such a sample is rejected by the in-grid test below.)
            const float t = fma32(nx, nx, ny * ny);
            bool near;
            float win = nmfp::expw_fast(__builtin_fminf(t, nmfp::EXPW_TMAX), near);
            if (__builtin_expect(__any(near), 0)) win = (float)nmfp::exp_spec_in_range((double)__builtin_fminf(t, 5600.0f) / 8.0);
            // floor((double)n - 0.5) and (float)((double)n - (bin + 0.5)) in binary32, bit for bit: (double)n - 0.5 is exact, so
            // the floor is floor(n) - [n - floor(n) < 0.5] (both exact in binary32); bin + 0.5 is exact in binary32, and the
            // one rounding of n - (bin + 0.5) is the same rounding of the same real number (|n| < 2^22 inside a window)
            const float fx = __builtin_floorf(nx), fy = __builtin_floorf(ny);
            const int binx = (int)fx - ((nx - fx) < 0.5f ? 1 : 0);
            const int biny = (int)fy - ((ny - fy) < 0.5f ? 1 : 0);
            const int bint = (int)__builtin_floorf(nt);
            const float rbinx = nx - ((float)binx + 0.5f);
            const float rbiny = ny - ((float)biny + 0.5f);
            const float rbint = nt - (float)bint;
            const float wm = win * mod;
            // votes outside the 4x4 grid (or outside the window) become +0 into a landing word behind the histogram, so the 8
            // addresses of a sample never collide and a round can be issued as 4 two-word loads, 8 adds, 4 two-word stores
            const int tw = (bint & 7) * DESC_PITCH;                  // bint in [0, 8]; the second vote lies DESC_PITCH further
            float *const dummy = part + DESC_DUMMY + tx + tw;
            // word = (9 * cell + slot) * DESC_PITCH + tx, cell = (binx + 2) + 4 (biny + 2), with 24-bit multiplies
            float *const base = mine + (__mul24(binx, 9 * DESC_PITCH) + __mul24(biny, 36 * DESC_PITCH)) + tw;
            const bool okx[2] = {(unsigned)(binx + 2) < 4u, (unsigned)(binx + 3) < 4u};
            const bool oky[2] = {(unsigned)(biny + 2) < 4u, (unsigned)(biny + 3) < 4u};
            // The in-grid test depends on (dbx, dby) only: it is applied to the partial product (wm * ax) * ay and to the
            // (binx, biny) part of the address -- four selects each instead of eight. A rejected vote is +0 either way
            // (0 * |..| = +0: the factors are finite and non-negative).
            float wt[8];
            float *loc[4];
            const float at0 = __builtin_fabsf(1.f - rbint), at1 = __builtin_fabsf(0.f - rbint);
#pragma unroll
            for (int dbx = 0; dbx < 2; ++dbx)
#pragma unroll
                for (int dby = 0; dby < 2; ++dby) {
                    const int c4 = dbx * 2 + dby;
                    const bool ok = inwin && okx[dbx] && oky[dby];
                    const float w2 = wm * __builtin_fabsf((1.f - dbx) - rbinx) * __builtin_fabsf((1.f - dby) - rbiny);
                    const float w2s = ok ? w2 : 0.f;
                    loc[c4] = ok ? base + (dbx * 9 + dby * 36) * DESC_PITCH : dummy;
                    wt[2 * c4] = w2s * at0;
                    wt[2 * c4 + 1] = w2s * at1;
                }
            // rounds none of whose 16 samples has a vote inside the grid are skipped (their votes are all +0 into the landing words)
            const unsigned long long voters = __ballot(inwin && (okx[0] || okx[1]) && (oky[0] || oky[1]));
#pragma unroll
            for (int k = 0; k < 4; ++k) {          // rows of this pass in increasing cy: 16 lanes per round, LDS in order
                if (!((voters >> (16 * k)) & 0xFFFFull)) continue;
                if (tyg == k) {
                    float o[8];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) { o[2 * c4] = loc[c4][0]; o[2 * c4 + 1] = loc[c4][DESC_PITCH]; }
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) { loc[c4][0] = o[2 * c4] + wt[2 * c4]; loc[c4][DESC_PITCH] = o[2 * c4 + 1] + wt[2 * c4 + 1]; }
                }
                // The four rounds are mutually exclusive per THREAD, so the compiler may merge or reorder them; their
                // order only matters across lanes (same word, different rows). A compiler-level memory fence between
                // the rounds pins the program order that the in-order LDS then executes.
                asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    }
    __builtin_amdgcn_wave_barrier();

    // lane b owns descriptor elements b and b + 64 (element = 8 cell + t): pairwise tree over the 16 partials (strides 1, 2,
    // 4, 8) of its slot row, all in registers; orientation bin 0 also takes the tree of slot 8 (the wrapped votes)
    auto tree = [&](int row) {
        const float4 *r4 = reinterpret_cast<const float4 *>(part + row * DESC_PITCH);
        const float4 a = r4[0], b = r4[1], c = r4[2], d = r4[3];
        const float s01 = a.x + a.y, s23 = a.z + a.w, s45 = b.x + b.y, s67 = b.z + b.w;
        const float s89 = c.x + c.y, sab = c.z + c.w, scd = d.x + d.y, sef = d.z + d.w;
        const float t0 = s01 + s23, t1 = s45 + s67, t2 = s89 + sab, t3 = scd + sef;
        return (t0 + t1) + (t2 + t3);
    };
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int e = lane + 64 * hb, cell = e >> 3, t = e & 7;
        float v = tree(9 * cell + t);
        if (t == 0) v = v + tree(9 * cell + 8);
        desc[e] = v;
    }
}

__device__ __forceinline__ void describe_wave(const float4 kp, const float angle0, const float2 *__restrict__ grad,
                                              int ow, int oh, int num_dogs, float xper, float *__restrict__ desc,
                                              float *__restrict__ xp, float *__restrict__ yp, float *part)
{
    DescSetup d;
    desc_setup(kp, angle0, grad, ow, oh, num_dogs, xper, d);
    desc_run(d, desc, xp, yp, part);
}

// ---- API kernels (one octave, one level list per launch) ----
__global__ __launch_bounds__(256) void orientations_kernel(const float4 *__restrict__ key_pts,
                                                          const float2 *__restrict__ grad, int num_pts, int ow, int oh,
                                                          float gauss_factor, float xper, float2 *__restrict__ result)
{
    __shared__ __attribute__((aligned(16))) float s_part[4][ORI_LDS];
    const int wave = threadIdx.x >> 6;
    for (int pt = blockIdx.x * 4 + wave; pt < num_pts; pt += gridDim.x * 4) {
        float th0, th1;
        const int npk = orient_wave(key_pts[pt], grad, ow, oh, gauss_factor, xper, th0, th1, s_part[wave]);
        if ((threadIdx.x & 63) == 0) {            // only found peaks are written (orientation.cu:117-128)
            if (npk >= 1) result[pt].x = th0;
            if (npk >= 2) result[pt].y = th1;
        }
    }
}

__global__ __launch_bounds__(64) void descriptors_kernel(const float4 *__restrict__ key_pts,
                                                        const float2 *__restrict__ orients,
                                                        const float2 *__restrict__ grad, int num_pts, int ow, int oh,
                                                        int num_dogs, float xper, float *__restrict__ desc,
                                                        float *__restrict__ xp, float *__restrict__ yp)
{
    __shared__ __attribute__((aligned(16))) float part[DESC_LDS];
    for (int pt = blockIdx.x; pt < num_pts; pt += gridDim.x)
        describe_wave(key_pts[pt], orients[pt].x, grad, ow, oh, num_dogs, xper, desc + (size_t)pt * 128, xp + pt,
                      yp + pt, part);
}

// ---- API kernels, up to 3 level lists of one octave per launch (compute_orientations / compute_descriptors) ----
struct NmLevelLists {
    const float4 *key_pts[3];
    float2 *orients[3];
    float *desc[3], *x[3], *y[3];
    int num_pts[3];
    int n_levels;
    // Device-sized form (round 5; lazy_count.h): d_counts != NULL -- the three raw level counts live on the device (what
    // nm_compact_keypoints3 left), num_pts[] is ignored and the grid is sized for an upper bound. An empty level ends the octave
    // (sift/siftfunctions.cu:145,160). Descriptors: output slot of level l's keypoint pt = base + (kept keypoints of the levels
    // before it) + pt, base = *d_base_in (or host_base when NULL), clipped at `capacity` (siftfunctions.cu:165-169); desc[0] /
    // x[0] / y[0] are then the container's arrays from slot 0. One lane mirrors the counts / the new running count into
    // mapped host words.
    const int *d_counts;
    const int *d_base_in; int host_base, capacity;
    int *d_items_out;
    int *h_counts, *h_items;           // mapped pinned host words (or NULL)
};

// kept keypoints (n) and first output slot (base) of level l in the device-sized form
__device__ __forceinline__ void level_extent_dev(const NmLevelLists &a, int l, bool clip, int &n, int &base, int &run_out)
{
    int run = clip ? (a.d_base_in ? *a.d_base_in : a.host_base) : 0;
    bool live = true;
    n = 0; base = run;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int cnt = live ? a.d_counts[k] : 0;
        if (cnt <= 0) { live = false; cnt = 0; }
        int keep = cnt;
        if (clip) { if (keep + run > a.capacity) keep = a.capacity - run; if (keep < 0) keep = 0; }
        if (k == l) { n = keep; base = run; }
        run += clip ? keep : 0;
    }
    run_out = run;
}

// Both kernels walk the three level lists as ONE list (index = level-major): a grid row per level left the workgroups of the
// short lists idle while those of the longest ran ~3 rounds (octave 0 of a 1080p frame: 4.5k / 2.2k / 0.8k keypoints).
__device__ __forceinline__ void level_counts(const NmLevelLists &a, bool clip, int (&n)[3], int (&base)[3], int &run)
{
    if (a.d_counts) {
        level_extent_dev(a, 0, clip, n[0], base[0], run);
        level_extent_dev(a, 1, clip, n[1], base[1], run);
        level_extent_dev(a, 2, clip, n[2], base[2], run);
    } else {
#pragma unroll
        for (int l = 0; l < 3; ++l) { n[l] = l < a.n_levels ? max(a.num_pts[l], 0) : 0; base[l] = 0; }
        run = 0;
    }
}

__global__ __launch_bounds__(256) void orientations_levels_kernel(NmLevelLists a, const float2 *__restrict__ grad, int ow,
                                                                 int oh, float gauss_factor, float xper)
{
    __shared__ __attribute__((aligned(16))) float s_part[4][ORI_LDS];
    const int wave = threadIdx.x >> 6;
    int n[3], base[3], run;
    level_counts(a, false, n, base, run);
    if (a.d_counts && a.h_counts && blockIdx.x == 0 && threadIdx.x < 3) a.h_counts[threadIdx.x] = a.d_counts[threadIdx.x];
    const int total = n[0] + n[1] + n[2];
    for (int idx = blockIdx.x * 4 + wave; idx < total; idx += gridDim.x * 4) {
        const int l = idx < n[0] ? 0 : (idx < n[0] + n[1] ? 1 : 2);
        const int pt = idx - (l > 0 ? n[0] : 0) - (l > 1 ? n[1] : 0);
        const float4 *kp = l == 0 ? a.key_pts[0] : (l == 1 ? a.key_pts[1] : a.key_pts[2]);
        float2 *out = l == 0 ? a.orients[0] : (l == 1 ? a.orients[1] : a.orients[2]);
        float th0, th1;                           // unset components are -1, as the pre-fill of pyramidata.cu:90 leaves them
        orient_wave(kp[pt], grad, ow, oh, gauss_factor, xper, th0, th1, s_part[wave]);
        if ((threadIdx.x & 63) == 0) out[pt] = make_float2(th0, th1);
    }
}

__global__ __launch_bounds__(64) void descriptors_levels_kernel(NmLevelLists a, const float2 *__restrict__ grad, int ow,
                                                               int oh, int num_dogs, float xper)
{
    __shared__ __attribute__((aligned(16))) float part[DESC_LDS];
    int n[3], base[3], run;
    level_counts(a, true, n, base, run);
    if (a.d_counts && blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.d_items_out) *a.d_items_out = run;
        if (a.h_items) *a.h_items = run;
    }
    const int total = n[0] + n[1] + n[2];
    for (int idx = blockIdx.x; idx < total; idx += gridDim.x) {
        const int l = idx < n[0] ? 0 : (idx < n[0] + n[1] ? 1 : 2);
        const int pt = idx - (l > 0 ? n[0] : 0) - (l > 1 ? n[1] : 0);
        const float4 *kp = l == 0 ? a.key_pts[0] : (l == 1 ? a.key_pts[1] : a.key_pts[2]);
        const float2 *ori = l == 0 ? a.orients[0] : (l == 1 ? a.orients[1] : a.orients[2]);
        float *desc, *xs, *ys;
        if (a.d_counts) {                         // the container's arrays from slot 0
            const int slot = (l == 0 ? base[0] : (l == 1 ? base[1] : base[2])) + pt;
            desc = a.desc[0] + (size_t)slot * 128; xs = a.x[0] + slot; ys = a.y[0] + slot;
        } else {
            desc = (l == 0 ? a.desc[0] : (l == 1 ? a.desc[1] : a.desc[2])) + (size_t)pt * 128;
            xs = (l == 0 ? a.x[0] : (l == 1 ? a.x[1] : a.x[2])) + pt;
            ys = (l == 0 ? a.y[0] : (l == 1 ? a.y[1] : a.y[2])) + pt;
        }
        describe_wave(kp[pt], ori[pt].x, grad, ow, oh, num_dogs, xper, desc, xs, ys, part);
    }
}

// ---- frame-driver kernels: all octaves of a frame in one launch, counts read from the device-side book ----
__device__ __forceinline__ int octave_of(const NmFrameBook *book, int num_octaves, int i)
{
    int o = 0;
    while (o + 1 < num_octaves && i >= book->oct_base[o + 1]) ++o;
    return o;
}

// (Round 6, measured and removed: the keypoints dealt to the XCDs by LOCALITY instead of round robin. Consecutive keypoints are
// neighbours in (octave, level, y, x) and their gradient windows overlap; dealt round robin, every window line is fetched by all
// eight L2s: 92 % of the orientation kernel's L2 requests and 79 % of the descriptor kernel's miss (142 MB per frame for 66 MB of
// planes). Giving XCD x the x-th eighth of every (octave, level) list -- a raster band per level, the same mix of scales on every
// XCD -- raised the hit rates to 42 % / 59 % and made both kernels SLOWER (descriptors 46.9 against 44.9 us per frame, orientation
// 18.7 against 17.4, headline -2 %; one contiguous eighth of the whole list: 52 against 46, the XCDs with the large scales carry
// 25 % more work). The misses are not what these kernels wait for; the bands' uneven keypoint density is worse than the traffic.
// profiles/r06_m_gather_traffic.txt, r06_n_desc_xcd_locality.txt)
__global__ __launch_bounds__(256) void frame_orient_kernel(NmDescribeArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_part[4][ORI_LDS];
    const int wave = threadIdx.x >> 6;
    const int frame = blockIdx.y;
    const NmFrameBook *book = a.book[frame];
    const float4 *kpts = reinterpret_cast<const float4 *>(a.kpts[frame]);
    float2 *orients = reinterpret_cast<float2 *>(a.orients[frame]);
    const int first = book->oct_base[a.o_begin], n_end = book->oct_base[a.o_end];   // output slots [first, n_end) of octaves [o_begin, o_end)
    const int vstride = gridDim.x * 4;
    int v = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    auto pick = [&](int vv) { return first + vv < n_end ? first + vv : -1; };
    OriSamples cur, nxt;
    auto fetch = [&](int p, const float4 kp, OriSamples &o) {
        const int oc = octave_of(book, a.num_octaves, p);
        orient_fetch(kp, reinterpret_cast<const float2 *>(a.grad0[frame] + a.grad_off[oc]), a.geom[oc].ow, a.geom[oc].oh, 1.5f,
                     a.geom[oc].xper, o);
    };
    // (Round 6: the loop carries the keypoint INDICES of the next two iterations instead of re-deriving their bounds from pt: the
    // same requests in the same order, 18.0 against 18.8 us per frame, alternating on one box.)
    // Two loads deep: the samples of keypoint i + 1 are requested from the keypoint record loaded during keypoint i - 1, behind
    // the VOTES of keypoint i (the only part that reads samples) and in front of its sums, smoothing and peaks. Round 5 stamps
    // of the loop as it was (next keypoint's record loaded and its samples requested at the TOP of an iteration): 23 % of a
    // wave's time per keypoint in that fetch -- the record's latency, exposed, before the dependent gathers could issue --
    // and a third in the votes, whose first use of the current samples waits with vmcnt(0), i.e. for the gathers issued a
    // few hundred cycles earlier as well (`profiles/r05_ak_orient_stamps.txt`).
    const float4 none = make_float4(0.f, 0.f, 0.f, -1.f);           // w < 0: orient_fetch / votes / peaks do nothing
    float4 kp_next = none, kp_after = none;
    int pt = pick(v), pt_next = pick(v + vstride), pt_after = -1;
    if (pt >= 0) fetch(pt, kpts[pt], cur);
    if (pt_next >= 0) kp_next = kpts[pt_next];
    for (; pt >= 0; v += vstride) {
        const bool more = pt_next >= 0;
        orient_votes(cur, s_part[wave]);
        if (more) fetch(pt_next, kp_next, nxt);                         // (kp_next landed an iteration ago: nothing newer is in flight)
        pt_after = more ? pick(v + 2 * vstride) : -1;
        if (pt_after >= 0) kp_after = kpts[pt_after];
        float th0, th1;                           // unset components stay -1 (pyramidata.cu:90)
        orient_peaks(cur, th0, th1, s_part[wave]);
        if ((threadIdx.x & 63) == 0) orients[pt] = make_float2(th0, th1);
        if (more) { cur = nxt; kp_next = kp_after; }
        pt = pt_next; pt_next = pt_after;
    }
}

// (Round 6, measured and removed: chunk PAIRS. The samples of partial tx in chunks c and c + 1 at the same row of their chunk lie
// exactly (16, 16) pixels apart = (16 (ct0 + st0), 16 (ct0 - st0)) / SBP cells, more than 2 cells in one component when SBP <= 7.95,
// so they never vote into the same word and a pass of 2 rows x 2 chunks needs TWO rounds of 32 lanes instead of four of 16: half
// the vote rounds' LDS instructions and exec-masked adds. 46.8 against 46.9 us per frame (the 2 x 2-row passes are skipped less
// often than the 4-row passes of one chunk, 128 VGPRs), and NOT the same sums: rows 0.. of chunk c + 1 are then added before rows
// ..15 of chunk c, which share words with them. profiles/r06_g_desc_pairs_ab.txt)
// (Round 6, measured and removed: the weights and the votes' adds as packed fp32 -- (wm ax) ay and * at as v_pk_mul_f32, a round's
// eight adds as four v_pk_add_f32 on the ds_read2 pairs; |1 - r| = 1 - r and |0 - r| = r hold bit for bit on r in [0, 1] -- 165 -> 141
// vector instructions per pass, bit-identical, and 3 % SLOWER (47.3 against 45.8 us per frame, alternating): an exec-masked
// v_add_f32 with 16 live lanes costs less than its share of a packed instruction. profiles/r06_k_desc_packed_ab.txt. Taking that
// at its word -- a half wave (two of a pass's four rows) none of whose samples can vote leaves the pass through EXEC -- is
// bit-identical too and 2 % slower: 46.8 against 45.9 us per frame, five alternations.)
// (Setting up keypoint pt + stride -- window and first chunk of samples -- before computing keypoint pt was measured:
// 983 vs 930 us per 16 frames. This kernel is bound by VALU issue, not by the gather latency, and the second setup costs
// 12 VGPRs. The straightforward loop stays.)
__global__ __launch_bounds__(64) void frame_desc_kernel(NmDescribeArgs a)
{
    __shared__ __attribute__((aligned(16))) float part[DESC_LDS];
    const int frame = blockIdx.y;
    const NmFrameBook *book = a.book[frame];
    const int n = book->oct_base[a.o_end];
    const float4 *kpts = reinterpret_cast<const float4 *>(a.kpts[frame]);
    const float2 *orients = reinterpret_cast<const float2 *>(a.orients[frame]);
    for (int pt = book->oct_base[a.o_begin] + blockIdx.x; pt < n; pt += gridDim.x) {
        const int o = octave_of(book, a.num_octaves, pt);
        describe_wave(kpts[pt], orients[pt].x, reinterpret_cast<const float2 *>(a.grad0[frame] + a.grad_off[o]), a.geom[o].ow,
                      a.geom[o].oh, a.num_dogs, a.geom[o].xper, a.desc[frame] + (size_t)pt * 128, a.x[frame] + pt,
                      a.y[frame] + pt, part);
    }
}

// Exhaustive self-test of nmfp::expw_fast: every float t in [0, EXPW_TMAX] against the spec sequence.
// out[0] = inputs whose fast value differs from the spec's although `near` was not reported (must be 0), out[1] = inputs that
// report `near` (they take the spec sequence in the kernel; ~2^-15 of all), out[2] = inputs tested.
__global__ __launch_bounds__(256) void selftest_expw_kernel(unsigned long long *out)
{
    const uint32_t last = __float_as_uint(nmfp::EXPW_TMAX);
    unsigned long long bad = 0, nears = 0, n = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x) {
        const float t = __uint_as_float((uint32_t)b);
        bool near;
        const float f = nmfp::expw_fast(t, near);
        const float e = (float)nmfp::exp_spec_in_range((double)t / 8.0);
        ++n;
        if (near) ++nears;
        else if (__float_as_uint(f) != __float_as_uint(e)) ++bad;
    }
    for (int d = 32; d >= 1; d >>= 1) { bad += __shfl_xor(bad, d); nears += __shfl_xor(nears, d); n += __shfl_xor(n, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], nears); atomicAdd(&out[2], n); }
}

// Self-test of the orientation kernel's hoisted arithmetic against the expressions it replaces (tests/test_gpu_stages.py):
// out[0] = binary32 x (ALL 2^32) with third_f32_exact(x) and third_f32(x) != (float)((double)x / 3.0)          (must be 0)
// out[1] = binary32 x that third_f32_exact rejects (2^24 + 1: the NaNs, the infinities, -0)
// out[2] = (r2, W) with (r2 < orient_r2cap(W W + 0.6)) != ((double)r2 < W W + 0.6), all 2^32 r2 x W = 1..10  (must be 0)
// out[3] = pseudo-random (num, den) of div_by's domain with div_by(den)(num) != num / den                     (must be 0)
// out[4] = pairs tested for out[3]
__global__ __launch_bounds__(256) void selftest_orient_kernel(unsigned long long *out)
{
    unsigned long long bad3 = 0, rej = 0, badc = 0, badd = 0, nd = 0;
    float cap[10];
    for (int W = 1; W <= 10; ++W) cap[W - 1] = orient_r2cap((double)(W * W) + 0.6);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += nth) {
        const float x = __uint_as_float((uint32_t)b);
        if (!nmfp::third_f32_exact(x)) ++rej;
        else if (__float_as_uint(nmfp::third_f32(x)) != __float_as_uint((float)((double)x / 3.0))) ++bad3;
#pragma unroll
        for (int W = 1; W <= 10; ++W)
            if ((x < cap[W - 1]) != ((double)x < (double)(W * W) + 0.6)) ++badc;
    }
    unsigned long long st = 0x9E3779B97F4A7C15ull * (tid + 1);
    for (int it = 0; it < 2048; ++it) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned a = (unsigned)(st >> 32), c = (unsigned)st;
        const int ed = (int)((a >> 23) % 41u) - 20;                                    // den in [2^-20, 2^21): clipped to the domain below
        float den = __uint_as_float(((unsigned)(ed + 127) << 23) | (a & 0x7FFFFFu));
        if (den > 0x1p20f) den = 0x1p20f;
        const int en = (int)((c >> 23) % 108u) - 100;                                  // num in [2^-100, 2^7), one in 256 exactly 0
        const float num = (c >> 24) == 0u ? 0.0f : __uint_as_float(((unsigned)(en + 127) << 23) | (c & 0x7FFFFFu));
        ++nd;
        if (__float_as_uint(nmfp::div_by(nmfp::div_by(den), num)) != __float_as_uint(num / den)) ++badd;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        bad3 += __shfl_xor(bad3, d); rej += __shfl_xor(rej, d); badc += __shfl_xor(badc, d); badd += __shfl_xor(badd, d); nd += __shfl_xor(nd, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], bad3); atomicAdd(&out[1], rej); atomicAdd(&out[2], badc); atomicAdd(&out[3], badd); atomicAdd(&out[4], nd);
    }
}

}  // namespace

// workgroups (of 4 keypoint-waves) per frame: few enough that a wave walks several keypoints and the prefetch pays
constexpr int NM_ORIENT_BLOCKS = 512;
constexpr int NM_DESC_BLOCKS = 4096;      // one keypoint-wave each

int nm_launch_frame_describe(const NmDescribeArgs &a, hipStream_t stream)
{
    if (a.n <= 0 || a.o_end <= a.o_begin) return 0;
    // the small octaves hold a few hundred keypoints at most: a grid sized for octave 0 would be thousands of empty workgroups
    int ob = a.o_begin >= 2 ? 64 : NM_ORIENT_BLOCKS, db = a.o_begin >= 2 ? 512 : NM_DESC_BLOCKS;
    // (tuning hooks: workgroups per frame of the two launches; a smaller grid leaves wave slots and LDS of every CU to the
    // other streams' scale-space and detection launches)
    static const int env_ob = [] { const char *e = getenv("NM_ORIENT_BLOCKS"); return e ? atoi(e) : 0; }();
    static const int env_db = [] { const char *e = getenv("NM_DESC_BLOCKS"); return e ? atoi(e) : 0; }();
    if (env_ob > 0 && a.o_begin < 2) ob = max(1, env_ob / a.n);
    if (env_db > 0 && a.o_begin < 2) db = max(1, env_db / a.n);
    nm_prof_begin(NM_PROF_ORIENT, stream);
    hipLaunchKernelGGL(frame_orient_kernel, dim3(ob, a.n), dim3(256), 0, stream, a);
    nm_prof_end(NM_PROF_ORIENT, stream);
    NM_LAUNCH_CHECK();
    nm_prof_begin(NM_PROF_DESCRIBE, stream);
    hipLaunchKernelGGL(frame_desc_kernel, dim3(db, a.n), dim3(64), 0, stream, a);
    nm_prof_end(NM_PROF_DESCRIBE, stream);
    NM_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int nm_selftest_expw(unsigned long long *d_out, void *stream)
{
    if (!d_out) return (int)hipErrorInvalidValue;
    NM_RETURN_IF(hipMemsetAsync(d_out, 0, 3 * sizeof(unsigned long long), nm_stream(stream)));
    hipLaunchKernelGGL(selftest_expw_kernel, dim3(8192), dim3(256), 0, nm_stream(stream), d_out);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_selftest_orient(unsigned long long *d_out, void *stream)
{
    if (!d_out) return (int)hipErrorInvalidValue;
    NM_RETURN_IF(hipMemsetAsync(d_out, 0, 5 * sizeof(unsigned long long), nm_stream(stream)));
    hipLaunchKernelGGL(selftest_orient_kernel, dim3(8192), dim3(256), 0, nm_stream(stream), d_out);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_detect_orientations(const float *key_pts, const float *grad, int num_pts, int octave_width, int octave_height,
                           float gauss_factor, float xper, float *result, void *stream)
{
    if (num_pts <= 0) return 0;
    const int blocks = min(nm_divup(num_pts, 4), 4096);
    hipLaunchKernelGGL(orientations_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<const float4 *>(key_pts), reinterpret_cast<const float2 *>(grad), num_pts,
                       octave_width, octave_height, gauss_factor, xper, reinterpret_cast<float2 *>(result));
    NM_LAUNCH_CHECK();
    return 0;
}

// detect_orientations for up to three level lists of one octave in one launch. Unlike nm_detect_orientations, both
// components of every result are written (unset = -1), so the caller need not pre-fill (pyramidata.cu:90).
int nm_detect_orientations_levels(int n_levels, const float *const *key_pts, const int *num_pts, const float *grad,
                                  int octave_width, int octave_height, float gauss_factor, float xper,
                                  float *const *result, void *stream)
{
    if (n_levels <= 0) return 0;
    if (n_levels > 3 || !key_pts || !num_pts || !result) return (int)hipErrorInvalidValue;
    NmLevelLists a{};
    int most = 0;
    a.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        a.key_pts[l] = reinterpret_cast<const float4 *>(key_pts[l]);
        a.orients[l] = reinterpret_cast<float2 *>(result[l]);
        a.num_pts[l] = num_pts[l];
        most = max(most, num_pts[l]);
    }
    if (most <= 0) return 0;
    int sum = 0;
    for (int l = 0; l < n_levels; ++l) sum += num_pts[l] > 0 ? num_pts[l] : 0;
    hipLaunchKernelGGL(orientations_levels_kernel, dim3(min(nm_divup(sum, 4), 4096)), dim3(256), 0,
                       nm_stream(stream), a, reinterpret_cast<const float2 *>(grad), octave_width, octave_height,
                       gauss_factor, xper);
    NM_LAUNCH_CHECK();
    return 0;
}

// The device-sized forms (lazy_count.h): the three level counts are read on the device, nothing comes back to the host except
// through the mapped words h_counts[3] / h_items[1] (device-accessible pointers of pinned host memory, or NULL).
// max_pts: upper bound of a level's count (sizes the grid; the lists hold at least that many entries).
int nm_detect_orientations_levels_dev(const float *const *key_pts, const int *d_counts, int max_pts, const float *grad,
                                      int octave_width, int octave_height, float gauss_factor, float xper, float *const *result,
                                      int *h_counts, void *stream)
{
    if (!key_pts || !d_counts || !result || max_pts <= 0) return (int)hipErrorInvalidValue;
    NmLevelLists a{};
    a.n_levels = 3;
    for (int l = 0; l < 3; ++l) {
        a.key_pts[l] = reinterpret_cast<const float4 *>(key_pts[l]);
        a.orients[l] = reinterpret_cast<float2 *>(result[l]);
    }
    a.d_counts = d_counts; a.h_counts = h_counts;
    hipLaunchKernelGGL(orientations_levels_kernel, dim3(min(nm_divup(max_pts, 4), 1024)), dim3(256), 0, nm_stream(stream), a,
                       reinterpret_cast<const float2 *>(grad), octave_width, octave_height, gauss_factor, xper);
    NM_LAUNCH_CHECK();
    return 0;
}

// desc / x / y: the CONTAINER's arrays (slot 0); the running item count is read from d_base_in (NULL: host_base) and the new
// one written to d_items_out and *h_items.
int nm_compute_sift_descriptors_levels_dev(const float *const *key_pts, const float *const *orients, const int *d_counts,
                                           int max_pts, const int *d_base_in, int host_base, int capacity, int *d_items_out,
                                           int *h_items, const float *grad, int octave_width, int octave_height, int num_dogs,
                                           float xper, float *desc, float *x, float *y, void *stream)
{
    if (!key_pts || !orients || !d_counts || !desc || !x || !y || max_pts <= 0 || capacity <= 0) return (int)hipErrorInvalidValue;
    NmLevelLists a{};
    a.n_levels = 3;
    for (int l = 0; l < 3; ++l) {
        a.key_pts[l] = reinterpret_cast<const float4 *>(key_pts[l]);
        a.orients[l] = const_cast<float2 *>(reinterpret_cast<const float2 *>(orients[l]));
    }
    a.desc[0] = desc; a.x[0] = x; a.y[0] = y;
    a.d_counts = d_counts; a.d_base_in = d_base_in; a.host_base = host_base; a.capacity = capacity;
    a.d_items_out = d_items_out; a.h_items = h_items;
    hipLaunchKernelGGL(descriptors_levels_kernel, dim3(min(min(max_pts, capacity), 4096)), dim3(64), 0, nm_stream(stream), a,
                       reinterpret_cast<const float2 *>(grad), octave_width, octave_height, num_dogs, xper);
    NM_LAUNCH_CHECK();
    return 0;
}

// compute_sift_descriptors for up to three level lists of one octave in one launch.
int nm_compute_sift_descriptors_levels(int n_levels, const float *const *key_pts, const float *const *orients,
                                       const int *num_pts, const float *grad, int octave_width, int octave_height,
                                       int num_dogs, float xper, float *const *desc, float *const *x, float *const *y,
                                       void *stream)
{
    if (n_levels <= 0) return 0;
    if (n_levels > 3 || !key_pts || !orients || !num_pts || !desc || !x || !y) return (int)hipErrorInvalidValue;
    NmLevelLists a{};
    int most = 0;
    a.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        a.key_pts[l] = reinterpret_cast<const float4 *>(key_pts[l]);
        a.orients[l] = const_cast<float2 *>(reinterpret_cast<const float2 *>(orients[l]));
        a.desc[l] = desc[l]; a.x[l] = x[l]; a.y[l] = y[l];
        a.num_pts[l] = num_pts[l];
        most = max(most, num_pts[l]);
    }
    if (most <= 0) return 0;
    int sum = 0;
    for (int l = 0; l < n_levels; ++l) sum += num_pts[l] > 0 ? num_pts[l] : 0;
    hipLaunchKernelGGL(descriptors_levels_kernel, dim3(min(sum, 4096)), dim3(64), 0, nm_stream(stream), a,
                       reinterpret_cast<const float2 *>(grad), octave_width, octave_height, num_dogs, xper);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_compute_sift_descriptors(const float *key_pts, const float *orients, const float *grad, int num_pts,
                                int octave_width, int octave_height, int num_dogs, float xper, float *desc, float *x,
                                float *y, void *stream)
{
    if (num_pts <= 0) return 0;
    const int blocks = min(num_pts, 4096);
    hipLaunchKernelGGL(descriptors_kernel, dim3(blocks), dim3(64), 0, nm_stream(stream),
                       reinterpret_cast<const float4 *>(key_pts), reinterpret_cast<const float2 *>(orients),
                       reinterpret_cast<const float2 *>(grad), num_pts, octave_width, octave_height, num_dogs, xper, desc,
                       x, y);
    NM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
