// nm_match.hip -- brute-force 128-D squared-L2 matcher for gfx950.
// Replaces kernels/transpose.cu:9-40, kernels/match.cu:14-150 and sift/siftfunctions.cu:15-40.
//
// Fused path (nm_sift_match_f32 / nm_sift_match_shard_f32), never materialising the nA x nB matrix:
//   1. norms_kernel        ||a||^2, ||b||^2
//   2. match_top2_kernel   fp32 MFMA (v_mfma_f32_32x32x2_f32): d~[j][i] = nb_j + na_i - 2 a_i.b_j as ONE accumulation
//                          chain (K = 128 products + 1 augmented k-pair carrying the norms); every lane keeps a running
//                          best/second-best/third-best per query over the candidates it sees (integer keys, see Top3);
//                          256 persistent workgroups over balanced unit ranges, a partial list per segment (MatchPlan).
//   3. match_finalize_kernel  per query: pick the 4 best partial candidates, recompute their distances EXACTLY in the
//                          reference's order (sum_k fma(t,t,acc), t = a_k - b_k, k ascending: match.cu:36-42), then apply
//                          the scan semantics of match.cu:91-116 (lowest index wins ties, min2 initial 2139095040.0f,
//                          result untouched when min2 <= 0). Each segment also reports the VALUE of its third best; when
//                          anything not recomputed comes within the error margin of the exact min2, the query goes to
//   4. match_fallback_kernel (+ match_fallback_merge_kernel), which scan all candidates exactly (a fraction of a
//                          percent of the queries), 64 candidate slices per listed query.
// API building blocks (transpose / bf_distance / get_sift_matches) keep the reference's layouts and are exact.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "nm_common.hpp"
#include "../../include/nm_abi.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DIM = 128;
constexpr int KP = 132;            // LDS row pitch (floats): 128 data + norm slot + pad; 132 mod 64 = 4 -> b128 reads conflict-free
constexpr int TILE_C = 128;        // candidates per LDS tile
constexpr int QB = 256;            // queries per workgroup (32 per wave, fragments resident in VGPRs)
constexpr float MIN2_INIT = 2139095040.0f;   // (float)0x7f800000, match.cu:91
constexpr int MAX_CHUNKS = 64;     // upper bound of the number S of segments (partial lists) per query block of any plan
// Domain of the MFMA screens: descriptors whose squared norms are finite and below NORM_LIMIT (every distance is then finite,
// <= 4e37). A query row outside it, or ANY candidate outside it, is matched by the exact fallback alone -- the reference's
// scan, whose behaviour on NaN / inf distances (match.cu:91-116: comparisons with a NaN are false) is reproduced there.
constexpr float NORM_LIMIT = 1.0e37f;
// The fp16 coarse pass of the two-stage screen needs |2 x| <= 65504 for every element: squared norms below 1e9 (|x| < 31623).
constexpr float F16_NORM_LIMIT = 1.0e9f;

// Work = qblocks x T units, a unit being (256 queries) x (one 128-candidate tile). The grid is G persistent workgroups
// (one per CU: the kernel owns the LDS), each taking a contiguous range of `base` or `base+1` units of a linear order:
// every CU carries the same MFMA load to within one tile and the chip drains at once. A range is processed as SEGMENTS:
// maximal runs of consecutive tiles of one query block (the query fragments are reloaded per segment). Segment k of a
// query block writes partial slot k; the segment that finishes the block also blanks the slots up to S.
//
// The linear order is XCD-aware (round 2). Workgroups b and b + X share an XCD and its 4 MB L2 (X = 8 on MI355X: blocks
// are dealt round-robin over the XCDs; speed only, never correctness). Group x = the workgroups {x, x + X, ...} gets a
// contiguous share of the query blocks, and orders its units PIECE-major: the candidate tiles are cut into C chunks of
// Tc tiles (Tc ~ the units per workgroup), and the order runs chunk by chunk, inside a chunk query block by query block.
// The ~nq workgroups that work on one chunk at the same time then stream the SAME candidate tiles, so the XCD's L2
// serves them once instead of every query block re-streaming the candidate set from the Infinity Cache. With X = 1 and
// C = 1 this is the plain query-block-major order (used for small problems and single-XCD partitions).
struct MatchPlan { int qblocks, T, G, S, X, Gx, Tc, C, q_base, q_rem; };
// Unit indices: qblocks x T <= 2^14 x 2^15 (the entries refuse sets of 2^22 rows or more), so 32 bits hold them. (They were
// 64-bit until round 3: the ~20 integer divisions a workgroup makes per segment to find its way through the plan were
// expanded to 64-bit software division on the vector unit -- ~7 us per segment of the coarse pass.)
typedef int unit_t;
struct PlanGroup { int nq, q0, base, rem; };

#define NM_HD __host__ __device__ __forceinline__
// x / d for 0 <= x, 0 < d. On the device an integer division is ~40 vector instructions, and a workgroup makes a few dozen of
// them per segment on values that are the same for all its lanes: below 2^20 (305 k units at 100k x 100k) the quotient comes
// from one v_rcp_f32 and a remainder check instead (the estimate is off by at most one there: 2^20 x 2^-22 relative error
// + the truncation).
NM_HD int pdiv(int x, int d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (x < (1 << 20)) {
        int q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
        int r = x - q * d;
        if (r < 0) { q -= 1; r += d; }
        if (r >= d) q += 1;
        return q;
    }
#endif
    return x / d;
}
NM_HD PlanGroup plan_group(const MatchPlan &p, int x)
{
    PlanGroup g;
    g.nq = p.q_base + (x < p.q_rem ? 1 : 0);
    g.q0 = x * p.q_base + (x < p.q_rem ? x : p.q_rem);
    const unit_t U = (unit_t)g.nq * p.T;
    g.base = pdiv(U, p.Gx);
    g.rem = U - g.base * p.Gx;
    return g;
}
NM_HD unit_t group_begin(const PlanGroup &g, int v) { return (unit_t)v * g.base + (v < g.rem ? v : g.rem); }
NM_HD int group_owner(const PlanGroup &g, unit_t ul)            // local workgroup whose range holds local unit ul
{
    const unit_t cut = (unit_t)g.rem * (g.base + 1);
    if (ul < cut) return pdiv(ul, g.base + 1);            // (one division, not both: the values are uniform)
    return g.rem + pdiv(ul - cut, g.base);
}
// local unit ul -> chunk c, local query block qbl, tile offset tt inside the piece, piece length Lc
NM_HD void plan_locate(const MatchPlan &p, const PlanGroup &g, unit_t ul, int &c, int &qbl, int &tt, int &Lc)
{
    const unit_t per_chunk = (unit_t)g.nq * p.Tc;
    c = pdiv(ul, per_chunk);
    const int r = (int)(ul - (unit_t)c * per_chunk);
    Lc = p.T - c * p.Tc < p.Tc ? p.T - c * p.Tc : p.Tc;
    qbl = pdiv(r, Lc);
    tt = r - qbl * Lc;
}
// workgroups (of the group) that work on piece (c, qbl)
NM_HD void piece_owners(const MatchPlan &p, const PlanGroup &g, int c, int qbl, int &first, int &last)
{
    const int Lc = p.T - c * p.Tc < p.Tc ? p.T - c * p.Tc : p.Tc;
    const unit_t P = (unit_t)g.nq * c * p.Tc + (unit_t)qbl * Lc;
    first = group_owner(g, P);
    last = group_owner(g, P + Lc - 1);
}
// partial-list slot of the segment that local workgroup v owns in piece (c, qbl)
NM_HD int plan_slot(const MatchPlan &p, const PlanGroup &g, int c, int qbl, int v)
{
    int slot = 0, f, l;
    for (int cc = 0; cc < c; ++cc) { piece_owners(p, g, cc, qbl, f, l); slot += l - f + 1; }
    piece_owners(p, g, c, qbl, f, l);
    return slot + (v - f);
}

// A workgroup's segments in PROCESSING order. Its range is contiguous in the piece-major order, so only its first segment
// can start in the middle of a piece (the tail another workgroup left). If more pieces of the same chunk follow, that
// tail is processed AFTER them: every workgroup then walks a chunk's tiles in ascending tile index from the chunk's first
// tile, in step with the other workgroups of its XCD on the same chunk -- the first reader of a candidate tile misses in
// L2, the others hit. (Slots and results do not depend on the processing order.)
struct SegIter {
    unit_t u, u_end, u0;
    bool have_def;
    unit_t def_u;
    NM_HD void init(unit_t b, unit_t e) { u = b; u0 = b; u_end = e; have_def = false; def_u = 0; }
    // next segment: local unit where it starts (its length follows from plan_locate); false when done
    NM_HD bool next(const MatchPlan &p, const PlanGroup &g, unit_t &seg_u)
    {
        for (;;) {
            if (u >= u_end) {
                if (have_def) { have_def = false; seg_u = def_u; return true; }
                return false;
            }
            int c, qbl, tt, Lc;
            plan_locate(p, g, u, c, qbl, tt, Lc);
            const unit_t n = (Lc - tt) < (u_end - u) ? (Lc - tt) : (u_end - u);
            if (have_def) {
                int dc, dq, dt, dl;
                plan_locate(p, g, def_u, dc, dq, dt, dl);
                if (dc != c) { have_def = false; seg_u = def_u; return true; }     // chunk changes: flush the tail first
            } else if (u == u0 && tt > 0 && u + n < u_end) {
                int c2, q2, t2, l2;
                plan_locate(p, g, u + n, c2, q2, t2, l2);
                if (c2 == c) { have_def = true; def_u = u; u += n; continue; }      // defer the leading tail
            }
            seg_u = u;
            u += n;
            return true;
        }
    }
};

// Largest number of partial-list slots any query block needs under plan p. The (group, local query block) pairs are dealt
// over `nlanes` callers (host: 1; device: the lanes of one wave, which then take the maximum over the wave).
NM_HD int plan_max_slots_part(const MatchPlan &p, int lane, int nlanes)
{
    int S = 1;
    // query block k belongs to group x = the one whose [q0, q0 + nq) holds it (plan_group), found without a loop so that
    // the lanes of a wave run the same instructions on their own k (a `continue` per foreign k would serialise them)
    const int cut = p.q_rem * (p.q_base + 1);
    for (int k = lane; k < p.qblocks; k += nlanes) {
        const int x = (k < cut) ? k / (p.q_base + 1) : p.q_rem + (k - cut) / p.q_base;
        const PlanGroup g = plan_group(p, x);
        const int qbl = k - g.q0;
        int f, l;
        piece_owners(p, g, p.C - 1, qbl, f, l);
        const int n = plan_slot(p, g, p.C - 1, qbl, l) + 1;
        if (n > S) S = n;
    }
    return S;
}

NM_HD int hd_divup(int a, int b) { return (a + b - 1) / b; }
// Largest descriptor-set size (exclusive) the matcher accepts: the SRDs address rows with 32-bit byte offsets, and the work
// plan's unit indices are 32-bit (unit_t): qblocks x T = (2^22 / 256) x (2^22 / 128) = 2^29 units at the limit. Every entry
// that makes a plan rejects larger sets first (hipErrorInvalidValue).
constexpr int MATCH_MAX_ROWS = 1 << 22;

// The plan for (nA, nB) on a device of n_cu compute units in n_xcd XCDs. REDUCE(S) turns a caller's partial maximum into
// the maximum over all callers (identity on the host). Every caller computes the same plan.
template <typename Reduce>
NM_HD MatchPlan make_plan_on(int nA, int nB, int n_cu, int n_xcd, int lane, int nlanes, Reduce reduce)
{
    MatchPlan p;
    p.qblocks = hd_divup(nA > 0 ? nA : 1, QB);
    p.T = hd_divup(nB > 0 ? nB : 1, TILE_C);
    const unit_t U = (unit_t)p.qblocks * p.T;
    // XCD-grouped order: the whole chip is used, the query blocks split over the XCDs to within 3 %, and every XCD has a
    // few query blocks to share tiles between
    if (n_xcd > 1 && n_cu % n_xcd == 0 && U >= 4L * n_cu && p.qblocks >= 2 * n_xcd &&
        (unit_t)hd_divup(p.qblocks, n_xcd) * n_xcd * 100 <= (unit_t)p.qblocks * 103) {
        p.G = n_cu; p.X = n_xcd; p.Gx = n_cu / n_xcd;
        p.q_base = p.qblocks / n_xcd; p.q_rem = p.qblocks % n_xcd;
        const unit_t upw = U / n_cu;                                   // units per workgroup
        int C = (int)((p.T + upw / 2) / (upw > 0 ? upw : 1));       // chunks ~ T / units-per-workgroup
        if (C < 1) C = 1;
        if (C > p.T) C = p.T;
        p.Tc = hd_divup(p.T, C);
        p.C = hd_divup(p.T, p.Tc);
        p.S = reduce(plan_max_slots_part(p, lane, nlanes));
        if (p.S <= MAX_CHUNKS) return p;
    }
    // plain query-block-major order over one group
    const int min_len = hd_divup(p.T, MAX_CHUNKS - 2);            // a block spans <= MAX_CHUNKS - 2 whole ranges + 2 ends
    unit_t G = U / min_len;
    if (G > n_cu) G = n_cu;
    if (G < 1) G = 1;
    p.G = (int)G; p.X = 1; p.Gx = p.G;
    p.q_base = p.qblocks; p.q_rem = 0;
    p.Tc = p.T; p.C = 1;
    p.S = reduce(plan_max_slots_part(p, lane, nlanes));
    return p;
}

static MatchPlan make_plan(int nA, int nB, int wg_per_cu = 1)
{
    // one persistent workgroup per CU of the current device (256 on MI355X SPX); two for the coarse pass of the two-stage
    // screen, whose 68 KiB of LDS let two workgroups share a CU (four waves per SIMD instead of two)
    return make_plan_on(nA, nB, nm_cu_count() * wg_per_cu, nm_xcd_count(), 0, 1, [](int s) { return s; });
}

// Everything the small launches around the MFMA kernel need for one (A, B) pair. A batched call (nm_sift_match_batch_f32)
// runs norms / finalize / fallback / merge ONCE for all its pairs (the pair is a grid dimension) and only the MFMA
// kernel once per pair: the ~45 us of dependent-launch gaps and tiny launches per match shrink to a few per call.
struct MatchPair {
    const float *A, *B;
    float *na, *nb;
    float *nbmax;              // max of the candidate norms (one float), for the finalize bound
    float4 *partial;
    float *partial3;
    int *fb_count, *fb_list;
    int *result;
    float *min1, *min2;
    int *idx1;
    unsigned *As, *Bs;         // bf16x3 screen: split images of A (scaled by -2) and B, 512 B per row (hi | lo)
    uint4 *nbslot;             // bf16x3 screen: the candidates' norm k-slots, padded to whole tiles
    // two-stage screen (f16 coarse pass, then bf16x3 on the rows it could not prove): fp16 images of A (scaled by -2) and B,
    // 256 B per row; the 2-norms of their rounding residuals a - a_h, b - b_h (rounded up); the list of unproven rows
    // (its counter, the max of rb and the second work plan live in the pair's 256-byte counter block, see pair_f1_count)
    // and the norms of the listed rows in list order. As above holds the split images of the LISTED rows in this mode.
    unsigned *Ah, *Bh;
    float *ra, *rb, *na2;
    int *f1_list;
    int nA, nB, S, mode, index_offset;
    // Device-sized call (nm_sift_match_batch_dev_f32): the real sizes are read from device memory (what the frame driver
    // left in d_num_items), nA / nB above are the CAPACITIES every grid and the workspace are laid out for, and the work
    // plan is made on the device (nbmax_kernel) into d_plan. NULL for the host-sized entries.
    const int *d_nA, *d_nB;
    MatchPlan *d_plan;
};
// sizes / partial-list stride of a pair as the kernels see them
__device__ __forceinline__ int pair_nA(const MatchPair &c) { return c.d_nA ? min(max(*c.d_nA, 0), c.nA) : c.nA; }
__device__ __forceinline__ int pair_nB(const MatchPair &c) { return c.d_nB ? min(max(*c.d_nB, 0), c.nB) : c.nB; }
__device__ __forceinline__ int pair_S(const MatchPair &c) { return c.d_plan ? c.d_plan->S : c.S; }
// the pair's 256-byte counter block: [0] fallback count, [4] count of the rows the coarse pass left to the bf16x3 pass,
// [16] max candidate norm, [17] max candidate residual norm, [32..41] device-side plan, [44..53] plan of the bf16x3 pass
__device__ __forceinline__ int *pair_f1_count(const MatchPair &c) { return c.fb_count + 4; }
__device__ __forceinline__ MatchPlan *pair_plan2(const MatchPair &c) { return reinterpret_cast<MatchPlan *>(c.fb_count + 44); }
constexpr int MATCH_MAX_BATCH = 16;
struct MatchBatch {
    MatchPair p[MATCH_MAX_BATCH];
    int n;
    float ambiguity;
    float err_coeff;           // |screen value - exact d| <= err_coeff (sqrt na + sqrt nb)^2 for the screen in use
    float err_coeff2;          // two-stage screen: the same for its second (bf16x3) pass; err_coeff then covers the fp32
                               // accumulation of the coarse pass only, the fp16 rounding enters through ra / rb
    int n_cu, n_xcd;           // device-sized calls: the geometry the device-side plan is made for (n_cu = persistent workgroups)
    int n_cu2;                 // two-stage screen: workgroups of the second pass (one per CU), for the plan fine_rows_kernel makes
    int pair_xcd;              // coarse pass (round 6): > 0 = the number of XCDs when every pair of the call belongs to ONE of them
                               // (pair q to XCD q mod pair_xcd: the workgroups wg with wg mod pair_xcd == q mod pair_xcd, local
                               // index wg / pair_xcd, under a one-group plan for n_cu workgroups); 0 = every workgroup on every pair
};
static_assert(sizeof(MatchBatch) <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ int nm_divup_dev(int a, int b) { return (a + b - 1) / b; }

// bf16 pieces. rne: round to nearest even (the guide's integer form; finite inputs). The split x = hi + lo + r has
// |x - hi| <= 2^-8 |x| and |r| <= 2^-16 |x|  (bf16 carries 8 significant bits).
__device__ __forceinline__ unsigned bf16_rne(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split(float x, unsigned &hi, unsigned &lo)
{
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float(hi << 16));          // the difference is exact
}
// A non-negative float as the EXACT sum of three bf16 values (8 + 8 + 8 significant bits, by truncation).
__device__ __forceinline__ void bf16_three(float n, unsigned &h, unsigned &m, unsigned &l)
{
    const unsigned uh = __float_as_uint(n) & 0xFFFF0000u;
    const float r1 = n - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = uh >> 16; m = um >> 16; l = __float_as_uint(r2) >> 16;
}
constexpr unsigned BF16_ONE = 0x3F80u;
constexpr unsigned BF16_BIG = 0x7F7Fu;      // largest finite bf16: the "norm" of the rows that pad the last tile

// Half a wave per row of A (nA rows) and B (nB rows, then the padding of the last 128-candidate tile) of every pair, four
// rows per wave in flight (a lane holds one float4 of two rows):
//   * ||x||^2 as RN32 of a binary64 sum of exact squares (error <= 2^-24 relative: the screens' bounds count on it);
//     rows past nB get +inf (fp32 screen) -- the MFMA kernels stage whole tiles;
//   * MODE 1 (bf16x3 screen): the row's split image [128 x bf16 hi | 128 x bf16 lo] (A is scaled by -2 first: exact), and
//     for candidates the 16-byte k-slot (nb_h, nb_m, nb_l, 1, 1, 1, 0, 0) that adds the norms inside the MFMA chain;
//   * MODE 2 (two-stage screen): the same for the candidates only (the rows of A that need the bf16x3 pass are split
//     later, fine_rows_kernel), plus the fp16 images [128 x fp16] of -2 A and of B (round to nearest even) and an upper
//     bound of the 2-norm of each row's rounding residual x - x_h (x_h = what the image holds, unscaled).
constexpr int PREP_ROWS = 16;       // rows per 256-thread workgroup
__device__ __forceinline__ unsigned f16_bits(float x)
{
    const _Float16 h = (_Float16)x;                      // v_cvt_f16_f32: round to nearest even (the kernels never change the mode)
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_value(unsigned bits)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
}
template <int MODE>
__global__ __launch_bounds__(256) void prep_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.y];
    const int nA = pair_nA(c), nB = pair_nB(c), lane = threadIdx.x & 63, k4 = lane & 31;
    const int padded = nm_divup_dev(nB, TILE_C) * TILE_C;
    if (blockIdx.x == 0 && threadIdx.x == 0 && c.fb_count) {     // first launch of a match call: resets the row lists
        *c.fb_count = 0;
        if (MODE == 2) *pair_f1_count(c) = 0;
    }
    const int r0 = blockIdx.x * PREP_ROWS + (threadIdx.x >> 6) * 4 + (lane >> 5);    // this lane's rows: r0 and r0 + 2
    float4 v[2];
    int row[2]; bool isA[2], live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int i = r0 + 2 * q;
        isA[q] = i < nA;
        if (!isA[q]) i -= nA;
        row[q] = i;
        live[q] = isA[q] || i < nB;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live[q]) v[q] = reinterpret_cast<const float4 *>((isA[q] ? c.A : c.B) + (size_t)i * DIM)[k4];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = row[q];
        if (!live[q]) {                                   // uniform per half wave
            if (i < padded && k4 == 0) {
                c.nb[i] = __builtin_inff();
                if (MODE) c.nbslot[i] = make_uint4(BF16_BIG, BF16_ONE << 16, BF16_ONE | (BF16_ONE << 16), 0u);
            }
            continue;
        }
        const float4 x = v[q];
        double acc = ((double)x.x * (double)x.x + (double)x.y * (double)x.y) + ((double)x.z * (double)x.z + (double)x.w * (double)x.w);
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        const float nrm = (float)acc;
        const float sc = isA[q] ? -2.0f : 1.0f;
        if (MODE == 1 || (MODE == 2 && !isA[q])) {
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            bf16_split(sc * x.x, h0, l0); bf16_split(sc * x.y, h1, l1);
            bf16_split(sc * x.z, h2, l2); bf16_split(sc * x.w, h3, l3);
            unsigned *dst = (isA[q] ? c.As : c.Bs) + (size_t)i * DIM;
            reinterpret_cast<uint2 *>(dst)[k4] = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
            reinterpret_cast<uint2 *>(dst + 64)[k4] = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
        }
        float res = 0.f;
        if (MODE == 2) {
            const unsigned b0 = f16_bits(sc * x.x), b1 = f16_bits(sc * x.y), b2 = f16_bits(sc * x.z), b3 = f16_bits(sc * x.w);
            unsigned *dst = (isA[q] ? c.Ah : c.Bh) + (size_t)i * (DIM / 2);
            reinterpret_cast<uint2 *>(dst)[k4] = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
            // residual of what the image holds: x - x_h is exact in binary32 (x_h is x rounded to fewer bits; the scaling
            // by -2 and back is exact). An infinite x_h (|2 x| beyond the fp16 range) gives an infinite residual: such
            // rows lie outside the coarse pass's domain anyway (F16_NORM_LIMIT) and go to the bf16x3 pass.
            const float inv = isA[q] ? -0.5f : 1.0f;
            const float e0 = x.x - inv * f16_value(b0), e1 = x.y - inv * f16_value(b1);
            const float e2 = x.z - inv * f16_value(b2), e3 = x.w - inv * f16_value(b3);
            res = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) res += __shfl_xor(res, d);
            // upper bound of the residual's 2-norm: the 131 binary32 roundings above cost < 1e-5 relative, squares that
            // underflow < 1e-18 absolute
            res = __builtin_sqrtf(res) * 1.00002f + 1e-18f;
        }
        if (k4 == 0) {
            (isA[q] ? c.na : c.nb)[i] = nrm;
            if (MODE == 2) (isA[q] ? c.ra : c.rb)[i] = res;
            if (MODE && !isA[q]) {
                unsigned h, m, l;
                bf16_three(nrm, h, m, l);
                c.nbslot[i] = (nrm < 3.0e38f) ? make_uint4(h | (m << 16), l | (BF16_ONE << 16), BF16_ONE | (BF16_ONE << 16), 0u)
                                              : make_uint4(BF16_BIG, BF16_ONE << 16, BF16_ONE | (BF16_ONE << 16), 0u);
            }
        }
    }
}

// max_j ||b_j||^2 of every pair (one workgroup per pair): tightens the error bound of match_finalize_kernel. Two-stage
// screen (c.rb set): also the largest candidate residual norm, into the next float.
__global__ __launch_bounds__(1024) void nbmax_kernel(MatchBatch bt)
{
    __shared__ float s[32];
    const MatchPair &c = bt.p[blockIdx.x];
    const int nB = pair_nB(c);
    if (c.d_plan && threadIdx.x >= 960) {
        // device-sized call: the last wave makes the work plan for the real sizes (make_plan_on: the same function the host
        // runs for the host-sized entries), its lanes sharing the slot count, while the others reduce the norms
        const int lane = threadIdx.x & 63;
        const MatchPlan p = make_plan_on(pair_nA(c), nB, bt.n_cu, bt.n_xcd, lane, 64, [](int v) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
            return v;
        });
        if (lane == 0) *c.d_plan = p;
    }
    // +inf as soon as one candidate norm is not a finite number below NORM_LIMIT (fmax would drop a NaN): the finalize
    // pass then sends EVERY row of the pair to the exact fallback, which is the reference's own scan
    // (eight loads of either array in flight per thread: the kernel is ONE workgroup per pair, i.e. a chain of memory round trips;
    // with one element per trip it took 9 us of a 127 us single-pair match)
    float m = 0.f, mr = 0.f;
    for (int i0 = threadIdx.x; i0 < nB; i0 += 8 * 1024) {
        float v[8], e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 1024;
            v[k] = (i < nB) ? c.nb[i] : 0.f;
            e[k] = (c.rb && i < nB) ? c.rb[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m = (v[k] < NORM_LIMIT) ? __builtin_fmaxf(m, v[k]) : __builtin_inff();
            mr = (e[k] < NORM_LIMIT) ? __builtin_fmaxf(mr, e[k]) : __builtin_inff();
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { m = __builtin_fmaxf(m, __shfl_xor(m, d)); mr = __builtin_fmaxf(mr, __shfl_xor(mr, d)); }
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = m; s[16 + (threadIdx.x >> 6)] = mr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { m = __builtin_fmaxf(m, s[w]); mr = __builtin_fmaxf(mr, s[16 + w]); }
        c.nbmax[0] = m;
        c.nbmax[1] = mr;
    }
}

// Running best / second best / third best of one lane's query, as integer KEYS. The MFMA pipe and the VALU do not overlap
// on a SIMD here (measured: every epilogue instruction adds to the kernel time), so the selection is written for the
// fewest instructions: a key is the distance's bit pattern with its low 5 bits replaced by the slot (0..31) of the
// candidate inside the current 64-candidate group (v_and_or_b32), keys order like the distances under signed integer
// comparison (positive floats; the slightly negative values that rounding can produce for near-duplicates all lie
// within the finalize margin of zero), and inserting a key into a sorted triple is min + med3 + med3 -- the slot
// travels inside the key. Per group the triple of the 32 accumulator values is built this way (4 instructions per
// element) and its three keys are merged into the running triple, whose two best also carry the group number.
// Truncating 5 mantissa bits lowers a value by < 2^-18 relative: the finalize margin accounts for it.
constexpr int KEY_SLOT_BITS = 5;
constexpr int KEY_INF = 0x7fffffff;

__device__ __forceinline__ int imed3(int a, int b, int c)        // the compiler only recognises some of the min/max forms
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void key_insert3(int &k1, int &k2, int &k3, int k)
{
    const int n3 = imed3(k2, k3, k), n2 = imed3(k1, k2, k);
    k1 = min(k1, k); k2 = n2; k3 = n3;
}
struct Top3 { int k1, k2, k3, t1, t2; };                          // keys + group tags of the two best
__device__ __forceinline__ void top3_merge(Top3 &t, int k, int tag)
{
    t.k3 = imed3(t.k2, t.k3, k);
    const bool lt1 = k < t.k1, lt2 = k < t.k2;
    t.k2 = lt1 ? t.k1 : (lt2 ? k : t.k2);
    t.t2 = lt1 ? t.t1 : (lt2 ? tag : t.t2);
    t.k1 = lt1 ? k : t.k1;
    t.t1 = lt1 ? tag : t.t1;
}
template <int BITS = KEY_SLOT_BITS>
__device__ __forceinline__ float key_value(int k) { return __int_as_float(k & ~((1 << BITS) - 1)); }

// used by the lane-half merge at the end of the kernel (values + full indices)
struct Top2 { float m1, m2, m3; int i1, i2; };

__device__ __forceinline__ void top2_insert(Top2 &t, float d, int j)
{
    t.m3 = __builtin_amdgcn_fmed3f(t.m2, t.m3, d);      // third smallest of {m1, m2, m3, d} (m2 <= m3)
    const bool lt1 = d < t.m1, lt2 = d < t.m2;
    t.m2 = lt1 ? t.m1 : (lt2 ? d : t.m2);
    t.i2 = lt1 ? t.i1 : (lt2 ? j : t.i2);
    t.m1 = lt1 ? d : t.m1;
    t.i1 = lt1 ? j : t.i1;
}

// grid = G persistent workgroups (see MatchPlan), 512 threads = 8 waves (2 per SIMD). Wave w owns queries i0 + 32 w .. +31
// of the current query block with their MFMA fragments resident in VGPRs; candidate tiles of 128 rows stream through a
// double-buffered LDS image (row pitch KP). Dynamic LDS: 2 * TILE_C * KP floats.
// MFMA orientation: rows (accumulator registers) = candidates, columns (lanes) = queries, so every lane scans its own
// query's candidates in increasing index order and the running best/second-best never crosses lanes in the loop.
//
// What the instruction stream is built around (measured on MI355X, profiles/r02_mfma_*_microbench.txt): the fp32 MFMA
// shares the SIMD's vector datapath with the VALU -- every VALU instruction of either wave of a SIMD takes ~4-5 cycles
// away from the MFMAs (a bare 32x32x2 loop runs 154 TFLOP/s at 2.39 GHz; 2 VALU per MFMA cut it to 121) -- and an LDS
// operand read that is waited for right after its issue stalls the wave for the LDS latency every 4 MFMAs. So:
//   * staging uses raw buffer loads (the SRD's range check zero-fills rows >= nB: no compares, no address arithmetic
//     beyond one per-lane offset; the tile advances through the scalar offset) and the norms array is padded with +inf;
//   * the candidate fragments of k-group t+1 are requested before the MFMAs of k-group t;
//   * selection stays branch-free at 4 VALU instructions per value (Top3). A threshold test per value (one v_cmp +
//     wave vote + scalar branch, inserting only values below the running third best) was measured SLOWER (315 vs 298 us):
//     a segment restarts its triple every ~2 300 candidates, so ~30 % of the values still take the insertion and the 32
//     branches per group cost more than they save; staggering the two waves of a SIMD by half a tile changed nothing.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// fold the 2 x 16 accumulator values of one 64-candidate group into the running triple (see Top3 above)
template <int BITS = KEY_SLOT_BITS>
__device__ __forceinline__ void select_half(const f32x16 &acc0, const f32x16 &acc1, Top3 &best, int tag)
{
    int g1 = KEY_INF, g2 = KEY_INF, g3 = KEY_INF;
#pragma unroll
    for (int e = 0; e < 16; ++e) key_insert3(g1, g2, g3, (__float_as_int(acc0[e]) & ~((1 << BITS) - 1)) | e);
#pragma unroll
    for (int e = 0; e < 16; ++e) key_insert3(g1, g2, g3, (__float_as_int(acc1[e]) & ~((1 << BITS) - 1)) | (16 + e));
    if (__any(g1 < best.k3)) {
        top3_merge(best, g1, tag);
        top3_merge(best, g2, tag);
        top3_merge(best, g3, tag);
    }
}

// 2 x (32 candidates x 32 queries x 128 + 1 k-pairs): acc[g] = |b_j|^2 + |a_i|^2 - 2 a_i.b_j for candidates
// 64 half + 32 g + (row of the accumulator layout). rowp = this lane's candidate row of group 0 at its k offset 4 h.
// The MFMAs are hand-placed (inline asm): the two accumulator chains alternate, and the "memory" clobbers keep the
// fragment reads of k-group t+1 (ordinary loads, counted and waited for by the compiler at their first use) above the
// MFMAs of k-group t. hipcc's own schedule of the equivalent builtins waits for every read right after issuing it and
// runs the chains one after the other.
#define NM_MFMA "v_mfma_f32_32x32x2_f32 "
// k-groups T0 .. T1 - 1 (8 values of k each) of both accumulators; NORMS: the chain starts with the augmented k-pair, otherwise
// from the inline constant 0. mfma_half = all 16 groups (the matcher's screen). The distance pass splits k into two chains.
template <int T0, int T1, bool NORMS>
__device__ __forceinline__ void mfma_kgroups(f32x16 &acc0, f32x16 &acc1, const float *rowp, const float *normp,
                                             const float4 (&qf)[16], float nq)
{
    const float *r0 = rowp, *r1 = rowp + 32 * KP;
    float4 c0 = *reinterpret_cast<const float4 *>(r0 + 8 * T0), c1 = *reinterpret_cast<const float4 *>(r1 + 8 * T0);
    if (NORMS) {
        const float cn0 = normp[0], cn1 = normp[32 * KP];  // column 128 + h: (nb_j, 1) for h = (0, 1)
        // augmented k-pair: (nb_j * 1) + (1 * na_i), accumulators start from the inline constant 0
        asm volatile(NM_MFMA "%0, %2, %4, 0\n\t" NM_MFMA "%1, %3, %4, 0"
                     : "=&v"(acc0), "=&v"(acc1) : "v"(cn0), "v"(cn1), "v"(nq) : "memory");
    }
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        float4 n0 = c0, n1 = c1;
        if (t + 1 < T1) {                               // next k-group's fragments fly during this group's 8 MFMAs
            n0 = *reinterpret_cast<const float4 *>(r0 + 8 * (t + 1));
            n1 = *reinterpret_cast<const float4 *>(r1 + 8 * (t + 1));
        }
        if (!NORMS && t == T0)                          // first instruction of a chain without the norm pair: C = 0
            asm volatile(NM_MFMA "%0, %2, %10, 0\n\t" NM_MFMA "%1, %6, %10, 0\n\t"
                         NM_MFMA "%0, %3, %11, %0\n\t" NM_MFMA "%1, %7, %11, %1\n\t"
                         NM_MFMA "%0, %4, %12, %0\n\t" NM_MFMA "%1, %8, %12, %1\n\t"
                         NM_MFMA "%0, %5, %13, %0\n\t" NM_MFMA "%1, %9, %13, %1"
                         : "=&v"(acc0), "=&v"(acc1)
                         : "v"(c0.x), "v"(c0.y), "v"(c0.z), "v"(c0.w), "v"(c1.x), "v"(c1.y), "v"(c1.z), "v"(c1.w),
                           "v"(qf[t].x), "v"(qf[t].y), "v"(qf[t].z), "v"(qf[t].w)
                         : "memory");
        else
            asm volatile(NM_MFMA "%0, %2, %10, %0\n\t" NM_MFMA "%1, %6, %10, %1\n\t"
                         NM_MFMA "%0, %3, %11, %0\n\t" NM_MFMA "%1, %7, %11, %1\n\t"
                         NM_MFMA "%0, %4, %12, %0\n\t" NM_MFMA "%1, %8, %12, %1\n\t"
                         NM_MFMA "%0, %5, %13, %0\n\t" NM_MFMA "%1, %9, %13, %1"
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(c0.x), "v"(c0.y), "v"(c0.z), "v"(c0.w), "v"(c1.x), "v"(c1.y), "v"(c1.z), "v"(c1.w),
                           "v"(qf[t].x), "v"(qf[t].y), "v"(qf[t].z), "v"(qf[t].w)
                         : "memory");
        c0 = n0; c1 = n1;
    }
    // an MFMA's result may be read by a non-MFMA instruction only 18 wait states after its issue (16-pass XDL op):
    // the compiler does not see inside the asm statements, so the padding is explicit
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
}
__device__ __forceinline__ void mfma_half(f32x16 &acc0, f32x16 &acc1, const float *rowp, const float *normp,
                                          const float4 (&qf)[16], float nq)
{
    mfma_kgroups<0, 16, true>(acc0, acc1, rowp, normp, qf, nq);
}
#undef NM_MFMA

// The bf16x3 screen: the same two 32 x 32 accumulators from v_mfma_f32_32x32x16_bf16 on the SPLIT operands
//   a = a_h + a_l + r_a,  b = b_h + b_l + r_b   (bf16 pieces, |r| <= 2^-16 |x|):   a.b ~ a_h.b_h + a_h.b_l + a_l.b_h
// -- 3 x 8 instructions of 32 cycles per accumulator instead of 64 fp32 instructions of 64 cycles (5.3x fewer MFMA
// cycles), plus one instruction whose k-slots carry the norms as exact sums of three bf16 each. rowp = this lane's
// candidate row of group 0 at its k offset (8 h bf16 = 4 h dwords); a row is [hi: 64 dwords | lo: 64 dwords | slot: 4].
// qf[s] / qf[8 + s] = the query's hi / lo pieces of k-step s (k = 16 s + 8 h ..+7), already scaled by -2.
// The value differs from the exact distance by at most MatchBatch::err_coeff (sqrt na + sqrt nb)^2.
#define NM_MFMA "v_mfma_f32_32x32x16_bf16 "
// One k-step (k = 16 T .. 16 T + 15) of both accumulators: 6 MFMAs, and -- SELECT -- between them the insertion of four
// values of the PREVIOUS 64-candidate group (p0..p3 = its accumulator registers 4 T .. 4 T + 3, slots E0 .. E0 + 3) into
// that group's running key triple (g1 <= g2 <= g3): v_and_or (key), v_med3, v_med3, v_min per value, two or three
// between consecutive MFMAs. Unlike the fp32 MFMA, the bf16 MFMA leaves the SIMD's VALU port free while it runs
// (MI355X_MICROARCH.md, "instructions hidden per MFMA gap"), so the selection of one group costs nothing while the next
// group's products are formed; with the selection AFTER the MFMAs (as in the fp32 kernel) the two waves of a SIMD ran in
// lockstep, their selections coincided, and the MFMA pipe idled 49 % of the time (profiles/r02_j_*).
template <bool SELECT, int E0>
__device__ __forceinline__ void bf16_kstep(f32x16 &acc0, f32x16 &acc1, const u32x4 h0, const u32x4 l0, const u32x4 h1,
                                           const u32x4 l1, const u32x4 qh, const u32x4 ql, float p0, float p1, float p2,
                                           float p3, int &g1, int &g2, int &g3)
{
    if (SELECT) {
        int k;
        const int mask = ~((1 << KEY_SLOT_BITS) - 1);
        asm volatile(NM_MFMA "%0, %6, %10, %0\n\t"
                     "v_and_or_b32 %5, %12, %16, %17\n\t"
                     "v_med3_i32 %4, %3, %4, %5\n\t"
                     "v_med3_i32 %3, %2, %3, %5\n\t"
                     NM_MFMA "%1, %8, %10, %1\n\t"
                     "v_min_i32 %2, %2, %5\n\t"
                     "v_and_or_b32 %5, %13, %16, %18\n\t"
                     "v_med3_i32 %4, %3, %4, %5\n\t"
                     NM_MFMA "%0, %6, %11, %0\n\t"
                     "v_med3_i32 %3, %2, %3, %5\n\t"
                     "v_min_i32 %2, %2, %5\n\t"
                     "v_and_or_b32 %5, %14, %16, %19\n\t"
                     NM_MFMA "%1, %8, %11, %1\n\t"
                     "v_med3_i32 %4, %3, %4, %5\n\t"
                     "v_med3_i32 %3, %2, %3, %5\n\t"
                     "v_min_i32 %2, %2, %5\n\t"
                     NM_MFMA "%0, %7, %10, %0\n\t"
                     "v_and_or_b32 %5, %15, %16, %20\n\t"
                     "v_med3_i32 %4, %3, %4, %5\n\t"
                     "v_med3_i32 %3, %2, %3, %5\n\t"
                     NM_MFMA "%1, %9, %10, %1\n\t"
                     "v_min_i32 %2, %2, %5"
                     : "+v"(acc0), "+v"(acc1), "+v"(g1), "+v"(g2), "+v"(g3), "=&v"(k)
                     : "v"(h0), "v"(l0), "v"(h1), "v"(l1), "v"(qh), "v"(ql), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(mask),
                       "n"(E0), "n"(E0 + 1), "n"(E0 + 2), "n"(E0 + 3)
                     : "memory");
    } else {
        asm volatile(NM_MFMA "%0, %2, %6, %0\n\t" NM_MFMA "%1, %4, %6, %1\n\t"
                     NM_MFMA "%0, %2, %7, %0\n\t" NM_MFMA "%1, %4, %7, %1\n\t"
                     NM_MFMA "%0, %3, %6, %0\n\t" NM_MFMA "%1, %5, %6, %1"
                     : "+v"(acc0), "+v"(acc1)
                     : "v"(h0), "v"(l0), "v"(h1), "v"(l1), "v"(qh), "v"(ql)
                     : "memory");
    }
}

// (acc0, acc1) <- the 64-candidate group at rowp; SELECT: the previous group's accumulators (prev0, prev1) are folded into
// the key triple (g1, g2, g3) meanwhile. The group's values may be read by the VALU 10 wait states after the last MFMA
// (8-pass XDL op): the next call's first VALU instruction comes after two more MFMAs and the explicit s_nop below.
template <bool SELECT>
__device__ __forceinline__ void mfma_half_bf16(f32x16 &acc0, f32x16 &acc1, const f32x16 &prev0, const f32x16 &prev1,
                                               const char *tb, const unsigned (&foff)[8], const char *slotp,
                                               const u32x4 (&qf)[16], const u32x4 qslot, int &g1, int &g2, int &g3)
{
    // tb = this lane's candidate row of group 0 in the tile image (512-byte rows, 16-byte chunks XOR-swizzled by row & 15,
    // see bf16 staging in match_top2_kernel); foff[t] = byte offset of the lane's hi chunk of k-step t; lo = + 256;
    // group 1 = + 32 rows
    constexpr int G1 = 32 * 512;
    u32x4 h0 = *reinterpret_cast<const u32x4 *>(tb + foff[0]), l0 = *reinterpret_cast<const u32x4 *>(tb + 256 + foff[0]);
    u32x4 h1 = *reinterpret_cast<const u32x4 *>(tb + G1 + foff[0]), l1 = *reinterpret_cast<const u32x4 *>(tb + G1 + 256 + foff[0]);
    const u32x4 s0 = *reinterpret_cast<const u32x4 *>(slotp), s1 = *reinterpret_cast<const u32x4 *>(slotp + 32 * 16);
    asm volatile(NM_MFMA "%0, %2, %4, 0\n\t" NM_MFMA "%1, %3, %4, 0\n\ts_nop 15\n\ts_nop 3"
                 : "=&v"(acc0), "=&v"(acc1) : "v"(s0), "v"(s1), "v"(qslot) : "memory");
#define NM_KSTEP(T)                                                                                                      \
    {                                                                                                                    \
        u32x4 nh0 = h0, nl0 = l0, nh1 = h1, nl1 = l1;                                                                    \
        if (T + 1 < 8) { /* next k-step's fragments fly during this step's 6 MFMAs */                                    \
            nh0 = *reinterpret_cast<const u32x4 *>(tb + foff[(T + 1) & 7]);                                              \
            nl0 = *reinterpret_cast<const u32x4 *>(tb + 256 + foff[(T + 1) & 7]);                                        \
            nh1 = *reinterpret_cast<const u32x4 *>(tb + G1 + foff[(T + 1) & 7]);                                         \
            nl1 = *reinterpret_cast<const u32x4 *>(tb + G1 + 256 + foff[(T + 1) & 7]);                                   \
        }                                                                                                                \
        const f32x16 &pv = (T < 4) ? prev0 : prev1;                                                                      \
        bf16_kstep<SELECT, 4 * T>(acc0, acc1, h0, l0, h1, l1, qf[T], qf[8 + T], pv[(4 * T) & 15], pv[(4 * T + 1) & 15],    \
                                  pv[(4 * T + 2) & 15], pv[(4 * T + 3) & 15], g1, g2, g3);                                \
        h0 = nh0; l0 = nl0; h1 = nh1; l1 = nl1;                                                                          \
    }
    NM_KSTEP(0) NM_KSTEP(1) NM_KSTEP(2) NM_KSTEP(3) NM_KSTEP(4) NM_KSTEP(5) NM_KSTEP(6) NM_KSTEP(7)
#undef NM_KSTEP
}
#undef NM_MFMA

// The coarse pass of the two-stage screen: ONE product per k (v_mfma_f32_32x32x16_f16 on the fp16 images of -2 A and B:
// 8 + 1 MFMAs per accumulator and 64-candidate group instead of 24 + 1), norms through the same bf16 k-slot instruction.
// Its value differs from the exact distance by the fp16 rounding of the operands, which match_finalize_kernel<1> bounds
// per row from the residual norms (prep_kernel<2>); the rows it cannot prove are screened again by the bf16x3 kernel.
// tb = this lane's candidate row of group 0 in the tile image (256-byte rows, 16-byte chunks XOR-swizzled by row & 15);
// foff[t] = byte offset of the lane's chunk of k-step t; group 1 = + 32 rows. qf[s] = the query's k-step s (k = 16 s + 8 h ..+7).
#ifdef NM_STUB_NOMFMA                           // diagnostic builds only (see f16_kstep): the instruction becomes an assembler comment
#define NM_MFMA_H "; v_mfma_f32_32x32x16_f16 "
#else
#define NM_MFMA_H "v_mfma_f32_32x32x16_f16 "
#endif
// The coarse pass folds once per TILE ITERATION: the keys of the two groups selected in it (2 n - 1, selected beside the first
// group of tile n, and 2 n) share one pair of running keys, told apart by a sixth slot bit; 2^-17 instead of 2^-18 of a
// value is dropped, which the finalize constants cover.
constexpr int COARSE_SLOT_BITS = 6;
// Selection in the coarse pass keeps the TWO smallest keys of a 64-candidate group (v_and_or, v_med3, v_min: three
// instructions per value instead of four -- the selection is this kernel's largest single cost) and hands the second
// one to the running triple a second time as the group's third: every value of the group it did not keep is >= it, which
// is all the third key is for (the lower bound `rest` of the finalize pass). The price: a segment whose two best
// candidates share a group reports third == second and its row goes to the second pass (~0.5 % of the rows more).
// Diagnostic builds only (tools/build_variant.py stub_X nm_match.hip -DNM_STUB_X=1; tools/gpu_coarse_ab.sh): NM_STUB_NOSELECT drops
// the selection instructions, NM_STUB_NOMFMA the matrix instructions, NM_STUB_NODMA the tile requests of the ring. Round 6, 16-pair
// launch, one box, three alternations, us per pair: whole 36.8 | no selection 31.7 | no MFMA 22.3 | no requests 33.6 | neither
// selection nor MFMA 16.2 | none of the three 12.6 (fragment reads, norm slots, barriers, fold, publication). The MFMAs' share is
// their own pipe time (37.8 GFLOP at the 2.5 PFLOP/s peak = 15.1 us per pair): NOTHING else of an iteration runs under them -- the
// per-tile barrier keeps the eight waves in the same phase, so a SIMD's two waves want the matrix pipe together and stall together.
// profiles/r06_y_coarse_stubs.txt. No stub executes in the product.
template <bool SELECT_, int E0>
__device__ __forceinline__ void f16_kstep(f32x16 &acc0, f32x16 &acc1, const u32x4 h0, const u32x4 h1, const u32x4 qh,
                                          float p0, float p1, float p2, float p3, int &g1, int &g2)
{
#ifdef NM_STUB_NOSELECT
    constexpr bool SELECT = false;
#else
    constexpr bool SELECT = SELECT_;
#endif
    if (SELECT) {
        int k;
        const int mask = ~((1 << COARSE_SLOT_BITS) - 1);
        asm volatile(NM_MFMA_H "%0, %5, %7, %0\n\t"
                     "v_and_or_b32 %4, %8, %12, %13\n\t"
                     "v_med3_i32 %3, %2, %3, %4\n\t"
                     "v_min_i32 %2, %2, %4\n\t"
                     "v_and_or_b32 %4, %9, %12, %14\n\t"
                     "v_med3_i32 %3, %2, %3, %4\n\t"
                     "v_min_i32 %2, %2, %4\n\t"
                     NM_MFMA_H "%1, %6, %7, %1\n\t"
                     "v_and_or_b32 %4, %10, %12, %15\n\t"
                     "v_med3_i32 %3, %2, %3, %4\n\t"
                     "v_min_i32 %2, %2, %4\n\t"
                     "v_and_or_b32 %4, %11, %12, %16\n\t"
                     "v_med3_i32 %3, %2, %3, %4\n\t"
                     "v_min_i32 %2, %2, %4"
                     : "+v"(acc0), "+v"(acc1), "+v"(g1), "+v"(g2), "=&v"(k)
                     : "v"(h0), "v"(h1), "v"(qh), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(mask),
                       "n"(E0), "n"(E0 + 1), "n"(E0 + 2), "n"(E0 + 3)
                     : "memory");
    } else {
        asm volatile(NM_MFMA_H "%0, %2, %4, %0\n\t" NM_MFMA_H "%1, %3, %4, %1"
                     : "+v"(acc0), "+v"(acc1) : "v"(h0), "v"(h1), "v"(qh) : "memory");
    }
}

// The coarse pass is software-pipelined over HALF groups (4 k-steps of both accumulators): with two MFMAs per k-step a
// k-step lasts ~64-128 cycles, less than an LDS read under load, so the fragments of a whole half group (8 x 16 bytes per
// lane) are requested while the previous half group is multiplied.
// fr[2 t], fr[2 t + 1] = the lane's chunks of k-step KS0 + t for the two accumulators.
// (Round 6, measured: the fragment reads are NOT what the kernel waits for -- a diagnostic build that issues HALF of them (the
// second accumulator reusing the first one's chunk; what 64 queries per wave would save) runs the 16-pair launch in 593.4-595.4
// against 600.2-601.0 us, three alternations on one box: -1.2 %. profiles/r06_y_coarse_half_fragment_reads.txt)
template <int KS0>
__device__ __forceinline__ void f16_fetch(u32x4 (&fr)[8], const char *tb, const unsigned (&foff)[8])
{
    constexpr int G1 = 32 * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fr[2 * t] = *reinterpret_cast<const u32x4 *>(tb + foff[KS0 + t]);
        fr[2 * t + 1] = *reinterpret_cast<const u32x4 *>(tb + G1 + foff[KS0 + t]);
    }
}
// the norms of a 64-candidate group start its two accumulators (bf16 k-slots, as in the bf16x3 kernel)
__device__ __forceinline__ void f16_slots(f32x16 &acc0, f32x16 &acc1, const char *slotp, const u32x4 qslot)
{
    const u32x4 s0 = *reinterpret_cast<const u32x4 *>(slotp), s1 = *reinterpret_cast<const u32x4 *>(slotp + 32 * 16);
    // No wait states behind them: the next instruction on acc0 is an MFMA that accumulates into exactly the registers this
    // one writes (the XDL pipe interlocks that case, whatever the operand type), and the first VALU instruction that reads
    // an accumulator reads one of the PREVIOUS group, whose last MFMA is more than 16 instructions back.
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %4, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, 0"
                 : "=&v"(acc0), "=&v"(acc1) : "v"(s0), "v"(s1), "v"(qslot) : "memory");
}
// k-steps KS0 .. KS0 + 3 of (acc0, acc1); SELECT: the 16 values of pv (one accumulator of the PREVIOUS group; slots
// 4 KS0 .. 4 KS0 + 15 of that group) are folded into the key triple meanwhile
template <bool SELECT, int KS0, int EB = 0>        // EB: slot base of the selected group (0: group 2 n - 1, 32: group 2 n)
__device__ __forceinline__ void f16_half(f32x16 &acc0, f32x16 &acc1, const f32x16 &pv, const u32x4 (&fr)[8],
                                         const u32x4 (&qf)[16], int &g1, int &g2)
{
    f16_kstep<SELECT, EB + 4 * KS0>(acc0, acc1, fr[0], fr[1], qf[KS0], pv[0], pv[1], pv[2], pv[3], g1, g2);
    f16_kstep<SELECT, EB + 4 * KS0 + 4>(acc0, acc1, fr[2], fr[3], qf[KS0 + 1], pv[4], pv[5], pv[6], pv[7], g1, g2);
    f16_kstep<SELECT, EB + 4 * KS0 + 8>(acc0, acc1, fr[4], fr[5], qf[KS0 + 2], pv[8], pv[9], pv[10], pv[11], g1, g2);
    f16_kstep<SELECT, EB + 4 * KS0 + 12>(acc0, acc1, fr[6], fr[7], qf[KS0 + 3], pv[12], pv[13], pv[14], pv[15], g1, g2);
}
#undef NM_MFMA_H

// End of a segment: decode (value, candidate index) of the two best, merge the two lane halves (same query, disjoint
// candidates), publish into this segment's slot of the query block. c0 = first candidate of the segment.
// BITS 5: tag = 64-candidate group of the segment. BITS 6 (coarse pass): tag = tile iteration n, the sixth slot bit tells
// group 2 n - 1 (0) from group 2 n (1).
template <int BITS = KEY_SLOT_BITS>
__device__ __forceinline__ void segment_publish(const Top3 &best, const MatchPlan &plan, const PlanGroup &grp, int pc, int qbl,
                                                int vg, bool ends_piece, int c0, int h, int qi, int nA, int S,
                                                float4 *__restrict__ partial, float *__restrict__ partial3)
{
    auto index_of = [&](int k, int tag) {
        const int slot = k & 31, g = slot >> 4, e = slot & 15;
        const int group = (BITS == 6) ? 2 * tag - 1 + ((k >> 5) & 1) : tag;
        return c0 + group * 64 + 32 * g + (e & 3) + 8 * (e >> 2) + 4 * h;
    };
    Top2 m;
    m.m1 = key_value<BITS>(best.k1); m.m2 = key_value<BITS>(best.k2); m.m3 = key_value<BITS>(best.k3);
    m.i1 = (best.k1 != KEY_INF) ? index_of(best.k1, best.t1) : -1;
    m.i2 = (best.k2 != KEY_INF) ? index_of(best.k2, best.t2) : -1;
    if (best.k1 == KEY_INF) m.m1 = __builtin_inff();
    if (best.k2 == KEY_INF) m.m2 = __builtin_inff();
    if (best.k3 == KEY_INF) m.m3 = __builtin_inff();
    Top2 o;
    o.m1 = __shfl_xor(m.m1, 32); o.m2 = __shfl_xor(m.m2, 32); o.m3 = __shfl_xor(m.m3, 32);
    o.i1 = __shfl_xor(m.i1, 32); o.i2 = __shfl_xor(m.i2, 32);
    if (o.i1 >= 0) top2_insert(m, o.m1, o.i1);
    if (o.i2 >= 0) top2_insert(m, o.m2, o.i2);
    m.m3 = __builtin_fminf(m.m3, o.m3);         // o.m3 >= o.m2 >= the merged m2: only the third value can change
    // slot = how many segments of this query block come before this one in the plan's order
    const int slot = plan_slot(plan, grp, pc, qbl, vg);
    if (h == 0 && qi < nA) {
        partial[(size_t)qi * S + slot] = make_float4(m.m1, __int_as_float(m.i1), m.m2, __int_as_float(m.i2));
        partial3[(size_t)qi * S + slot] = m.m3;
        if (pc == plan.C - 1 && ends_piece) {     // this segment ends the block: blank the slots nobody writes
            for (int k = slot + 1; k < S; ++k) {
                partial[(size_t)qi * S + k] = make_float4(__builtin_inff(), __int_as_float(-1), __builtin_inff(), __int_as_float(-1));
                partial3[(size_t)qi * S + k] = __builtin_inff();
            }
        }
    }
}

// SCR 0 (fp32 screen): A, B are the fp32 descriptor rows. SCR 1 (bf16x3): A, B are the split images written by prep_kernel
// (same 512-byte rows) and nbslot replaces nb.
// Device-sized launches (d_plan != NULL): nA_arg / nB_arg are the capacities, the real sizes and the plan made for them
// (nbmax_kernel) are read from device memory, and the grid is one workgroup per CU of which the first plan.G work.
template <int SCR>
__device__ __forceinline__ void match_top2_body(const float *__restrict__ A, int nA_arg,
                                                const float *__restrict__ B, int nB_arg,
                                                const float *__restrict__ na, const float *__restrict__ nb,
                                                const uint4 *__restrict__ nbslot,
                                                MatchPlan plan_arg, float4 *__restrict__ partial,
                                                float *__restrict__ partial3,
                                                const int *__restrict__ d_nA, const int *__restrict__ d_nB,
                                                const MatchPlan *__restrict__ d_plan, const int wg)
{
    static_assert(SCR == 0 || SCR == 1, "the coarse pass of the two-stage screen has its own kernel (match_coarse_kernel)");
    constexpr bool BF16 = SCR == 1, DMA = SCR != 0;
    constexpr int ROWB = DIM * 4;                         // bytes per row of the A / B images these screens read
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 5, scol = (tid & 31) * 4;     // staging coordinates: 16 rows x 32 float4 per pass
    int nA = nA_arg, nB = nB_arg;
    MatchPlan plan = plan_arg;
    if (d_plan) {                                         // uniform: scalar loads
        if (d_nA) nA = min(max(*d_nA, 0), nA_arg);
        if (d_nB) nB = min(max(*d_nB, 0), nB_arg);
        plan = *d_plan;
        if (nA <= 0 || nB <= 0 || wg >= plan.G) return;   // an empty set is a no-op for the pair, as in the reference
    }
    if (wg >= plan.G) return;                             // (group launches are sized for the chip, not for the plan)
    const int S = plan.S;
    const int xg = wg % plan.X, vg = wg / plan.X;         // XCD group (blocks b, b + X share an XCD) and position in it
    const PlanGroup grp = plan_group(plan, xg);
    const unit_t u_begin = group_begin(grp, vg), u_end = group_begin(grp, vg + 1);
    SegIter it;
    it.init(u_begin, u_end);

    // range-checked views: rows >= nB / nA read as zeros (the host refuses sets of 2^22 rows or more)
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, nB * ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, nA * ROWB, 0x00020000);
    const int voff = (srow * DIM + scol) * 4;

    u32x4 st[8];
    float stn = 0.f;
    uint4 sts = make_uint4(0u, 0u, 0u, 0u);
    auto stage_load = [&](int tile) {                     // candidates tile*128 .. +127 -> registers
        const int jb = tile * TILE_C;
#pragma unroll
        for (int it = 0; it < 8; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voff, (jb + 16 * it) * (DIM * 4), 0);
        if (tid < TILE_C) {                               // padded up to T * 128 by prep_kernel
            if (DMA) sts = nbslot[jb + tid];
            else stn = nb[jb + tid];
        }
    };
    auto stage_write = [&](float *buf) {
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4 *>(&buf[(srow + 16 * it) * KP + scol]) = st[it];
        if (tid < TILE_C) {                               // augmented k-pair / k-slot
            if (DMA) *reinterpret_cast<uint4 *>(&buf[tid * KP + DIM]) = sts;
            else *reinterpret_cast<float2 *>(&buf[tid * KP + DIM]) = make_float2(stn, 1.0f);
        }
    };

    // bf16x3 screen: the candidate tiles go global -> LDS directly (buffer_load ... lds: no staging registers, no
    // ds_write pass). One wave-instruction lands 1 KiB = two 512-byte rows, lane-linear, so the tile image has NO row
    // padding; bank conflicts are avoided by XOR-swizzling the 16-byte chunks of a row with (row & 15) -- applied to the
    // per-lane SOURCE address here and to the fragment reads' offsets (foff), the same involution on both sides. The
    // 16-byte norm slots live in their own 2 KiB per buffer behind the two 64 KiB images (same total as the padded layout).
    constexpr int IMG = TILE_C * ROWB, SLOT0 = 2 * IMG, SLOTB = TILE_C * 16;
    constexpr int DMA_N = IMG / (8 * 1024);               // wave-instructions per wave and tile
    char *const ldsb = reinterpret_cast<char *>(lds);
    unsigned foff[8], dvoff[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        foff[t] = (unsigned)(((2 * t + h) ^ (r & 15)) << 4);
        const int p = lane & 31, key = (2 * t + (lane >> 5)) & 15;        // wave-instruction t of a wave covers rows 2 (8 wave + t) + (lane >> 5)
        dvoff[t] = (unsigned)((lane >> 5) * (DIM * 4) + (((p & 16) | ((p & 15) ^ key)) << 4));
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // scalar for the compiler: M0 and the scalar offset depend on it
    auto dma_tile = [&](int tile, int b) {
#pragma unroll
        for (int t = 0; t < DMA_N; ++t) {
            typedef __attribute__((address_space(3))) void lds_void;
            lds_void *dst = (lds_void *)(ldsb + b * IMG + (wave_u * DMA_N + t) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, (int)dvoff[t],
                                                     (tile * TILE_C + (1024 / ROWB) * (wave_u * DMA_N + t)) * ROWB, 0, 0);
        }
    };

    unit_t u;
    while (it.next(plan, grp, u)) {
        int pc, qbl, tt, Lc;
        plan_locate(plan, grp, u, pc, qbl, tt, Lc);
        const int qb = grp.q0 + qbl, t0 = pc * plan.Tc + tt;
        const int ntiles = min(Lc - tt, (int)(u_end - u));
        const int i0 = qb * QB;
        const int c0 = t0 * TILE_C;

        // ---- segment prologue: the first candidate tile is requested first (its latency hides behind the query staging),
        //      queries -> LDS (coalesced) -> per-lane MFMA fragments in VGPRs ----
        if (DMA) { if (tid < TILE_C) sts = nbslot[t0 * TILE_C + tid]; }
        else stage_load(t0);
#pragma unroll 4
        for (int it = 0; it < QB / 16; ++it) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, (i0 + 16 * it) * (DIM * 4), 0);
            *reinterpret_cast<u32x4 *>(&lds[(srow + 16 * it) * KP + scol]) = v;
        }
        __syncthreads();
        float4 qf[16];
        u32x4 qw[16];
        const int qi = i0 + wave * 32 + r;
#pragma unroll
        for (int t = 0; t < 16; ++t) {      // the -2 of  |a|^2 + |b|^2 - 2 a.b  rides on the query fragments (once per segment)
            if (BF16) {              // hi pieces of k-steps 0..7, then the lo pieces (prep_kernel applied the -2)
                qw[t] = *reinterpret_cast<const u32x4 *>(&lds[(wave * 32 + r) * KP + 8 * t + 4 * h]);
            } else {
                const float4 v = *reinterpret_cast<const float4 *>(&lds[(wave * 32 + r) * KP + 8 * t + 4 * h]);
                qf[t] = make_float4(-2.f * v.x, -2.f * v.y, -2.f * v.z, -2.f * v.w);
            }
        }
        const float nav = (qi < nA) ? na[qi] : 0.f;
        const float nq = (h == 0) ? 1.0f : nav;
        u32x4 qslot = {0u, 0u, 0u, 0u};     // k-slots (1, 1, 1, na_h, na_m, na_l, 0, 0) on the lanes that hold k = 0..7
        if (DMA && h == 0) {
            unsigned nh, nm, nl;
            bf16_three(nav, nh, nm, nl);
            qslot = (u32x4){BF16_ONE | (BF16_ONE << 16), BF16_ONE | (nh << 16), nm | (nl << 16), 0u};
        }
        __syncthreads();
        if (DMA) {
            dma_tile(t0, 0);
            if (tid < TILE_C) *reinterpret_cast<uint4 *>(ldsb + SLOT0 + tid * 16) = sts;
            // A wave's LDS-DMA rows are read by all 8 waves after the barrier: every wave must have drained ITS transfers
            // before it arrives. hipcc emits this wait today, but the workgroup-scope fence model does not oblige it to
            // (LDS-DMA is tracked per wave): a half-landed tile would make the screen miss a true neighbour and the finalize
            // pass would then "prove" a wrong row. Explicit, so that no toolchain change can move it.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            stage_write(lds);
        }
        __syncthreads();

        Top3 best;
        best.k1 = best.k2 = best.k3 = KEY_INF; best.t1 = best.t2 = 0;

        // A wave multiplies its 32 queries with 64 candidates at a time (two 32 x 32 accumulators), then folds the 32
        // values each lane holds into its running triple. Within a lane the candidate index increases with (tile, half, g, e).
        f32x16 a0, a1;
        if (DMA) {
            // software pipeline over the 64-candidate groups: the MFMAs of group i run while group i - 1 is selected
            f32x16 b0, b1;
            const bool wave_live = __builtin_amdgcn_readfirstlane(i0 + wave * 32) < nA;
            auto fold = [&](int g1, int g2, int g3, int tag) {
                if (__any(g1 < best.k3)) {
                    top3_merge(best, g1, tag);
                    top3_merge(best, g2, tag);
                    top3_merge(best, g3, tag);
                }
            };
            // tile n + 1 starts landing in the other buffer at the START of tile n (every wave left that buffer before the
            // last barrier); the barrier at the end of tile n waits for it (vmcnt(0), emitted by the compiler)
            for (int n = 0; n < ntiles; ++n) {
                const int b = n & 1;
                const char *tb = ldsb + b * IMG + r * ROWB;
                const char *sp = ldsb + SLOT0 + b * SLOTB + r * 16;
                if (n + 1 < ntiles) {
                    if (tid < TILE_C) sts = nbslot[(t0 + n + 1) * TILE_C + tid];
                    dma_tile(t0 + n + 1, b ^ 1);
                }
                int g1 = KEY_INF, g2 = KEY_INF, g3 = KEY_INF;
                if (!wave_live) {
                    // none of this wave's 32 queries exists (the second pass of the two-stage screen lists a few dozen rows
                    // per 256-row block): it only moves its share of the tiles
                } else if (n == 0) {
                    mfma_half_bf16<false>(a0, a1, b0, b1, tb, foff, sp, qw, qslot, g1, g2, g3);
                } else {
                    mfma_half_bf16<true>(a0, a1, b0, b1, tb, foff, sp, qw, qslot, g1, g2, g3);
                    fold(g1, g2, g3, 2 * n - 1);
                }
                g1 = g2 = g3 = KEY_INF;
                if (wave_live) {
                    mfma_half_bf16<true>(b0, b1, a0, a1, tb + 64 * ROWB, foff, sp + 64 * 16, qw, qslot, g1, g2, g3);
                    fold(g1, g2, g3, 2 * n);
                }
                if (n + 1 < ntiles && tid < TILE_C) *reinterpret_cast<uint4 *>(ldsb + SLOT0 + (b ^ 1) * SLOTB + tid * 16) = sts;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile n + 1 has landed (this wave's share): see the prologue
                __syncthreads();
            }
            if (wave_live) {
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(b0), "+v"(b1));
                select_half(b0, b1, best, 2 * ntiles - 1);
            }
        } else
        for (int n = 0; n < ntiles; ++n) {
            const float *buf = lds + (n & 1) * (TILE_C * KP);
            const float *rowp = buf + r * KP + 4 * h, *normp = buf + r * KP + DIM + h;
            if (n + 1 < ntiles) stage_load(t0 + n + 1);
            mfma_half(a0, a1, rowp, normp, qf, nq);
            select_half(a0, a1, best, 2 * n);
            mfma_half(a0, a1, rowp + 64 * KP, normp + 64 * KP, qf, nq);
            select_half(a0, a1, best, 2 * n + 1);
            if (n + 1 < ntiles) stage_write(lds + ((n + 1) & 1) * (TILE_C * KP));
            __syncthreads();
        }

        segment_publish(best, plan, grp, pc, qbl, vg, tt + ntiles == Lc, c0, h, qi, nA, S, partial, partial3);
        __syncthreads();                                  // the next segment's prologue reuses the LDS
    }
}

// The coarse pass of the two-stage screen as ONE stream of candidate tiles per workgroup. A workgroup's range is 2-3
// segments of ~8 tiles, and with one product per k a tile lasts ~2 us: the segment prologue of the kernels above (query
// rows, their norms and the first candidate tile requested one after the other, each global-memory latency exposed to all
// eight waves) cost as much as four tiles. Here
//   * the query block has its own 64 KiB LDS region, filled by LDS-DMA in the same swizzled row layout as the tiles;
//   * the NEXT segment's queries are requested as soon as every wave holds the current fragments (after the first barrier
//     of the segment), its first two candidate tiles as stream tiles g + 1, g + 2 of the ring below, its norms into
//     registers: a segment change is a barrier that is taken anyway, eight LDS reads and the publication of the previous
//     segment's result;
//   * the tile ring runs across segments: after the barrier of stream tile g (taken once every wave holds its last
//     fragments of g, after the third of the four half groups) tile g + 2 is requested into the buffer g just left.
// A segment that follows a one-tile segment (or starts the range) cannot have been requested ahead and takes the slow
// path: request, wait, barrier. LDS: 2 x 32 KiB tiles | 2 x 2 KiB norm slots | 64 KiB queries = 135 168 B.
// ONE launch per call: the workgroups are persistent over the PAIRS of the call too (every pair's plan is in device memory,
// nbmax_kernel), so the launch, the kernel-argument and plan loads and the drain of the slowest workgroup are paid once
// per call instead of once per pair (~9 us of fixed cost per pair beside ~40 us of tiles when each pair was a launch).
#ifndef NM_COARSE_CROSS
#define NM_COARSE_CROSS 1
#endif
#ifndef NM_COARSE_PRIO_LO
#define NM_COARSE_PRIO_LO 0
#endif
#ifndef NM_COARSE_PRIO_HI
#define NM_COARSE_PRIO_HI 2
#endif
#ifndef NM_COARSE_PRIO
#define NM_COARSE_PRIO 1
#endif
// Diagnostic builds only (tools/build_variant.py ... -DNM_COARSE_STAMPS=1, read by tools/kcoarse_stamps.py): s_memtime at eight
// points of a tile iteration, every wave of workgroup 0, the first 64 iterations. No stamp executes in the product.
#ifdef NM_COARSE_STAMPS
__device__ unsigned long long nm_coarse_stamps[8 * 64 * 8];
#define NM_STAMP(K)                                                                                                      \
    do {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t[K]) :: "memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)
#else
#define NM_STAMP(K) do { } while (0)
#endif
__global__ __launch_bounds__(512, 1) void match_coarse_kernel(MatchBatch bt)
{
    constexpr int ROWB = DIM * 2, IMG = TILE_C * ROWB, SLOT0 = 2 * IMG, SLOTB = TILE_C * 16, QREG = SLOT0 + 2 * SLOTB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *const ldsb = reinterpret_cast<char *>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wg = blockIdx.x;

    // LDS-DMA: one wave-instruction lands 1 KiB = four 256-byte rows, lane-linear; the 16-byte chunks of a row are XOR-swizzled
    // with (row & 15) on the SOURCE side here and in the fragment reads' offsets (foff). Wave-instruction t of a wave covers
    // rows 4 (N wave + t) + (lane >> 4), N = 4 (tile) or 8 (query block): row & 15 = (4 t + (lane >> 4)) & 15 either way.
    unsigned foff[8], dvoff[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) foff[t] = (unsigned)(((2 * t + h) ^ (r & 15)) << 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) dvoff[t] = (unsigned)((lane >> 4) * ROWB + (((lane & 15) ^ ((4 * t + (lane >> 4)) & 15)) << 4));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((address_space(3))) void lds_void;
    // Pair directory: sizes and plan of every pair of the call, read once by sixteen lanes (the device-memory latency of
    // the plan used to be exposed to all eight waves at every pair change) and kept in 1 KiB of LDS behind the query region.
    constexpr int DIR = QREG + QB * ROWB, DIR_STRIDE = 16, NAV = DIR + 1024;     // + 8 x 256 B of query norms
    static_assert(sizeof(MatchPlan) == 10 * sizeof(int) && DIR_STRIDE >= 12, "directory entry: nA, nB, the plan");
    int *const dir = reinterpret_cast<int *>(ldsb + DIR);
    if (tid < bt.n) {
        const MatchPair &c = bt.p[tid];
        const int *pl = reinterpret_cast<const int *>(c.d_plan);
        int *d = dir + tid * DIR_STRIDE;
        d[0] = pair_nA(c); d[1] = pair_nB(c);
#pragma unroll
        for (int k = 0; k < 10; ++k) d[2 + k] = pl[k];
    }
    __syncthreads();

    // What a workgroup needs of the pair a segment belongs to. The segments of ALL pairs of the call form one stream: the
    // segment after a pair's last one is the first of the next pair in which this workgroup has work, requested ahead like any
    // other (until round 5 every pair began with the slow path below: plan load, request, wait, barrier -- ~2 us of 41).
    struct PairCtx {
        int pq, nA, vg;
        MatchPlan plan;
        PlanGroup grp;
        unit_t u_end;
        SegIter it;
        __amdgpu_buffer_rsrc_t rsA, rsB, rsS, rsN;        // fp16 images of the queries / candidates, the candidates' norm k-slots,
        float4 *partial;                                  // the queries' norms
        float *partial3;
    };
    struct Seg { int pc, qbl, tt, Lc, qb, t0, ntiles; };
    // the first pair >= from in which this workgroup has a segment; u: the local unit where that segment starts
    auto open_pair = [&](int from, PairCtx &cx, unit_t &u) -> bool {
        for (int pq = from; pq < bt.n; ++pq) {
            const int *d = dir + pq * DIR_STRIDE;
            const int nA = __builtin_amdgcn_readfirstlane(d[0]), nB = __builtin_amdgcn_readfirstlane(d[1]);
            if (nA <= 0 || nB <= 0) continue;             // an empty set is a no-op for the pair, as in the reference
            MatchPlan plan;
            plan.qblocks = __builtin_amdgcn_readfirstlane(d[2]); plan.T = __builtin_amdgcn_readfirstlane(d[3]);
            plan.G = __builtin_amdgcn_readfirstlane(d[4]); plan.S = __builtin_amdgcn_readfirstlane(d[5]);
            plan.X = __builtin_amdgcn_readfirstlane(d[6]); plan.Gx = __builtin_amdgcn_readfirstlane(d[7]);
            plan.Tc = __builtin_amdgcn_readfirstlane(d[8]); plan.C = __builtin_amdgcn_readfirstlane(d[9]);
            plan.q_base = __builtin_amdgcn_readfirstlane(d[10]); plan.q_rem = __builtin_amdgcn_readfirstlane(d[11]);
            int xg, vg;
            if (bt.pair_xcd > 0) {                        // the pair lives on one XCD: its A and B images cross the fabric once
                if (pq % bt.pair_xcd != wg % bt.pair_xcd) continue;
                xg = 0; vg = wg / bt.pair_xcd;            // (the plan is a one-group plan: X = 1)
                if (vg >= plan.G) continue;
            } else {
                if (wg >= plan.G) continue;
                xg = wg % plan.X; vg = wg / plan.X;
            }
            const MatchPair &c = bt.p[pq];
            cx.pq = pq; cx.nA = nA; cx.plan = plan;
            cx.vg = vg;
            cx.grp = plan_group(plan, xg);
            cx.u_end = group_begin(cx.grp, cx.vg + 1);
            cx.it.init(group_begin(cx.grp, cx.vg), cx.u_end);
            cx.rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(c.Bh), 0, nB * ROWB, 0x00020000);
            cx.rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(c.Ah), 0, nA * ROWB, 0x00020000);
            // (prep_kernel pads the slots to whole tiles)
            cx.rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(c.nbslot), 0, plan.T * SLOTB, 0x00020000);
            cx.rsN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.na), 0, nA * 4, 0x00020000);
            cx.partial = c.partial; cx.partial3 = c.partial3;
            if (cx.it.next(cx.plan, cx.grp, u)) return true;
        }
        return false;
    };
    // A tile = its 32 KiB image (four 1 KiB pieces per wave) and its 2 KiB of norm k-slots (one piece each from waves 0 and 1),
    // all by LDS-DMA. (Until round 5 the slots went through registers -- loaded when the tile was requested, written to LDS an
    // iteration later: the compiler carried them around the loop in a phi, and its copy on the back edge waited with vmcnt(0),
    // i.e. for the tile image requested a few hundred cycles earlier. Every iteration exposed a whole request latency:
    // ~900 of ~5 500 cycles, `tools/kcoarse_stamps.py`.)
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto dma_piece = [&](rsrc_t rsB, int tile, int b, int t) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void *)(ldsb + b * IMG + (wave_u * 4 + t) * 1024), 16, (int)dvoff[t],
                                                 (tile * TILE_C + 4 * (wave_u * 4 + t)) * ROWB, 0, 0);
    };
    auto dma_slots = [&](rsrc_t rsS, int tile, int b) {
        if (wave_u < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lds_void *)(ldsb + SLOT0 + b * SLOTB + wave_u * 1024), 16, lane * 16,
                                                     tile * SLOTB + wave_u * 1024, 0, 0);
    };
    // (Round 6, measured: a tile requested by FOUR of the eight waves, eight pieces each -- waves 0-3, or waves 4-7, the ones the
    // stamps show on the critical path -- is 0.8 % slower either way: 603-608 against 599 us per 16-pair launch, three alternations.
    // profiles/r06_zz_coarse_dma_waves.txt)
    auto dma_tile = [&](rsrc_t rsB, rsrc_t rsS, int tile, int b) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dma_piece(rsB, tile, b, t);
        dma_slots(rsS, tile, b);
    };
    auto dma_queries = [&](rsrc_t rsA, int i0) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void *)(ldsb + QREG + (wave_u * 8 + t) * 1024), 16, (int)dvoff[t & 3],
                                                     (i0 + 4 * (wave_u * 8 + t)) * ROWB, 0, 0);
    };
    auto locate = [&](const PairCtx &cx, unit_t u) {
        Seg sg;
        plan_locate(cx.plan, cx.grp, u, sg.pc, sg.qbl, sg.tt, sg.Lc);
        sg.qb = cx.grp.q0 + sg.qbl; sg.t0 = sg.pc * cx.plan.Tc + sg.tt;
        sg.ntiles = min(sg.Lc - sg.tt, (int)(cx.u_end - u));
        return sg;
    };
    // the norms of a wave's 32 queries: 128 B of the wave's own 256-byte LDS slot, by LDS-DMA as well (rows >= nA read as 0) -- no
    // load of the tile loop returns into registers, so no wait of the compiler's can stand in it
    auto dma_norms = [&](rsrc_t rsN, int qb) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (lds_void *)(ldsb + NAV + wave_u * 256), 4, r * 4, (qb * QB + wave_u * 32) * 4, 0, 0);
    };
    auto norm_read = [&]() { return *reinterpret_cast<const float *>(ldsb + NAV + wave * 256 + r * 4); };

    PairCtx pc_cur, pc_nxt;
    unit_t u;
#ifdef NM_COARSE_STAMPS
    int stamp_it = 0;
    unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    if (!open_pair(0, pc_cur, u)) return;
    pc_nxt = pc_cur;
    Seg cur = locate(pc_cur, u);
    bool ahead = false;                                   // the current segment's queries and first tile(s) were requested ahead
    int g = 0;                                            // stream tile index of the current segment's first tile (its parity picks the buffer)
    u32x4 frA[8], frB[8];
    for (;;) {
        unit_t u_next = 0;
        bool have_next = pc_cur.it.next(pc_cur.plan, pc_cur.grp, u_next);
        bool cross = false, restart = false;              // the next segment is the first of another pair (restart: A/B builds only)
        if (!have_next && open_pair(pc_cur.pq + 1, pc_nxt, u_next)) {
            if (NM_COARSE_CROSS) have_next = cross = true;
            else restart = true;
        }
        // the descriptors the NEXT segment's requests go through, by value (a reference picked at run time sent the
        // contexts to scratch memory and every request through a waterfall loop)
        const rsrc_t nA_rs = cross ? pc_nxt.rsA : pc_cur.rsA, nB_rs = cross ? pc_nxt.rsB : pc_cur.rsB;
        const rsrc_t nS_rs = cross ? pc_nxt.rsS : pc_cur.rsS, nN_rs = cross ? pc_nxt.rsN : pc_cur.rsN;
        Seg nxt = cur;
        if (have_next) nxt = cross ? locate(pc_nxt, u_next) : locate(pc_cur, u_next);
        const int ntiles = cur.ntiles, t0 = cur.t0;
        if (!ahead) {
            // slow path. Every wave is past the barrier of the previous stream tile (or this is the start): both tile buffers,
            // the slots and the query region are free
            dma_queries(pc_cur.rsA, cur.qb * QB);
            dma_norms(pc_cur.rsN, cur.qb);
            dma_tile(pc_cur.rsB, pc_cur.rsS, t0, g & 1);
            if (ntiles > 1) dma_tile(pc_cur.rsB, pc_cur.rsS, t0 + 1, (g + 1) & 1);    // stream tile g + 1
            // (LDS-DMA is tracked per wave: every wave drains ITS transfers before the barrier -- see match_top2_body)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f16_fetch<0>(frA, ldsb + (g & 1) * IMG + r * ROWB, foff);
        }
        const float nav = norm_read();                    // (requested ahead: landed before the previous segment's second barrier)
        // per-lane MFMA fragments of the wave's 32 queries (k-steps 0..7; prep_kernel applied the -2) and the norm k-slot
        u32x4 qw[16];
#pragma unroll
        for (int t = 0; t < 8; ++t) qw[t] = *reinterpret_cast<const u32x4 *>(ldsb + QREG + (wave * 32 + r) * ROWB + foff[t]);
        u32x4 qslot = {0u, 0u, 0u, 0u};
        if (h == 0) {
            unsigned nh, nm, nl;
            bf16_three(nav, nh, nm, nl);
            qslot = (u32x4){BF16_ONE | (BF16_ONE << 16), BF16_ONE | (nh << 16), nm | (nl << 16), 0u};
        }
        Top3 best;
        best.k1 = best.k2 = best.k3 = KEY_INF; best.t1 = best.t2 = 0;
        auto fold = [&](int g1, int g2, int g3, int tag) {
            if (__any(g1 < best.k3)) {
                top3_merge(best, g1, tag);
                top3_merge(best, g2, tag);
                top3_merge(best, g3, tag);
            }
        };
        bool next_ahead = false;
        f32x16 a0, a1, b0, b1;
        for (int n = 0; n < ntiles; ++n) {
            const int b = (g + n) & 1;
            const char *tb = ldsb + b * IMG + r * ROWB;
            const char *sp = ldsb + SLOT0 + b * SLOTB + r * 16;
            int g1 = KEY_INF, g2 = KEY_INF;               // of groups 2 n - 1 (slots 0..31) and 2 n (slots 32..63) together
            // Issue priority: the two waves of a SIMD do not share it evenly -- one of them runs almost at its solo speed and
            // then idles ~1 200 cycles at the tile's barrier, the other (the critical path) takes ~30 % longer through the same
            // instructions (tools/kcoarse_stamps.py). A wave is raised from the barrier to the end of the iteration (the
            // request of the next tile and the last half group) and lowered for the first three half groups: -1.0 to -1.4 % per
            // launch, same box (static priorities for half the waves: no change; alternating by tile parity: -0.7 %).
            if (NM_COARSE_PRIO) __builtin_amdgcn_s_setprio(NM_COARSE_PRIO_LO);
            NM_STAMP(0);
            f16_fetch<4>(frB, tb, foff);
            f16_slots(a0, a1, sp, qslot);
            if (n == 0) {
                f16_half<false, 0>(a0, a1, b0, frA, qw, g1, g2);
                NM_STAMP(1);
                f16_fetch<0>(frA, tb + 64 * ROWB, foff);
                f16_half<false, 4>(a0, a1, b1, frB, qw, g1, g2);
            } else {
                f16_half<true, 0>(a0, a1, b0, frA, qw, g1, g2);
                NM_STAMP(1);
                f16_fetch<0>(frA, tb + 64 * ROWB, foff);
                f16_half<true, 4>(a0, a1, b1, frB, qw, g1, g2);
            }
            NM_STAMP(2);
            f16_fetch<4>(frB, tb + 64 * ROWB, foff);
            f16_slots(b0, b1, sp + 64 * 16, qslot);
            f16_half<true, 0, 32>(b0, b1, a0, frA, qw, g1, g2);
            NM_STAMP(3);
            // stream tiles g + n + 1 (requested an iteration ago: landed by now) and g + n + 2 (to be requested)
            const bool in1 = n + 1 < ntiles, in2 = n + 2 < ntiles;
            const bool ex1 = in1 || have_next;
            bool ex2 = false;                                         // stream tile g + n + 2 exists and is requested in this iteration
            int tile2 = 0;
            if (ex1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                NM_STAMP(4);
                __syncthreads();
                if (NM_COARSE_PRIO) __builtin_amdgcn_s_setprio(NM_COARSE_PRIO_HI);
                NM_STAMP(5);
                const int k2 = n + 2 - ntiles;                        // index in the next segment when tile g + n + 2 lies there
                ex2 = in2 || (have_next && ntiles >= 2 && k2 < nxt.ntiles);
                tile2 = in2 ? t0 + n + 2 : nxt.t0 + k2;
                if (n == 0 && have_next && ntiles >= 2) {             // every wave holds its fragments: the query region is free
                    dma_queries(nA_rs, nxt.qb * QB);
                    dma_norms(nN_rs, nxt.qb);
                    next_ahead = true;
                }
                if (in1 || next_ahead) f16_fetch<0>(frA, ldsb + (b ^ 1) * IMG + r * ROWB, foff);
            }
            NM_STAMP(6);
            // (The request spread over the last half group, one piece behind each k-step, measured 1.5 % SLOWER than five
            // instructions back to back here: 651 against 642 us per 16-pair launch, same box.)
#ifndef NM_STUB_NODMA
            if (ex2) dma_tile(in2 ? pc_cur.rsB : nB_rs, in2 ? pc_cur.rsS : nS_rs, tile2, b);
#endif
            f16_half<true, 4, 32>(b0, b1, a1, frB, qw, g1, g2);
            fold(g1, g2, g2, n);
            NM_STAMP(7);
#ifdef NM_COARSE_STAMPS
            if (wg == 0 && stamp_it < 64 && lane == 0) {              // (the stores' cost falls into segment 7 -> 0')
#pragma unroll
                for (int k = 0; k < 8; ++k) nm_coarse_stamps[(wave * 64 + stamp_it) * 8 + k] = stamp_t[k];
            }
            ++stamp_it;
#endif
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(b0), "+v"(b1));
        select_half<COARSE_SLOT_BITS>(b0, b1, best, ntiles);       // group 2 ntiles - 1: slots 0..31 of "iteration" ntiles
        segment_publish<COARSE_SLOT_BITS>(best, pc_cur.plan, pc_cur.grp, cur.pc, cur.qbl, pc_cur.vg, cur.tt + ntiles == cur.Lc, t0 * TILE_C, h,
                        cur.qb * QB + wave * 32 + r, pc_cur.nA, pc_cur.plan.S, pc_cur.partial, pc_cur.partial3);
        if (!have_next) {
            if (!restart) break;
            __syncthreads();                              // NM_COARSE_CROSS=0: every pair starts with the slow path, as until round 5
            pc_cur = pc_nxt; cur = locate(pc_cur, u_next); ahead = false; g = 0;
            continue;
        }
        g += ntiles;
        ahead = next_ahead;
        cur = nxt;
        if (cross) pc_cur = pc_nxt;
    }
}

template <int SCR>
__global__ __launch_bounds__(512, 2) void match_top2_kernel(const float *__restrict__ A, int nA_arg,
                                                           const float *__restrict__ B, int nB_arg,
                                                           const float *__restrict__ na, const float *__restrict__ nb,
                                                           const uint4 *__restrict__ nbslot,
                                                           MatchPlan plan_arg, float4 *__restrict__ partial,
                                                           float *__restrict__ partial3,
                                                           const int *__restrict__ d_nA, const int *__restrict__ d_nB,
                                                           const MatchPlan *__restrict__ d_plan)
{
    match_top2_body<SCR>(A, nA_arg, B, nB_arg, na, nb, nbslot, plan_arg, partial, partial3, d_nA, d_nB, d_plan, (int)blockIdx.x);
}

// Round 6: the single-pass screens with ONE XCD PER PAIR, like the coarse pass: a launch covers `xcd` pairs of the call (first,
// first + 1, ...), pair first + (b mod xcd) on the workgroups b of that XCD with local index b / xcd, each under a one-group plan
// for n_cu / xcd workgroups (host-sized calls: plans.p; device-sized: the pair's d_plan, made for that geometry by nbmax_kernel).
struct MatchPlanPack { MatchPlan p[8]; };
template <int SCR>
__global__ __launch_bounds__(512, 2) void match_top2_group_kernel(MatchBatch bt, int first, int xcd, MatchPlanPack plans)
{
    const int k = (int)blockIdx.x % xcd;
    const MatchPair &c = bt.p[first + k];
    match_top2_body<SCR>(SCR ? reinterpret_cast<const float *>(c.As) : c.A, c.nA, SCR ? reinterpret_cast<const float *>(c.Bs) : c.B, c.nB,
                         c.na, c.nb, c.nbslot, plans.p[k], c.partial, c.partial3, c.d_nA, c.d_nB, c.d_plan, (int)blockIdx.x / xcd);
}

// Second pass of the two-stage screen, all pairs of a call in ONE launch (blockIdx.y = pair): the bf16x3 screen on the
// rows the coarse pass listed (fine_rows_kernel wrote their split images, norms and the work plan for their number).
// Usually a percent of the rows: most workgroups find plan.G below their index and leave.
__global__ __launch_bounds__(512, 2) void match_top2_rows_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.y];
    match_top2_body<1>(reinterpret_cast<const float *>(c.As), c.nA, reinterpret_cast<const float *>(c.Bs), c.nB, c.na2, c.nb,
                       c.nbslot, MatchPlan{}, c.partial, c.partial3, pair_f1_count(c), c.d_nB, pair_plan2(c), (int)blockIdx.x);
}

__device__ __forceinline__ float exact_dist(const float4 *__restrict__ a, const float4 *__restrict__ b)
{
    float4 x[DIM / 4], y[DIM / 4];
#pragma unroll
    for (int k = 0; k < DIM / 4; ++k) { x[k] = a[k]; y[k] = b[k]; }     // all 64 loads in flight together
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < DIM / 4; ++k) {
        float t;
        t = x[k].x - y[k].x; acc = __builtin_fmaf(t, t, acc);
        t = x[k].y - y[k].y; acc = __builtin_fmaf(t, t, acc);
        t = x[k].z - y[k].z; acc = __builtin_fmaf(t, t, acc);
        t = x[k].w - y[k].w; acc = __builtin_fmaf(t, t, acc);
    }
    return acc;
}

// Last step of the reference's scan (match.cu:91-116), given what the scan's comparisons make of a row: m1 = its smallest
// non-NaN distance at the LOWEST index idx (or NaN with idx 0 when the distance to candidate 0 is NaN: `current < NaN` is
// false for good), m2 = the smallest of the OTHER non-NaN distances, +inf when there is none. The scan starts min_2 at
// 2139095040.0f (the int 0x7f800000 converted, not +inf) and OVERWRITES it with the old minimum at every replacement
// (:97), so that initial value survives only while the minimum sits at candidate 0: clamp iff idx == 0 (idx < 0: the row
// has no distance below +inf at all; its ratio test fails either way).
// mode 0: ratio test -> result[i] (untouched when min2 <= 0); mode 1: emit the shard triple, min2 UNCLAMPED -- the merge
// over the shards applies the clamp, on the global index (idx here is already global).
__device__ __forceinline__ void emit_match(int i, float m1, int idx, float m2, int mode, float ambiguity,
                                           int *__restrict__ result, float *__restrict__ min1_out,
                                           int *__restrict__ idx_out, float *__restrict__ min2_out)
{
    if (mode == 1) { min1_out[i] = m1; idx_out[i] = idx; min2_out[i] = m2; return; }
    if (idx <= 0 && MIN2_INIT < m2) m2 = MIN2_INIT;
    if (m2 > 0) {
        const float q = m1 / m2;
        result[i] = (q < ambiguity) ? idx : -1;
    }
}

// Four lanes per query. Each lane scans a quarter of the query's 2*S partial candidates, the quad merges them into the
// 4 best by (approximate distance, index), every lane recomputes ONE of them EXACTLY (sum_k fma(t,t,acc), k ascending:
// all loads in flight together), and lane 0 of the quad forms the exact (min1, idx, min2).
// Proof obligation: every candidate NOT recomputed has an approximate distance >= rest (the minimum over the chunks'
// third-best values and everything that dropped out of a top-4 list). If rest is not safely above the exact min2
// (margin = bound on the MFMA formulation's error), the query is appended to the fallback list instead of being emitted.
//
// STAGE 0: the single-pass screens (fp32, bf16x3). Two-stage screen: STAGE 1 after the fp16 coarse pass -- the same proof
// with the coarse pass's error bound (below); an unproven row goes to the list of the bf16x3 pass instead of the exact
// fallback -- and STAGE 2 after the bf16x3 pass over the listed rows: partial lists, norms and the row count are in list
// order (r), everything else belongs to row i = f1_list[r].
template <int STAGE>
__device__ __forceinline__ void finalize_block(const MatchBatch &bt, const MatchPair &c, int block, int nA, int S)
{
    const float *__restrict__ A = c.A, *__restrict__ B = c.B;
    const int mode = c.mode, index_offset = c.index_offset;
    const float4 *__restrict__ partial = c.partial;
    const float *__restrict__ partial3 = c.partial3;
    const float *__restrict__ na = c.na;
    const float ambiguity = bt.ambiguity;
    int *__restrict__ result = c.result;
    float *__restrict__ min1_out = c.min1, *__restrict__ min2_out = c.min2;
    int *__restrict__ idx_out = c.idx1;
    int *__restrict__ fb_count = c.fb_count, *__restrict__ fb_list = c.fb_list;
    const int t = block * 256 + threadIdx.x;
    const int sub = t & 3;
    const bool live = (t >> 2) < nA;
    const int iq = live ? (t >> 2) : nA - 1;             // row of the partial lists
    const int i = (STAGE == 2) ? c.f1_list[iq] : iq;     // row of A / of the outputs
    float cd[4]; int ci[4];
    float rest = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; ++k) { cd[k] = __builtin_inff(); ci[k] = -1; }
    const int nB = pair_nB(c);
    auto insert = [&](float d, int j) {
        if (j < 0 || j >= nB) return;                // absent, or a row of the last tile's padding (bf16x3: finite "norm")
#pragma unroll
        for (int k = 0; k < 4; ++k) {                // sorted insertion by (approx distance, index)
            const bool lt = (d < cd[k]) || (d == cd[k] && j < ci[k]);
            if (lt) { const float td = cd[k]; const int tj = ci[k]; cd[k] = d; ci[k] = j; d = td; j = tj; }
        }
        if (j >= 0) rest = __builtin_fminf(rest, d);  // fell off the end of the list: not going to be recomputed
    };
    for (int s = sub; s < S; s += 4) {
        const float4 p = partial[(size_t)iq * S + s];
        rest = __builtin_fminf(rest, partial3[(size_t)iq * S + s]);
        insert(p.x, __float_as_int(p.y));
        insert(p.z, __float_as_int(p.w));
    }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {               // quad butterfly: afterwards all 4 lanes hold the same sorted top-4
        float od[4]; int oi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { od[k] = __shfl_xor(cd[k], m); oi[k] = __shfl_xor(ci[k], m); }
        rest = __builtin_fminf(rest, __shfl_xor(rest, m));
#pragma unroll
        for (int k = 0; k < 4; ++k) insert(od[k], oi[k]);
    }
    rest = __builtin_fminf(rest, __builtin_fminf(__shfl_xor(rest, 1), __shfl_xor(rest, 2)));
    rest = __builtin_fminf(rest, __builtin_fminf(__shfl_xor(rest, 1), __shfl_xor(rest, 2)));
    // (Round 3: staging the four candidate rows of every query through LDS -- half a wave per 512-byte row, whole cache lines,
    // then each lane walks its row at the conflict-free pitch -- was built and measured: 155 us per 16-pair call instead of
    // 146. The kernel moves ~500 MB of scattered 512-byte rows per call, ~3.3 TB/s out of L2 / Infinity Cache: it is bound by
    // that gather, not by the shape of its load instructions.)
    // Ratio test decided by the screen alone (mode 0: only result[i] is wanted). Let E(bn) bound |value(j) - d_ref(j)| for a
    // candidate of norm |b_j| <= bn (as in the proof below; + 2^-18 for the key truncation and gamma_130 for d_ref against d,
    // both relative to at most (sqrt na + bn)^2), and Et(X) := E(min(sqrt na + sqrt X, sqrt nb_max)): a candidate with
    // d_ref(j) <= X has |b_j| <= sqrt na + sqrt X (triangle inequality), so its value is <= X + Et(X). The two smallest
    // values v1 <= v2 sit at j1, j2, every other candidate has a value >= v2. Hence
    //   * every d_ref exceeds lo1 := v1 - Et(v1) (a candidate at or below it would have a value below v1), every d_ref
    //     but j1's exceeds lo2 := v2 - Et(v2), and min2 <= hi2 := max(v1 + E(|b_j1|), v2 + E(|b_j2|)) with the two
    //     candidates' own norms;
    //   * lo1 >= ambiguity hi2 (1 + 1e-5) with lo2 > 0 gives min1 / min2 >= ambiguity after its rounding -> -1 (the scan's
    //     clamp of min2 only lowers min2);
    //   * with hi1 := v1 + E(|b_j1|):  v2 > hi1 + Et(hi1) makes j1 the unique minimum (any other candidate at or below hi1
    //     would have a value below v2), min2 > lo2, and hi1 < ambiguity lo2 (1 - 1e-5) gives min1 / min2 < ambiguity -> j1
    //     (not taken for global index 0, where the scan's clamp of min2 could matter).
    // Most rows of a SIFT pair are decided here and never gather their candidates' 512-byte rows, which is what this kernel's
    // time was (3 KB per row; round 3). Any comparison with a NaN fails: the row takes the exact route.
    const float nai = na[i], nbm = c.nbmax[0];
    const bool in_domain = (STAGE == 1) ? (nai < F16_NORM_LIMIT && nbm < F16_NORM_LIMIT) : (nai < NORM_LIMIT && nbm < NORM_LIMIT);
    if (mode == 0 && in_domain && ci[1] >= 0) {
        const float sna = __builtin_sqrtf(nai), snb = __builtin_sqrtf(nbm);
        // (+ 2^-18 key truncation + gamma_130; the coarse pass's keys drop a sixth bit: 2^-17)
        const float coeff = (STAGE == 2 ? bt.err_coeff2 : bt.err_coeff) + (STAGE == 1 ? 1.6e-5f : 1.2e-5f);
        const float rai = (STAGE == 1) ? c.ra[i] : 0.f, rbm = (STAGE == 1) ? c.nbmax[1] : 0.f;
        auto E_of = [&](float bn, float rbj) {            // bn: upper bound of |b_j|, rbj: of its fp16 residual norm
            float e = coeff * ((sna + bn) * (sna + bn)) * 1.0001f + 1e-30f;
            if (STAGE == 1) e += 2.0f * (rai * (bn + rbj) + (sna * 1.000001f + rai) * rbj + rai * rbj) * 1.0001f;
            return e;
        };
        auto Et = [&](float X) {
            const float bn = __builtin_fminf(sna + __builtin_sqrtf(__builtin_fmaxf(X, 0.f)) * 1.00001f, snb);
            return E_of(bn, __builtin_fminf(rbm, 4.8829e-4f * bn + 7e-4f));
        };
        const float v1 = cd[0], v2 = cd[1];
        const float e1 = E_of(__builtin_sqrtf(c.nb[ci[0]]) * 1.000001f, (STAGE == 1) ? c.rb[ci[0]] : 0.f);
        const float e2 = E_of(__builtin_sqrtf(c.nb[ci[1]]) * 1.000001f, (STAGE == 1) ? c.rb[ci[1]] : 0.f);
        const float lo1 = v1 - Et(v1), lo2 = v2 - Et(v2), hi1 = v1 + e1, hi2 = __builtin_fmaxf(hi1, v2 + e2);
        int quick = 0;
        if (lo2 > 0.f && lo1 >= ambiguity * hi2 * 1.00001f) quick = 1;
        else if (lo2 > 0.f && v2 > hi1 + Et(hi1) && ambiguity > 0.f && hi1 < ambiguity * lo2 * 0.99999f && ci[0] + index_offset > 0) quick = 2;
        if (quick) {                                      // the same for the four lanes of the quad
            if (live && sub == 0) result[i] = (quick == 1) ? -1 : ci[0] + index_offset;
            return;
        }
    }
    const int mine = (sub == 0) ? ci[0] : (sub == 1) ? ci[1] : (sub == 2) ? ci[2] : ci[3];
    float d = 0.f;
    if (mine >= 0)
        d = exact_dist(reinterpret_cast<const float4 *>(A + (size_t)i * DIM),
                       reinterpret_cast<const float4 *>(B + (size_t)mine * DIM));
    float ed[4]; int ei[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ed[k] = __shfl(d, ((threadIdx.x & 63) & ~3) + k); ei[k] = ci[k]; }
    if (!live || sub != 0) return;
    if (STAGE == 1) {
        // outside the coarse pass's domain (|2 x| must stay inside the fp16 range: squared norms below F16_NORM_LIMIT; NaN and
        // inf fail the comparison too): the bf16x3 pass decides, or passes the row on to the exact scan
        if (!(nai < F16_NORM_LIMIT) || !(*c.nbmax < F16_NORM_LIMIT)) {
            const int pos = atomicAdd(pair_f1_count(c), 1);
            c.f1_list[pos] = i;
            return;
        }
    } else if (!(nai < NORM_LIMIT) || !(*c.nbmax < NORM_LIMIT)) {    // outside the screens' domain (NaN, inf, huge): exact scan
        const int pos = atomicAdd(fb_count, 1);
        fb_list[pos] = i;
        return;
    }
    // exact minimum at the lowest index and second smallest of the recomputed candidates (+inf: there is no second)
    float m1 = __builtin_inff(), m2 = __builtin_inff(); int idx = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ei[k] < 0) continue;
        if (ed[k] < m1 || (ed[k] == m1 && ei[k] < idx)) { m2 = m1; m1 = ed[k]; idx = ei[k]; }
        else if (ed[k] < m2) m2 = ed[k];
    }
    if (idx == 0x7fffffff) return;
    // Proof obligation ("matcher theorem"): every candidate j that was NOT recomputed must have a
    // reference distance d_ref(j) > m2. With u = 2^-24, gamma_n = n u / (1 - n u):
    //   * the MFMA value is a 130-step fma chain nb^ + na^ - 2 sum a_k b_k with 129 roundings, products exact:
    //     |d~ - (na^ + nb^ - 2 a.b)| <= gamma_129 (na^ + nb^ + 2 sum |a_k b_k|) <= gamma_129 (sqrt na + sqrt nb)^2 (1 + gamma_128)
    //   * the norms are 128-step fma chains of squares: |na^ - na| <= gamma_128 na, the same for nb
    //     => |d~(j) - d(j)| <= 2 gamma_129 (sqrt na + sqrt nb_j)^2 =: E_j
    //   * the reference chain sum fma(t,t,acc), t = fl(a_k - b_k), gives d_ref >= d (1 - gamma_130)
    //   * a key reports d~ with its low 5 mantissa bits replaced: |value - d~| < 2^-18 |d~|
    // Suppose d_ref(j) <= m2. Then d(j) <= M := m2 (1 + gamma_131), and sqrt nb_j <= min(sqrt na + sqrt M, sqrt nb_max)
    // (triangle inequality; nb_max = the largest candidate norm, nbmax_kernel), so with s := min(2 sqrt na + sqrt M,
    // sqrt na + sqrt nb_max):  value(j) <= (M + 2 gamma_129 s^2) (1 + 2^-18) =: bound, and rest <= value(j) <= bound.
    // Hence rest > bound proves the row; otherwise the row is re-scanned exactly. Deterministic for every input
    // (2 gamma_129 = 1.5378e-5; the constants below carry the slop of evaluating the bound itself in fp32).
    const float sna = __builtin_sqrtf(nai);
    const float sq = __builtin_fminf(2.0f * sna + __builtin_sqrtf(m2), sna + __builtin_sqrtf(*c.nbmax));
    float bound = (m2 * 1.00001f + (STAGE == 2 ? bt.err_coeff2 : bt.err_coeff) * (sq * sq)) * 1.00001f + 1e-30f;      // 1 + 2^-17 = 1.0000076
    if (STAGE == 1) {
        // Coarse pass: the MFMA chain sums the norms and the EXACT products of the fp16 images a_h (= -image / 2) and b_h,
        // so err_coeff sq^2 above bounds its distance from  na + nb - 2 a_h.b_h  (134 terms instead of 130: the coefficient
        // carries a factor 2 of slack), and   a.b - a_h.b_h = e_a.b_h + a_h.e_b + e_a.e_b   with e = x - x_h gives
        //   |d~(j) - d(j)| <= err_coeff sq^2 + 2 (ra |b_h| + |a_h| rb_j + ra rb_j),   ra = |e_a|, rb_j = |e_b_j|  (Cauchy-Schwarz),
        // ra and rb_j being the upper bounds prep_kernel<2> computed from the images themselves. For a candidate with
        // d_ref(j) <= m2:  |b_j| <= bn := sq - sqrt na  (the triangle inequality / nb_max, as above),  |b_h| <= bn + rb_j,
        // |a_h| <= sqrt na + ra,  and  rb_j <= min(rb_max, 2^-11 bn + 7e-4): round-to-nearest fp16 loses at most 2^-11
        // relative per element, 2^-14 absolute per element of the 128 below the normal range (subnormal or flushed alike).
        const float bn = (sq - sna) * 1.000001f;
        const float rbj = __builtin_fminf(c.nbmax[1], 4.8829e-4f * bn + 7e-4f);
        const float rai = c.ra[i];
        const float e16 = 2.0f * (rai * (bn + rbj) + (sna * 1.000001f + rai) * rbj + rai * rbj);
        bound = (bound + e16 * 1.00001f) * 1.00001f;
    }
    if (!(rest > bound) && rest < __builtin_inff()) {    // a NaN bound (norms at the edge of the domain) proves nothing
        if (STAGE == 1) {
            const int pos = atomicAdd(pair_f1_count(c), 1);
            c.f1_list[pos] = i;
        } else {
            const int pos = atomicAdd(fb_count, 1);
            fb_list[pos] = i;
        }
        return;
    }
    emit_match(i, m1, idx + index_offset, m2, mode, ambiguity, result, min1_out, idx_out, min2_out);
}

// 64 rows per 256-thread block; the blocks of a pair stride over its rows (STAGE 0 / 1: the grid covers the batch's largest
// set or the capacity, one block per workgroup; STAGE 2: a few workgroups per pair walk the short list)
template <int STAGE>
__global__ __launch_bounds__(256) void match_finalize_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.y];
    if (pair_nA(c) <= 0 || pair_nB(c) <= 0) return;      // device-sized call with an empty set: nothing was screened
    const int nA = (STAGE == 2) ? min(max(*pair_f1_count(c), 0), pair_nA(c)) : pair_nA(c);
    const int S = (STAGE == 2) ? pair_plan2(c)->S : pair_S(c);
    for (int block = blockIdx.x; block * 64 < nA; block += gridDim.x) finalize_block<STAGE>(bt, c, block, nA, S);
}

// Two-stage screen, between its passes: the rows the coarse pass listed get what the bf16x3 kernel reads -- split images
// (scaled by -2) and norms in LIST order -- and the first wave of every pair makes the work plan for their number (the
// same make_plan_on as everywhere). Half a wave per listed row, as in prep_kernel.
__global__ __launch_bounds__(256) void fine_rows_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.y];
    const int nA = pair_nA(c), nB = pair_nB(c), lane = threadIdx.x & 63, k4 = lane & 31;
    const int count = (nA > 0 && nB > 0) ? min(max(*pair_f1_count(c), 0), nA) : 0;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const MatchPlan p = make_plan_on(count, nB, bt.n_cu2, bt.n_xcd, lane, 64, [](int v) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
            return v;
        });
        MatchPlan q = p;
        if (count == 0) q.G = 0;                          // nothing listed: every workgroup of the second pass leaves at once
        if (lane == 0) *pair_plan2(c) = q;
    }
    for (int blk = blockIdx.x; blk * PREP_ROWS < count; blk += gridDim.x)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = blk * PREP_ROWS + (threadIdx.x >> 6) * 4 + (lane >> 5) + 2 * q;
        if (r >= count) continue;
        const int i = c.f1_list[r];
        const float4 x = reinterpret_cast<const float4 *>(c.A + (size_t)i * DIM)[k4];
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        bf16_split(-2.0f * x.x, h0, l0); bf16_split(-2.0f * x.y, h1, l1);
        bf16_split(-2.0f * x.z, h2, l2); bf16_split(-2.0f * x.w, h3, l3);
        unsigned *dst = c.As + (size_t)r * DIM;
        reinterpret_cast<uint2 *>(dst)[k4] = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        reinterpret_cast<uint2 *>(dst + 64)[k4] = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
        if (k4 == 0) c.na2[r] = c.na[i];
    }
}

// Exact squared distances, the reference's own arithmetic (match.cu:36-42): for every (row vector x, column vector y)
//   acc = 0;  for k = 0..127:  t = x_k - y_k;  acc = fma(t, t, acc)
// (fl(x - y) = -fl(y - x) exactly, so which operand is the query does not matter). This is VALU work by nature -- the
// difference has to be formed per pair, so it is not a contraction an MFMA could take -- and its floor is two VALU
// operations per (pair, k): 2 * 128 * rows * cols / 78.6e12 lane-ops/s = 481 us at 12k x 12k. A workgroup owns a 128 x 128
// tile of the output, a thread an 8 x 8 register tile; the operands stream through LDS k-major in chunks of 32 k (two
// b128 reads per operand feed 128 VALU operations: 1 LDS read per 32 VALU instead of the 17 per 32 of the round-1 kernel).
// X: row-side set (nX vectors), Y: column-side set. Element (v, k) is S[v * 128 + k], or S[k * n + v] when K_MAJOR.
// out[row * ld + col].
constexpr int XD_TILE = 128, XD_KC = 32, XD_PITCH = XD_TILE + 4;

template <bool K_MAJOR>
__device__ __forceinline__ void xd_stage(float *__restrict__ dst, const float *__restrict__ S, int n, int v0, int kc, int tid)
{
    if (K_MAJOR) {           // S[k][v]: rows of the LDS image are contiguous in memory
#pragma unroll
        for (int it = 0; it < (XD_KC * XD_TILE) / 256; ++it) {
            const int e = tid + 256 * it;
            const int k = e >> 7, v = e & 127;
            dst[k * XD_PITCH + v] = (v0 + v < n) ? S[(size_t)(kc + k) * n + v0 + v] : 0.f;
        }
    } else {                 // S[v][k]: 8 lanes read the 32 k of one vector (128 contiguous bytes), scattered k-major into LDS
#pragma unroll
        for (int it = 0; it < (XD_KC * XD_TILE) / (256 * 4); ++it) {
            const int e = tid + 256 * it;
            const int v = e >> 3, k4 = (e & 7) * 4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v0 + v < n) x = *reinterpret_cast<const float4 *>(S + (size_t)(v0 + v) * DIM + kc + k4);
            dst[(k4 + 0) * XD_PITCH + v] = x.x; dst[(k4 + 1) * XD_PITCH + v] = x.y;
            dst[(k4 + 2) * XD_PITCH + v] = x.z; dst[(k4 + 3) * XD_PITCH + v] = x.w;
        }
    }
}

// Exact fallback for the queries the finalize pass could not prove: the listed rows are re-scanned against ALL candidates
// with the reference's arithmetic, as a tiled exact-distance computation (128 listed rows x 128 candidates per step, 8 x 8
// register tiles as in exact_distance_kernel) that keeps only the running (min1, lowest index, second minimum) of every
// row. A workgroup owns one 128-row chunk of the list and one of FB_SPLIT contiguous candidate slices; slice results go to
// `part` and match_fallback_merge_kernel combines them in ascending slice order. A candidate row is read once per chunk,
// not once per listed row (the round-1 kernel scanned row by row: 9 ms for the 1.5 % of rows listed at 100k x 100k).
constexpr int FB_SPLIT = 64;
constexpr int FB_CHUNKS = 16;          // chunk loops in flight (gridDim.y); a workgroup strides over the chunks
constexpr int FB_ROWWISE_MAX = 24;     // up to this many listed rows: one pass per row instead of 128-row tiles

__device__ __forceinline__ void top2_merge(float &m1, int &i1, float &m2, float o1, int oi, float o2)
{
    const bool take = (o1 < m1) || (o1 == m1 && oi < i1);
    const float lo = take ? o1 : m1, hi = take ? m1 : o1;
    const float s2 = take ? o2 : m2;
    i1 = take ? oi : i1;
    m1 = lo;
    m2 = (hi < s2) ? hi : s2;
}

__global__ __launch_bounds__(256) void match_fallback_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.z];
    const float *__restrict__ A = c.A, *__restrict__ B = c.B;
    const int nB = pair_nB(c);
    const int *__restrict__ fb_list = c.fb_list;
    float4 *__restrict__ part = c.partial;
    __shared__ __attribute__((aligned(16))) float sXY[2 * XD_KC * XD_PITCH];
    float *const sX = sXY, *const sY = sXY + XD_KC * XD_PITCH;
    const int count = *c.fb_count;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int slice = nm_divup_dev(nB, FB_SPLIT);
    const int j0 = blockIdx.x * slice, j1 = min(j0 + slice, nB);
    if (count <= FB_ROWWISE_MAX) {
        // A handful of rows (the usual case: 0-10 of 12k on SIFT descriptors): a 128-row tile would be almost empty. The
        // listed rows wait in LDS; the slice's candidates pass through LDS 64 at a time (coalesced loads -- a lane that
        // streams its own 512-byte candidate row touches 64 cache lines per load instruction, which made this path
        // ~25 us per listed row); thread (candidate = tid % 64, wave w) then holds its candidate in registers and runs
        // the exact chain against the rows w, w + 4, ... (broadcast LDS reads). One workgroup per slice (blockIdx.y = 0).
        constexpr int RW = (FB_ROWWISE_MAX + 3) / 4;          // rows per wave
        __shared__ __attribute__((aligned(16))) float sQ[FB_ROWWISE_MAX * DIM];
        static_assert(64 * KP <= 2 * XD_KC * XD_PITCH, "64 candidates at pitch KP fit the tile buffers");
        if (blockIdx.y != 0 || count <= 0) return;
        for (int q = tid; q < count * (DIM / 4); q += 256) {
            const int e = q >> 5, k4 = q & 31;
            reinterpret_cast<float4 *>(sQ)[q] = reinterpret_cast<const float4 *>(A + (size_t)fb_list[e] * DIM)[k4];
        }
        const int cand = tid & 63, w = tid >> 6;
        float m1[RW], m2[RW]; int i1[RW];
#pragma unroll
        for (int s2 = 0; s2 < RW; ++s2) { m1[s2] = __builtin_inff(); m2[s2] = __builtin_inff(); i1[s2] = 0x7fffffff; }
        for (int c0 = j0; c0 < j1; c0 += 64) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (tid >> 5) + 8 * it, k4 = tid & 31;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + row < j1) v = reinterpret_cast<const float4 *>(B + (size_t)(c0 + row) * DIM)[k4];
                *reinterpret_cast<float4 *>(&sXY[row * KP + 4 * k4]) = v;
            }
            __syncthreads();
            const int j = c0 + cand;
            float4 y[DIM / 4];
#pragma unroll
            for (int k = 0; k < DIM / 4; ++k) y[k] = *reinterpret_cast<const float4 *>(&sXY[cand * KP + 4 * k]);
#pragma unroll
            for (int s2 = 0; s2 < RW; ++s2) {
                const int e = w + 4 * s2;
                if (e < count) {                                 // uniform per wave
                    const float4 *x = reinterpret_cast<const float4 *>(sQ + e * DIM);
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < DIM / 4; ++k) {
                        const float4 xv = x[k];
                        float tt;
                        tt = xv.x - y[k].x; acc = __builtin_fmaf(tt, tt, acc);
                        tt = xv.y - y[k].y; acc = __builtin_fmaf(tt, tt, acc);
                        tt = xv.z - y[k].z; acc = __builtin_fmaf(tt, tt, acc);
                        tt = xv.w - y[k].w; acc = __builtin_fmaf(tt, tt, acc);
                    }
                    if (j < j1) {
                        if (acc < m1[s2]) { m2[s2] = m1[s2]; m1[s2] = acc; i1[s2] = j; }
                        else if (acc < m2[s2]) m2[s2] = acc;
                    }
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < RW; ++s2) {
            const int e = w + 4 * s2;
            if (e >= count) continue;
            float a1 = m1[s2], a2 = m2[s2]; int ai = i1[s2];
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const float o1 = __shfl_xor(a1, sft), o2 = __shfl_xor(a2, sft);
                const int oi = __shfl_xor(ai, sft);
                top2_merge(a1, ai, a2, o1, oi, o2);
            }
            if (cand == 0) part[(size_t)e * FB_SPLIT + blockIdx.x] = make_float4(a1, __int_as_float(ai), a2, 0.f);
        }
        return;
    }
    for (int e0 = blockIdx.y * XD_TILE; e0 < count; e0 += gridDim.y * XD_TILE) {
        float m1[8], m2[8]; int i1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { m1[i] = __builtin_inff(); m2[i] = __builtin_inff(); i1[i] = 0x7fffffff; }
        for (int c0 = j0; c0 < j1; c0 += XD_TILE) {
            float acc[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
            for (int kc = 0; kc < DIM; kc += XD_KC) {
                __syncthreads();
#pragma unroll
                for (int it = 0; it < (XD_KC * XD_TILE) / (256 * 4); ++it) {       // listed query rows, gathered
                    const int e = tid + 256 * it;
                    const int v = e >> 3, k4 = (e & 7) * 4;
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e0 + v < count) x = *reinterpret_cast<const float4 *>(A + (size_t)fb_list[e0 + v] * DIM + kc + k4);
                    sX[(k4 + 0) * XD_PITCH + v] = x.x; sX[(k4 + 1) * XD_PITCH + v] = x.y;
                    sX[(k4 + 2) * XD_PITCH + v] = x.z; sX[(k4 + 3) * XD_PITCH + v] = x.w;
                }
                xd_stage<false>(sY, B, j1, c0, kc, tid);
                __syncthreads();
#pragma unroll 4
                for (int k = 0; k < XD_KC; ++k) {
                    const float4 xa = *reinterpret_cast<const float4 *>(&sX[k * XD_PITCH + ty * 8]);
                    const float4 xb = *reinterpret_cast<const float4 *>(&sX[k * XD_PITCH + ty * 8 + 4]);
                    const float4 ya = *reinterpret_cast<const float4 *>(&sY[k * XD_PITCH + tx * 8]);
                    const float4 yb = *reinterpret_cast<const float4 *>(&sY[k * XD_PITCH + tx * 8 + 4]);
                    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                    const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float t = xv[i] - yv[j];
                            acc[i][j] = __builtin_fmaf(t, t, acc[i][j]);
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {                       // ascending candidate index within the thread
                const int col = c0 + tx * 8 + j;
                if (col < j1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float d = acc[i][j];
                        if (d < m1[i]) { m2[i] = m1[i]; m1[i] = d; i1[i] = col; }
                        else if (d < m2[i]) m2[i] = d;
                    }
                }
            }
        }
        // the 16 threads that share a row: butterfly over tx (lanes of one wave), index breaks ties
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                const float o1 = __shfl_xor(m1[i], sft), o2 = __shfl_xor(m2[i], sft);
                const int oi = __shfl_xor(i1[i], sft);
                top2_merge(m1[i], i1[i], m2[i], o1, oi, o2);
            }
            const int e = e0 + ty * 8 + i;
            if (tx == 0 && e < count) part[(size_t)e * FB_SPLIT + blockIdx.x] = make_float4(m1[i], __int_as_float(i1[i]), m2[i], 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void match_fallback_merge_kernel(MatchBatch bt)
{
    const MatchPair &c = bt.p[blockIdx.y];
    const int *__restrict__ fb_count = c.fb_count, *__restrict__ fb_list = c.fb_list;
    const float4 *__restrict__ part = c.partial;
    const int mode = c.mode, index_offset = c.index_offset;
    const float ambiguity = bt.ambiguity;
    int *__restrict__ result = c.result;
    float *__restrict__ min1_out = c.min1, *__restrict__ min2_out = c.min2;
    int *__restrict__ idx_out = c.idx1;
    const int count = *fb_count;
    const bool holds0 = (mode == 0) || (index_offset == 0);      // this call's candidate 0 is the scan's candidate 0
    for (int e = blockIdx.x * 256 + threadIdx.x; e < count; e += gridDim.x * 256) {
        // the slices left NaN distances out and never record +inf (strict < from +inf): the scan's view of candidates >= 1
        float m1 = __builtin_inff(), m2 = __builtin_inff(); int i1 = 0x7fffffff;
        for (int sl = 0; sl < FB_SPLIT; ++sl) {
            const float4 p = part[(size_t)e * FB_SPLIT + sl];
            top2_merge(m1, i1, m2, p.x, __float_as_int(p.y), p.z);
        }
        const int row = fb_list[e];
        int idx = (i1 == 0x7fffffff) ? (holds0 ? 0 : -1) : i1 + index_offset;
        if (holds0) {
            // match.cu:90: the scan STARTS from the distance to candidate 0; a NaN there is never replaced
            const float d0 = exact_dist(reinterpret_cast<const float4 *>(c.A + (size_t)row * DIM), reinterpret_cast<const float4 *>(c.B));
            if (d0 != d0) { m2 = m1; m1 = d0; idx = 0; }
        }
        emit_match(row, m1, idx, m2, mode, ambiguity, result, min1_out, idx_out, min2_out);
    }
}

// Multi-GPU merge: shard-major triples, ascending shard order, strict < so the lowest global index wins ties.
__global__ __launch_bounds__(256) void match_merge_kernel(const float *__restrict__ min1, const int *__restrict__ idx1,
                                                         const float *__restrict__ min2, int n_shards, int nA,
                                                         float ambiguity, int *__restrict__ result)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA) return;
    float m1 = __builtin_inff(), m2 = __builtin_inff(); int idx = -1;             // the neutral (empty shard) triple
    for (int g = 0; g < n_shards; ++g) {
        const float a1 = min1[(size_t)g * nA + i], a2 = min2[(size_t)g * nA + i];
        const int ai = idx1[(size_t)g * nA + i];
        // a NaN minimum can only come from the shard holding global candidate 0 (every shard before it is empty): it is
        // the scan's min_1_distance for good (match.cu:90,96)
        if (a1 != a1) { m1 = a1; idx = 0; m2 = a2; }
        else if (a1 < m1) { m2 = (m1 < a2) ? m1 : a2; m1 = a1; idx = ai; }
        else if (a1 < m2) m2 = a1;
    }
    emit_match(i, m1, idx, m2, 0, ambiguity, result, nullptr, nullptr, nullptr);     // clamps iff the minimum sits at 0
}

// An empty candidate shard (world > nB, or uneven tiny sets): the neutral triple, so that the merge ignores the shard.
__global__ __launch_bounds__(256) void shard_neutral_kernel(float *__restrict__ min1, int *__restrict__ idx1,
                                                           float *__restrict__ min2, int nA)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA) return;
    min1[i] = __builtin_inff(); idx1[i] = -1; min2[i] = __builtin_inff();
}

// The same merge on the buffer an all-gather of per-rank (min1[nA], idx1[nA], min2[nA]) blocks produces: element c of row i
// of rank g sits at packed[(g * 3 + c) * nA + i].
__global__ __launch_bounds__(256) void match_merge_packed_kernel(const int *__restrict__ packed, int n_shards, int nA,
                                                                float ambiguity, int *__restrict__ result)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA) return;
    float m1 = __builtin_inff(), m2 = __builtin_inff(); int idx = -1;
    for (int g = 0; g < n_shards; ++g) {
        const int *p = packed + (size_t)g * 3 * nA;
        const float a1 = __int_as_float(p[i]), a2 = __int_as_float(p[2 * (size_t)nA + i]);
        const int ai = p[(size_t)nA + i];
        if (a1 != a1) { m1 = a1; idx = 0; m2 = a2; }
        else if (a1 < m1) { m2 = (m1 < a2) ? m1 : a2; m1 = a1; idx = ai; }
        else if (a1 < m2) m2 = a1;
    }
    emit_match(i, m1, idx, m2, 0, ambiguity, result, nullptr, nullptr, nullptr);
}

// ---- exact API building blocks ----
__global__ __launch_bounds__(256) void transpose_kernel(float *__restrict__ odata, const float *__restrict__ idata,
                                                       int width, int height)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    int x = blockIdx.x * 32 + tx;
    for (int k = 0; k < 32; k += 8) {
        const int y = blockIdx.y * 32 + ty + k;
        if (x < width && y < height) tile[ty + k][tx] = idata[(size_t)y * width + x];
    }
    __syncthreads();
    x = blockIdx.y * 32 + tx;                                     // output column = input row
    for (int k = 0; k < 32; k += 8) {
        const int y = blockIdx.x * 32 + ty + k;                   // output row = input column
        if (x < height && y < width) odata[(size_t)y * height + x] = tile[tx][ty + k];
    }
}

// t = x - y and acc = fma(t, t, acc) for TWO column elements per instruction (v_pk_add_f32 with the x element broadcast by
// op_sel and y negated by neg_lo / neg_hi, then v_pk_fma_f32): the same IEEE operations per element as the scalar pair, in
// the same order -- half the VALU instructions, so one wave alone keeps its SIMD's vector pipe busy (a wave issues one
// vector instruction per ~4 cycles whatever its width).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xd_step(v2f &acc, v2f xpair, v2f y, bool hi)
{
    v2f t;
    if (hi) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(xpair), "v"(y));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(xpair), "v"(y));
    asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(t));
}

template <bool X_KMAJOR, bool Y_KMAJOR>
__global__ __launch_bounds__(256) void exact_distance_kernel(const float *__restrict__ X, int nX,
                                                            const float *__restrict__ Y, int nY,
                                                            float *__restrict__ out, size_t ld,
                                                            const int *__restrict__ only_if_above, int threshold)
{
    // (behind the MFMA distance pass: runs only when that pass's list of uncovered entries overflowed -- uniform scalar load)
    if (only_if_above && *only_if_above <= threshold) return;
    __shared__ __attribute__((aligned(16))) float sX[XD_KC * XD_PITCH];
    __shared__ __attribute__((aligned(16))) float sY[XD_KC * XD_PITCH];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;                 // 16 x 16 threads, 8 x 8 outputs each
    const int c0 = blockIdx.x * XD_TILE, r0 = blockIdx.y * XD_TILE;
    v2f acc[8][4];                                          // [row][column pair]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v2f){0.f, 0.f};
    for (int kc = 0; kc < DIM; kc += XD_KC) {
        __syncthreads();
        xd_stage<X_KMAJOR>(sX, X, nX, r0, kc, tid);
        xd_stage<Y_KMAJOR>(sY, Y, nY, c0, kc, tid);
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < XD_KC; ++k) {
            const float4 xa = *reinterpret_cast<const float4 *>(&sX[k * XD_PITCH + ty * 8]);
            const float4 xb = *reinterpret_cast<const float4 *>(&sX[k * XD_PITCH + ty * 8 + 4]);
            const float4 ya = *reinterpret_cast<const float4 *>(&sY[k * XD_PITCH + tx * 8]);
            const float4 yb = *reinterpret_cast<const float4 *>(&sY[k * XD_PITCH + tx * 8 + 4]);
            const v2f xp[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}};
            const v2f yp[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xd_step(acc[i][j], xp[i >> 1], yp[j], (i & 1) != 0);
        }
    }
    const int col = c0 + tx * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = r0 + ty * 8 + i;
        if (row >= nX) continue;
        float *o = out + (size_t)row * ld + col;
        if (col + 7 < nY && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
            *reinterpret_cast<float4 *>(o) = make_float4(acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[i][2].x, acc[i][2].y, acc[i][3].x, acc[i][3].y);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col + j < nY) o[j] = (j & 1) ? acc[i][j >> 1].y : acc[i][j >> 1].x;
        }
    }
}

// ---- the materialised distance matrix on the fp32 MFMA (round 5; reference: kernels/match.cu:14-80, siftfunctions.cu:28-34) ----
// D[i][j] = |x_i - y_j|^2 as |x'_i|^2 + |y'_j|^2 - 2 x'_i . y'_j on v_mfma_f32_32x32x2_f32 with the store fused into the
// kernel, where x' = x - mu, y' = y - mu are the rows CENTRED on the mean row of Y: distances are translation-invariant, and
// the contraction's rounding error scales with the norms of what is multiplied, not with the distance (rows that share a large
// common component -- an offset, a dominant mean descriptor -- would otherwise all fail the test below).
// Every entry is within 1e-4 relative of the reference's chain (acc = fma(t, t, acc), t = x_k - y_k, match.cu:36-42):
//   * the value v is the sum of TWO accumulator chains (k = 0..63 with the norm pair, k = 64..127): no term passes through more
//     than 67 roundings (2 + 64 steps of a chain, the final add), and the tabulated norms carry 8 (distance_center_kernel), so
//     with d' the exact distance of the centred rows  |v - d'| <= (gamma_67 + gamma_8) (sqrt nx + sqrt ny)^2 = 4.47e-6 (..)^2;
//     DIST_C = 5.15e-6 is that with 15 % of slack. (Premise, as for the matcher's fp32 screen: an instruction is two fused
//     steps, C + a0 b0 + a1 b1 with at most two roundings -- nm_selftest_mfma_model(f32) measures it.)
//   * centring rounds once per element, |x'_k - (x_k - mu_k)| <= u |x'_k|: sqrt d differs from sqrt d' by at most
//     u (sqrt nx + sqrt ny);
//   * the reference's own chain is within gamma_130 d <= 7.8e-6 d of the true d.
//   An entry is ACCEPTED iff v >= K (sqrt nx + sqrt ny)^2, K = DIST_C / 7.8e-5 = 0.066: then |v - d'| <= 7.8e-5 v, the centring
//   moves d by < 4.7e-7 d', and |v - d_ref| < (7.8e-5 + 4.7e-7 + 7.8e-6 + products) d_ref < 8.7e-5 d_ref.
//   Every other entry -- near-duplicates, exact copies, negative or NaN values from cancellation or overflow -- belongs to a
//   32 x 32 block that is LISTED; distance_fixup_kernel re-applies the test to the block's stored values and recomputes what
//   fails it by the reference's own chain: bit-equal to the exact kernel. If the list overflows, the whole matrix is
//   (exact_distance_kernel launched behind it runs only then). Uncorrelated rows have d' ~ nx + ny >= (sqrt nx + sqrt ny)^2 / 2;
//   only pairs whose centred rows correlate above ~0.87 fail (1.7e-4 of the pairs of two 1080p frames' descriptors, 5 % of the
//   blocks). The test is v_add + v_mul + v_cmp per entry (sqrt(K n) is tabulated, rounded up).
// Layout: rows of the MFMA result = the STREAMED set X (128-row tiles through LDS, registers), columns = the RESIDENT set Y
// (32 rows per wave, fragments in VGPRs, lanes): a store instruction writes 128 contiguous bytes of D per half wave.
struct DistArgs {
    const float *Xc, *Yc;         // centred rows (n x 128)
    const float *nx, *ny;         // their squared norms
    const float *tx, *ty;         // sqrt(K norm), rounded up: the acceptance test is v >= (tx_i + ty_j)^2
    float *D; size_t ldd;         // D[i * ldd + j], i over X, j over Y
    int nX, nY;
    int *fix_count; int2 *fix_list; int fix_cap;
    int *fix_report;              // where the fix-up pass leaves the number of listed blocks (the list's memory is reused)
};
constexpr int DIST_P = 64;        // row slices of the column-sum pass
constexpr float DIST_C = 5.15e-6f;                    // |v - d'| <= DIST_C (sqrt nx + sqrt ny)^2
constexpr float DIST_K = DIST_C / 7.8e-5f;            // acceptance: v >= DIST_K (sqrt nx + sqrt ny)^2

// column sums of Y in DIST_P slices: part[slice][k] = sum of y[r][k] over rows r = slice, slice + DIST_P, ... (fixed order)
__global__ __launch_bounds__(128) void distance_colsum_kernel(const float *__restrict__ Y, int nY, float *__restrict__ part)
{
    const int k = threadIdx.x, sl = blockIdx.x;
    float s = 0.f;
    for (int r = sl; r < nY; r += DIST_P) s += Y[(size_t)r * DIM + k];
    part[sl * DIM + k] = s;
}

// blockIdx.y = 0: X, 1: Y. One wave per row: x' = x - mu, |x'|^2 (fma pair + wave tree: 8 roundings), sqrt(K |x'|^2) rounded up.
__global__ __launch_bounds__(256) void distance_center_kernel(const float *__restrict__ X, int nX, const float *__restrict__ Y, int nY,
                                                             const float *__restrict__ part, float *__restrict__ Xc,
                                                             float *__restrict__ Yc, float *__restrict__ nx, float *__restrict__ ny,
                                                             float *__restrict__ tx, float *__restrict__ ty, int *__restrict__ fix_count)
{
    __shared__ float mu[DIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < DIM) {
        float s = 0.f;
        for (int p = 0; p < DIST_P; ++p) s += part[p * DIM + tid];
        mu[tid] = s / (float)nY;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *fix_count = 0;
    __syncthreads();
    const bool second = blockIdx.y != 0;
    const float *S = second ? Y : X;
    float *Sc = second ? Yc : Xc, *nn = second ? ny : nx, *tt = second ? ty : tx;
    const int n = second ? nY : nX;
    const float2 m2 = *reinterpret_cast<const float2 *>(&mu[2 * lane]);
    for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
        const float2 v = *reinterpret_cast<const float2 *>(S + (size_t)r * DIM + 2 * lane);
        const float2 c = make_float2(v.x - m2.x, v.y - m2.y);
        *reinterpret_cast<float2 *>(Sc + (size_t)r * DIM + 2 * lane) = c;
        float q = __builtin_fmaf(c.x, c.x, c.y * c.y);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) q += __shfl_xor(q, d);
        if (lane == 0) { nn[r] = q; tt[r] = __builtin_sqrtf(q * DIST_K) * 1.000002f; }
    }
}

// Epilogue of one 32 x 32 accumulator: register e holds tile row rb + 8 (e / 4) + 4 h + (e % 4), column = this lane's j.
// Every value is stored. If ANY value of the 32 x 32 block is outside the bound, the BLOCK is listed (one atomic by one lane):
// distance_fixup_kernel reads the block's stored values back, applies the same test to the same values and recomputes exactly
// what fails it. FULL: all 128 rows of the tile exist (only the last row tile of X is ragged; its missing rows are masked).
template <bool FULL>
__device__ __forceinline__ void dist_emit(const f32x16 &acc, const DistArgs &a, __amdgpu_buffer_rsrc_t rsD, const float *tab,
                                          int rb, int h, unsigned long long jmask, float tyj, unsigned vo, unsigned ld4,
                                          int i_blk, int j_blk, int rows_here)
{
    // vo: this lane's byte offset inside the tile's rows of D, 0x80000000 for a lane whose column does not exist -- beyond any
    // tile's byte range (< 2^31), so the descriptor's range check drops that lane's stores whatever soffset adds (the probe
    // tools/micro/buffer_range.hip shows voffset + soffset checked on this device). jmask = lanes with a column.
    unsigned long long listed = 0;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const float4 t4 = *reinterpret_cast<const float4 *>(tab + rb + 8 * e4 + 4 * h);
        const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
            const int e = 4 * e4 + e1;
            // lanes whose value is NOT >= the acceptance term (unordered-or-less-than: a NaN is listed), as a wave mask
            const float sq = tv[e1] + tyj;
            const unsigned long long below = __builtin_amdgcn_fcmpf(acc[e], sq * sq, 12 /* FCMP_ULT */);
            if (FULL) {
                listed |= below;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e]), rsD, vo, (rb + 8 * e4 + e1) * ld4, 0);
            } else {                                        // last row tile of X: rows >= rows_here do not exist
                const bool ok = rb + 8 * e4 + 4 * h + e1 < rows_here;
                listed |= below & __ballot(ok);
                if (ok) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e]), rsD, vo, (rb + 8 * e4 + e1) * ld4, 0);
            }
        }
    }
    if (__builtin_expect((listed & jmask) != 0, 0)) {       // near-duplicates, cancellation, non-finite values
        if ((threadIdx.x & 63) == 0) {
            const int at = atomicAdd(a.fix_count, 1);
            if (at < a.fix_cap) a.fix_list[at] = make_int2(i_blk, j_blk);
        }
    }
}

__global__ __launch_bounds__(512, 1) void distance_mfma_kernel(DistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const tabuf = lds + 2 * TILE_C * KP;             // 2 x 128 acceptance terms of the streamed rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int srow = tid >> 5, scol = (tid & 31) * 4;       // staging coordinates: 16 rows x 32 float4 per pass
    const int nX = a.nX, nY = a.nY;
    const int n_rt = nm_divup_dev(nX, TILE_C), n_cb = nm_divup_dev(nY, QB);
    const long long U = (long long)n_rt * n_cb;
    const long long u0 = U * blockIdx.x / gridDim.x, u1 = U * (blockIdx.x + 1) / gridDim.x;
    if (u0 >= u1) return;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Xc), 0, nX * (DIM * 4), 0x00020000);
    const int voff = (srow * DIM + scol) * 4;

    u32x4 st[8];
    float stn = 0.f, stt = 0.f;
    auto stage_load = [&](int rt) {                         // rows rt * 128 .. + 127 of X -> registers (rows >= nX read as zeros)
        const int ib = rt * TILE_C;
        // (the row offset rides in voffset, which every reading of the ISA includes in the descriptor's range check; on this
        // device soffset is included too -- tools/micro/buffer_range.hip, profiles/r05_b_buffer_range_probe.txt)
#pragma unroll
        for (int it = 0; it < 8; ++it) st[it] = __builtin_amdgcn_raw_buffer_load_b128(rsX, voff + (ib + 16 * it) * (DIM * 4), 0, 0);
        if (tid < TILE_C) {
            const bool ok = ib + tid < nX;
            stn = ok ? a.nx[ib + tid] : 0.f;
            stt = ok ? a.tx[ib + tid] : 0.f;
        }
    };
    auto stage_write = [&](int b) {
        float *buf = lds + b * (TILE_C * KP);
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4 *>(&buf[(srow + 16 * it) * KP + scol]) = st[it];
        if (tid < TILE_C) {                                 // augmented k-pair (|x'|^2, 1) and the row's acceptance term
            *reinterpret_cast<float2 *>(&buf[tid * KP + DIM]) = make_float2(stn, 1.0f);
            tabuf[b * TILE_C + tid] = stt;
        }
    };

    float4 qf[16];
    float nq = 0.f, tyj = 0.f;
    int cb_held = -1;
    const int jw = wave * 32 + r;                           // this lane's column inside the 256-column block
    stage_load((int)(u0 % n_rt));
    stage_write(0);
    __syncthreads();
    int b = 0;
    for (long long u = u0; u < u1; ++u) {
        const int cb = (int)(u / n_rt), rt = (int)(u - (long long)cb * n_rt);
        if (u + 1 < u1) stage_load((int)((u + 1) % n_rt));
        const int j = cb * QB + jw;
        const bool j_ok = j < nY;
        if (cb != cb_held) {                                // resident fragments of this wave's 32 columns, scaled by -2 (exact)
            cb_held = cb;
            const float4 *yr = reinterpret_cast<const float4 *>(a.Yc + (size_t)(j_ok ? j : 0) * DIM) + h;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 v = j_ok ? yr[2 * t] : make_float4(0.f, 0.f, 0.f, 0.f);
                qf[t] = make_float4(-2.f * v.x, -2.f * v.y, -2.f * v.z, -2.f * v.w);
            }
            const float nyj = j_ok ? a.ny[j] : 0.f;
            tyj = j_ok ? a.ty[j] : 0.f;
            nq = (h == 0) ? 1.0f : nyj;
        }
        const int i0 = rt * TILE_C;
        const float *buf = lds + b * (TILE_C * KP);
        const float *rowp = buf + r * KP + 4 * h, *normp = buf + r * KP + DIM + h;
        const float *tab = tabuf + b * TILE_C;
        // the tile's rows of D as a range-checked view: rows >= nX are dropped by the hardware (32-bit byte range: 128 ldd 4)
        const int rows_here = min(TILE_C, nX - i0);
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(a.D + (size_t)i0 * a.ldd, 0,
                                                                             (int)((size_t)rows_here * a.ldd * 4), 0x00020000);
        unsigned ld4 = (unsigned)(a.ldd * 4);
        asm volatile("" : "+s"(ld4));                       // per unit: keeps the 128 row offsets (row x ld4) out of loop-invariant SGPRs
        const unsigned vo = j_ok ? (unsigned)j * 4u + (unsigned)(4 * h) * ld4 : 0x80000000u;
        const unsigned long long jmask = __ballot(j_ok);
        f32x16 a0, a1;
        const bool full = rows_here == TILE_C;              // uniform: only the last row tile of X is ragged
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // two chains per accumulator: k = 0..63 with the norm pair, k = 64..127 from zero, one add (the error bound above)
            f32x16 b0, b1;
            mfma_kgroups<0, 8, true>(a0, a1, rowp + 64 * half * KP, normp + 64 * half * KP, qf, nq);
            mfma_kgroups<8, 16, false>(b0, b1, rowp + 64 * half * KP, normp + 64 * half * KP, qf, nq);
            a0 += b0; a1 += b1;
            const int jb = cb * QB + wave * 32;
            if (full) {
                dist_emit<true>(a0, a, rsD, tab, 64 * half, h, jmask, tyj, vo, ld4, i0 + 64 * half, jb, rows_here);
                dist_emit<true>(a1, a, rsD, tab, 64 * half + 32, h, jmask, tyj, vo, ld4, i0 + 64 * half + 32, jb, rows_here);
            } else {
                dist_emit<false>(a0, a, rsD, tab, 64 * half, h, jmask, tyj, vo, ld4, i0 + 64 * half, jb, rows_here);
                dist_emit<false>(a1, a, rsD, tab, 64 * half + 32, h, jmask, tyj, vo, ld4, i0 + 64 * half + 32, jb, rows_here);
            }
        }
        if (u + 1 < u1) stage_write(b ^ 1);
        __syncthreads();
        b ^= 1;
    }
}

// The listed 32 x 32 blocks: one wave per block reads the block's stored values back, applies the acceptance test of the MFMA
// pass to them (same values, same tabulated terms: the same decision) and overwrites what fails it with the reference's own
// chain (the rows come from L2).
__global__ __launch_bounds__(256) void distance_fixup_kernel(const float *__restrict__ X, const float *__restrict__ Y, DistArgs a)
{
    const int n = *a.fix_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.fix_report) *a.fix_report = n;
    if (n > a.fix_cap) return;                              // overflow: the exact kernel behind this launch fills everything
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jl = lane & 31, ih = lane >> 5;
    for (int q = blockIdx.x * 4 + wave; q < n; q += gridDim.x * 4) {
        const int2 blk = a.fix_list[q];
        const int j = blk.y + jl;
        if (j >= a.nY) continue;
        const float tyj = a.ty[j];
        for (int rr = ih; rr < 32; rr += 2) {
            const int i = blk.x + rr;
            if (i >= a.nX) break;
            float *d = a.D + (size_t)i * a.ldd + j;
            const float sq = a.tx[i] + tyj;
            if (!(*d >= sq * sq))
                *d = exact_dist(reinterpret_cast<const float4 *>(X + (size_t)i * DIM), reinterpret_cast<const float4 *>(Y + (size_t)j * DIM));
        }
    }
}

// One wave per row: lane-local ascending scan + wave merge of (min1, lowest index, multiset second minimum).
__global__ __launch_bounds__(256) void set_matches_kernel(int *__restrict__ result, const float *__restrict__ distance,
                                                         int rows, int cols, int buffer_width, float ambiguity)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *d = distance + (size_t)row * buffer_width;
    float m1 = __builtin_inff(), m2 = __builtin_inff(); int i1 = 0x7fffffff;
    for (int j = lane; j < cols; j += 64) {
        const float c = d[j];
        if (c < m1) { m2 = m1; m1 = c; i1 = j; }
        else if (c < m2) m2 = c;
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const float o1 = __shfl_xor(m1, s), o2 = __shfl_xor(m2, s);
        const int oi = __shfl_xor(i1, s);
        const bool take = (o1 < m1) || (o1 == m1 && oi < i1);
        const float lo = take ? o1 : m1, hi = take ? m1 : o1;        // hi = the larger of the two minima
        const float s2 = take ? o2 : m2;                              // second of the winner's own list
        i1 = take ? oi : i1;
        m1 = lo;
        m2 = (hi < s2) ? hi : s2;                                     // the loser's own second is >= hi
    }
    if (lane == 0) {
        // (m1, i1, m2): what the scan's comparisons see -- NaN entries left out, +inf never recorded (strict < from +inf)
        int idx = (i1 == 0x7fffffff) ? 0 : i1;
        const float d0 = d[0];                                        // match.cu:90: a NaN at column 0 is never replaced
        if (d0 != d0) { m2 = m1; m1 = d0; idx = 0; }
        emit_match(row, m1, idx, m2, 0, ambiguity, result, nullptr, nullptr, nullptr);
    }
}

struct MatchWs { float *na, *nb; float4 *partial; float *partial3; int *fb_count, *fb_list; unsigned *As, *Bs; uint4 *nbslot;
                 unsigned *Ah, *Bh; float *ra, *rb, *na2; int *f1_list; };

// Which MFMA screen match_top2 runs: 0 = fp32 (v_mfma_f32_32x32x2_f32, K = 128 exact products), 1 = bf16x3 (split
// operands on v_mfma_f32_32x32x16_bf16), 2 = two-stage (a coarse pass with ONE fp16 product per k on
// v_mfma_f32_32x32x16_f16, then the bf16x3 screen on the rows -- about a percent on SIFT data -- whose coarse result
// cannot be proven). All feed the same exact finalize / fallback, so the results are identical; only the time differs.
// NM_MATCH_SCREEN=f32|bf16x3|f16 or nm_sift_match_set_screen() select it (process-wide).
static std::atomic<int> g_screen{-1};
static int match_screen()
{
    int v = g_screen.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("NM_MATCH_SCREEN");
        v = (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) ? 0 : (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) ? 1 : 2;
        g_screen.store(v, std::memory_order_relaxed);
    }
    return v;
}
// |screen value - exact squared distance| <= coeff (sqrt na + sqrt nb)^2  
// (two-stage screen: the coarse pass's fp32 accumulation only -- 134 terms, with a factor 2 of slack -- see match_finalize_kernel<1>)
// Round 4: the bf16x3 coefficient rose from 2.75e-5 to 3.15e-5. Its accumulation share was 65 u (u = 2^-24): twice the
// 25 u + 7 u of the model H-bf16 as probed by hand in round 2. nm_selftest_mfma_model (4.2e9 random results per instruction
// on MI355X, profiles/r04_b_mfma_model.txt) measures single instructions up to 1.91 x (bf16) / 1.73 x (f16) that model --
// the combination of the two halves and C loses up to one ulp, not half of one -- so 65 u was the bound itself, without
// slack. The share is now 140 u = 8.3e-6: 2.2 x the 64 u that the measured law |D - exact| <= 2 (u |D| + 7 u (pmax_lo +
// pmax_hi)) gives for a chain of 25 instructions (2 (25 u + 7 u)), and 3.2 x the largest chain error the self-test's
// adversarial families reach (2.6e-6). The coarse pass's 3.2e-5 = 537 u stands: its 9 instructions need 2 (9 u + 7 u) = 32 u.
static float screen_err_coeff(int screen) { return screen == 2 ? 3.2e-5f : screen ? 3.15e-5f : 1.56e-5f; }
// The share of those coefficients that covers the accumulation inside the matrix instructions (a hardware premise, measured
// by nm_selftest_mfma_model and asserted by tests/test_gpu_match.py): the whole of the coarse pass's coefficient (its
// representation error is bounded separately from measured residual norms), 140 x 2^-24 of the bf16x3 coefficient (the rest
// is the split's representation error, 2.31e-5, and the norms), and the fp32 screen's 2 gamma_129 fma-chain bound.
extern "C" __attribute__((visibility("default"))) float nm_sift_match_accum_budget(int screen)
{
    return screen == 2 ? screen_err_coeff(2) : screen == 1 ? 140.0f * 5.9604645e-8f : screen_err_coeff(0);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static MatchWs carve(void *workspace, int nA, int nB)
{
    MatchWs w;
    char *base = static_cast<char *>(workspace);
    w.na = reinterpret_cast<float *>(base); base += align256((size_t)nA * 4);
    w.nb = reinterpret_cast<float *>(base); base += align256(((size_t)nB + TILE_C) * 4);      // + the +inf padding of a tile
    w.fb_count = reinterpret_cast<int *>(base); base += 256;
    w.fb_list = reinterpret_cast<int *>(base); base += align256((size_t)nA * 4);
    // partial / partial3 are dead once match_finalize_kernel has run: the fallback reuses the space from `partial` on
    // for its FB_SPLIT slice results per listed query (<= nA * MAX_CHUNKS float4, covered by the workspace bound)
    w.partial = reinterpret_cast<float4 *>(base); base += align256((size_t)nA * MAX_CHUNKS * sizeof(float4));
    w.partial3 = reinterpret_cast<float *>(base); base += align256((size_t)nA * MAX_CHUNKS * sizeof(float));
    w.As = reinterpret_cast<unsigned *>(base); base += align256((size_t)nA * DIM * 4);
    w.Bs = reinterpret_cast<unsigned *>(base); base += align256((size_t)nB * DIM * 4);
    w.nbslot = reinterpret_cast<uint4 *>(base); base += align256(((size_t)nB + TILE_C) * sizeof(uint4));
    w.Ah = reinterpret_cast<unsigned *>(base); base += align256((size_t)nA * DIM * 2);
    w.Bh = reinterpret_cast<unsigned *>(base); base += align256((size_t)nB * DIM * 2);
    w.ra = reinterpret_cast<float *>(base); base += align256((size_t)nA * 4);
    w.rb = reinterpret_cast<float *>(base); base += align256((size_t)nB * 4);
    w.na2 = reinterpret_cast<float *>(base); base += align256((size_t)nA * 4);
    w.f1_list = reinterpret_cast<int *>(base);
    return w;
}

static size_t pair_workspace_bytes(int nA, int nB)
{
    if (nA < 0) nA = 0;
    if (nB < 0) nB = 0;
    return align256((size_t)nA * 4) + align256(((size_t)nB + TILE_C) * 4) + align256((size_t)nA * MAX_CHUNKS * sizeof(float4)) +
           align256((size_t)nA * MAX_CHUNKS * sizeof(float)) + 256 + align256((size_t)nA * 4) + 256 +
           align256((size_t)nA * DIM * 4) + align256((size_t)nB * DIM * 4) + align256(((size_t)nB + TILE_C) * sizeof(uint4)) +
           align256((size_t)nA * DIM * 2) + align256((size_t)nB * DIM * 2) + 3 * align256((size_t)nA * 4) + align256((size_t)nB * 4);
}

// How nm_sift_match_f32 fills a requested `distance` matrix: 1 = on the fp32 MFMA (distance_mfma_kernel: every entry within
// 1e-4 relative of the reference's chain, listed entries bit-equal to it), 0 = exact_distance_kernel (VALU; every entry
// bit-equal). NM_MATCH_DISTANCE=exact|mfma or nm_sift_match_set_distance_mode() select it (process-wide). The match indexes do
// not depend on it: they are decided on exactly recomputed distances either way.
static std::atomic<int> g_dist_mode{-1};
static int distance_mode()
{
    int v = g_dist_mode.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("NM_MATCH_DISTANCE");
        v = (e && (!strcmp(e, "exact") || !strcmp(e, "0"))) ? 0 : 1;
        g_dist_mode.store(v, std::memory_order_relaxed);
    }
    return v;
}

// The scratch of the MFMA distance pass lives INSIDE the matcher's workspace of the same (nA, nB): the pass runs before the
// fused matcher on the same stream, and every region it uses is (re)written by the matcher's own prep / screen kernels
// afterwards -- centred copies in the bf16 split images (same 512 bytes per row), norms and acceptance terms in the fp16
// images, column sums + counter + list in the partial lists. Returns false when the partial-list pool is too small for that
// (a few dozen query rows): the caller then takes the exact kernel.
static bool distance_scratch(const MatchWs &w, int nA, DistArgs &a, float **part)
{
    char *pool = reinterpret_cast<char *>(w.partial);
    const size_t pool_bytes = align256((size_t)nA * MAX_CHUNKS * sizeof(float4)) + align256((size_t)nA * MAX_CHUNKS * sizeof(float));
    const size_t head = (size_t)DIST_P * DIM * sizeof(float) + 256;
    if (pool_bytes < head + 65536) return false;
    *part = reinterpret_cast<float *>(pool);
    a.fix_count = reinterpret_cast<int *>(pool + (size_t)DIST_P * DIM * sizeof(float));
    a.fix_list = reinterpret_cast<int2 *>(pool + head);
    const size_t cap = (pool_bytes - head) / sizeof(int2);
    a.fix_cap = (int)(cap > 0x3fffffff ? 0x3fffffff : cap);
    a.Xc = reinterpret_cast<const float *>(w.As); a.Yc = reinterpret_cast<const float *>(w.Bs);
    a.nx = reinterpret_cast<const float *>(w.Ah); a.tx = a.nx + nA;                 // nA x 256 bytes available
    a.ny = reinterpret_cast<const float *>(w.Bh); a.ty = nullptr;                   // set by the caller (needs nB)
    a.fix_report = w.fb_count + 60;                        // a free word of the pair's 256-byte counter block
    return true;
}

// More than a quarter of the 32 x 32 blocks listed (one NaN or inf in B makes the mean row NaN and lists EVERY block; so do sets of
// near-duplicates): the per-block fix-up -- one wave per block, the rows from L2 -- is then far slower than exact_distance_kernel,
// so the list counts as overflowed from there on and the exact kernel behind it fills the whole matrix (ADVICE r5;
// tests/test_gpu_match.py::test_distance_with_one_nan_takes_the_exact_kernel).
static int distance_list_cap(int pool_cap, int nA, int nB)
{
    const long long blocks = (long long)nm_divup(nA, 32) * nm_divup(nB, 32);
    const long long quarter = blocks / 4 > 1 ? blocks / 4 : 1;
    return pool_cap > quarter ? (int)quarter : pool_cap;
}

static int run_distance(const float *A, int nA, const float *B, int nB, float *distance, const MatchWs &w, hipStream_t st)
{
    DistArgs a{};
    float *part = nullptr;
    const dim3 xgrid(nm_divup(nB, XD_TILE), nm_divup(nA, XD_TILE));
    if (distance_mode() == 0 || nA >= MATCH_MAX_ROWS || nB >= MATCH_MAX_ROWS || !distance_scratch(w, nA, a, &part)) {
        hipLaunchKernelGGL((exact_distance_kernel<false, false>), xgrid, dim3(256), 0, st, A, nA, B, nB, distance, (size_t)nB, nullptr, 0);
        NM_LAUNCH_CHECK();
        return 0;
    }
    a.ty = a.ny + nB;
    a.D = distance; a.ldd = (size_t)nB; a.nX = nA; a.nY = nB;
    a.fix_cap = distance_list_cap(a.fix_cap, nA, nB);
    hipLaunchKernelGGL(distance_colsum_kernel, dim3(DIST_P), dim3(128), 0, st, B, nB, part);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(distance_center_kernel, dim3(min(1024, nm_divup(max(nA, nB), 4)), 2), dim3(256), 0, st, A, nA, B, nB, part,
                       const_cast<float *>(a.Xc), const_cast<float *>(a.Yc), const_cast<float *>(a.nx), const_cast<float *>(a.ny),
                       const_cast<float *>(a.tx), const_cast<float *>(a.ty), a.fix_count);
    NM_LAUNCH_CHECK();
    const size_t lds_bytes = (size_t)2 * TILE_C * KP * sizeof(float) + 2 * TILE_C * sizeof(float);
    NM_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void *>(distance_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes));
    const long long units = (long long)nm_divup(nA, TILE_C) * nm_divup(nB, QB);
    const int grid = (int)std::min<long long>(units, nm_cu_count());
    nm_prof_begin(NM_PROF_DISTANCE, st);
    hipLaunchKernelGGL(distance_mfma_kernel, dim3(grid), dim3(512), lds_bytes, st, a);
    nm_prof_end(NM_PROF_DISTANCE, st);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(distance_fixup_kernel, dim3(2048), dim3(256), 0, st, A, B, a);
    NM_LAUNCH_CHECK();
    // a list that overflowed (more 32 x 32 blocks with an entry outside the bound than the partial-list pool holds): the exact kernel fills the whole matrix;
    // otherwise its workgroups return at once
    hipLaunchKernelGGL((exact_distance_kernel<false, false>), xgrid, dim3(256), 0, st, A, nA, B, nB, distance, (size_t)nB,
                       a.fix_count, a.fix_cap);
    NM_LAUNCH_CHECK();
    return 0;
}

struct MatchJob {                 // host-side description of one pair of a (possibly batched) call
    const float *A, *B;
    int nA, nB, mode, index_offset;
    int *result;
    float *min1, *min2;
    int *idx1;
    void *workspace;
    const int *d_nA, *d_nB;       // device-sized call: nA / nB above are the capacities, the real sizes live here
};

// norms (1 launch for all pairs) -> MFMA top-2 (1 launch per pair) -> finalize, fallback, merge (1 launch each)
static int run_fused_batch(const MatchJob *jobs, int n, float ambiguity, hipStream_t st, int phases = 7)
{
    if (n <= 0) return 0;
    if (n > MATCH_MAX_BATCH) return (int)hipErrorInvalidValue;
    MatchBatch bt{};
    MatchPlan plans[MATCH_MAX_BATCH];
    const int screen = match_screen();
    bt.ambiguity = ambiguity;
    bt.err_coeff = screen_err_coeff(screen);
    bt.err_coeff2 = screen_err_coeff(1);
    int max_rows = 0, max_a = 0;
    bool dev_sized = false;
    // persistent workgroups per CU of the (first) screening pass. (Two for the coarse pass -- its 68 KiB of LDS would allow
    // it -- measured 70 us instead of 58: half as many tiles per segment, twice the segment prologues.)
    const int wg_per_cu = 1;
    for (int k = 0; k < n; ++k) {
        const MatchJob &j = jobs[k];
        if (j.nA <= 0 || j.nB <= 0) continue;                 // empty sets: a no-op for this pair, as in the reference
        // 32-bit byte ranges of the SRDs, and the plan's 32-bit unit arithmetic (qblocks x T must stay far below 2^31):
        // checked BEFORE any plan is made
        if (j.nA >= MATCH_MAX_ROWS || j.nB >= MATCH_MAX_ROWS) return (int)hipErrorInvalidValue;
        const int q = bt.n++;
        const bool dev = j.d_nA != nullptr;
        if (q > 0 && dev != dev_sized) return (int)hipErrorInvalidValue;
        dev_sized = dev;
        if (!dev) plans[q] = make_plan(j.nA, j.nB, wg_per_cu);
        const MatchWs w = carve(j.workspace, j.nA, j.nB);
        MatchPair &c = bt.p[q];
        c.A = j.A; c.B = j.B; c.nA = j.nA; c.nB = j.nB; c.na = w.na; c.nb = w.nb; c.partial = w.partial;
        c.partial3 = w.partial3; c.fb_count = w.fb_count; c.fb_list = w.fb_list; c.nbmax = reinterpret_cast<float *>(w.fb_count + 16);
        c.S = dev ? MAX_CHUNKS : plans[q].S; c.mode = j.mode;
        c.index_offset = j.index_offset; c.result = j.result; c.min1 = j.min1; c.min2 = j.min2; c.idx1 = j.idx1;
        c.As = w.As; c.Bs = w.Bs; c.nbslot = w.nbslot;
        if (screen == 2) { c.Ah = w.Ah; c.Bh = w.Bh; c.ra = w.ra; c.rb = w.rb; c.na2 = w.na2; c.f1_list = w.f1_list; }
        c.d_nA = j.d_nA; c.d_nB = j.d_nB;
        // inside the 256-byte counter block. The two-stage screen's coarse kernel takes every pair's plan from there (one
        // launch for all pairs of the call), so nbmax_kernel makes it for the host-sized entries too
        c.d_plan = (dev || screen == 2) ? reinterpret_cast<MatchPlan *>(w.fb_count + 32) : nullptr;
        max_rows = max(max_rows, j.nA + nm_divup(j.nB, TILE_C) * TILE_C);
        max_a = max(max_a, j.nA);
    }
    if (bt.n == 0) return 0;
    const int n_cu = nm_cu_count();
    // second pass of the two-stage screen: about a percent of the rows, so a quarter of the CUs per pair is plenty and the
    // launch (one grid row per pair) does not spend its time dispatching workgroups that find nothing to do
    // (calls of a few pairs -- the 100k x 100k all-pairs case lists thousands of rows -- keep the whole chip). A pair of a
    // 16-pair call lists 10-25 rows of one 256-row block: 16 workgroups of ~6 tiles each instead of 47 of 2, and the 16
    // pairs' second passes run side by side in one round of workgroups.
    const int xcd = nm_xcd_count();
    int n_wg2 = n_cu;
    if (bt.n >= 4) n_wg2 = max(2 * xcd, (n_cu / bt.n) / xcd * xcd);
    if (n_wg2 > n_cu || n_wg2 % xcd != 0) n_wg2 = n_cu;
    bt.n_cu = n_cu * wg_per_cu; bt.n_cu2 = n_wg2; bt.n_xcd = nm_xcd_count();
    // Coarse pass, round 6: with a multiple of the XCD count of pairs in the call, pair q is screened by the 32 workgroups of
    // XCD q mod 8 alone (the XCDs work on 8 pairs side by side) instead of by all 256 on pair after pair: every XCD's L2 then
    // fetches ITS pairs' fp16 images once (2 x 3.1 MB, which it holds) where each of the 8 L2s fetched every pair's candidate
    // tiles (643 MB per 16-pair launch against 146 MB algorithmic). NM_COARSE_PAIR_XCD=0 keeps the old division.
    bt.pair_xcd = 0;
    if (xcd > 1 && xcd <= 8 && n_cu % xcd == 0 && bt.n >= xcd && bt.n % xcd == 0) {
        static const int env = [] { const char *e = getenv("NM_COARSE_PAIR_XCD"); return e ? atoi(e) : 1; }();
        // (the single-pass screens, NM_MATCH_SCREEN=f32 / bf16x3, take the same division: one launch per `xcd` pairs)
        static const int env1 = [] { const char *e = getenv("NM_TOP2_PAIR_XCD"); return e ? atoi(e) : 1; }();
        if (screen == 2 ? env : env1) {
            bt.pair_xcd = xcd; bt.n_cu = n_cu / xcd; bt.n_xcd = 1;
            if (screen != 2 && !dev_sized)                           // host-sized plans are made here: for the XCD's share of the chip
                for (int q = 0; q < bt.n; ++q) {
                    plans[q] = make_plan_on(bt.p[q].nA, bt.p[q].nB, bt.n_cu, 1, 0, 1, [](int v) { return v; });
                    bt.p[q].S = plans[q].S;
                }
        }
    }
    if (phases & NM_MATCH_PHASE_PREP) {
        const dim3 pg(nm_divup(max_rows, PREP_ROWS), bt.n);
        if (screen == 2) hipLaunchKernelGGL(prep_kernel<2>, pg, dim3(256), 0, st, bt);
        else if (screen) hipLaunchKernelGGL(prep_kernel<1>, pg, dim3(256), 0, st, bt);
        else hipLaunchKernelGGL(prep_kernel<0>, pg, dim3(256), 0, st, bt);
        NM_LAUNCH_CHECK();
        hipLaunchKernelGGL(nbmax_kernel, dim3(bt.n), dim3(1024), 0, st, bt);
        NM_LAUNCH_CHECK();
    }
    const size_t lds_full = (size_t)2 * TILE_C * KP * sizeof(float);
    // coarse pass: two 32 KiB images + the norm slots; the staged query block (256 rows at a 272-byte pitch) is as large
    const size_t lds_bytes = screen == 2 ? lds_full + 1024 + 2048 : lds_full;      // + the coarse pass's pair directory and query norms
    static_assert(2 * TILE_C * (DIM * 2) + 2 * TILE_C * 16 + QB * (DIM * 2) == 2 * TILE_C * KP * sizeof(float),
                  "coarse pass: two tile images, the norm slots and the query region take what the other screens' tiles take");
    // per call: the attribute is per device, and a process may drive several (cheap host-side call, not a stream op)
    NM_RETURN_IF(hipFuncSetAttribute(screen == 2 ? reinterpret_cast<const void *>(match_coarse_kernel)
                                     : screen ? reinterpret_cast<const void *>(match_top2_kernel<1>)
                                              : reinterpret_cast<const void *>(match_top2_kernel<0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (screen == 2)
        NM_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void *>(match_top2_rows_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full));
    else if (bt.pair_xcd > 0)
        NM_RETURN_IF(hipFuncSetAttribute(screen ? reinterpret_cast<const void *>(match_top2_group_kernel<1>)
                                                : reinterpret_cast<const void *>(match_top2_group_kernel<0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (int q = 0; (phases & NM_MATCH_PHASE_SCREEN) && q < bt.n; ++q) {
        const MatchPair &c = bt.p[q];
        // device-sized: one workgroup per CU, of which the first plan.G (decided on the device) work
        const int grid = dev_sized ? n_cu * wg_per_cu : plans[q].G;
        const MatchPlan plan_arg = dev_sized ? MatchPlan{} : plans[q];
        // (profiling hook: one event pair per PAIR of the call. The two-stage screen's coarse pass is one launch for all
        // pairs: the first pair's events bracket it, the others are recorded back to back.)
        nm_prof_begin(NM_PROF_MATCH_TOP2, st);
        if (screen == 2) {
            if (q == 0) hipLaunchKernelGGL(match_coarse_kernel, dim3(n_cu), dim3(512), lds_bytes, st, bt);
        } else if (bt.pair_xcd > 0) {
            // one launch per `xcd` pairs (the first pair's events bracket it, the others' are recorded back to back)
            if (q % bt.pair_xcd == 0) {
                MatchPlanPack pk{};
                for (int k = 0; k < bt.pair_xcd; ++k) pk.p[k] = dev_sized ? MatchPlan{} : plans[q + k];
                if (screen) hipLaunchKernelGGL(match_top2_group_kernel<1>, dim3(n_cu), dim3(512), lds_bytes, st, bt, q, bt.pair_xcd, pk);
                else hipLaunchKernelGGL(match_top2_group_kernel<0>, dim3(n_cu), dim3(512), lds_bytes, st, bt, q, bt.pair_xcd, pk);
            }
        } else if (screen)
            hipLaunchKernelGGL(match_top2_kernel<1>, dim3(grid), dim3(512), lds_bytes, st,
                               reinterpret_cast<const float *>(c.As), c.nA, reinterpret_cast<const float *>(c.Bs), c.nB,
                               c.na, c.nb, c.nbslot, plan_arg, c.partial, c.partial3, c.d_nA, c.d_nB, c.d_plan);
        else
            hipLaunchKernelGGL(match_top2_kernel<0>, dim3(grid), dim3(512), lds_bytes, st, c.A, c.nA, c.B, c.nB,
                               c.na, c.nb, c.nbslot, plan_arg, c.partial, c.partial3, c.d_nA, c.d_nB, c.d_plan);
        nm_prof_end(NM_PROF_MATCH_TOP2, st);
        NM_LAUNCH_CHECK();
    }
    if (!(phases & NM_MATCH_PHASE_FINISH)) return 0;
    const dim3 fg(nm_divup(4 * max_a, 256), bt.n);
    if (screen == 2) {
        // coarse result proven -> emitted; the others are listed, split, screened by the bf16x3 kernel (all pairs in one
        // launch, its plan made on the device for the number listed) and finalized as the single-pass screens are
        hipLaunchKernelGGL(match_finalize_kernel<1>, fg, dim3(256), 0, st, bt);
        NM_LAUNCH_CHECK();
        // (the lists are short: a few workgroups per pair stride over them, see the kernels)
        hipLaunchKernelGGL(fine_rows_kernel, dim3(min(bt.n >= 4 ? 64 : 1024, nm_divup(max_a, PREP_ROWS)), bt.n), dim3(256), 0, st, bt);
        NM_LAUNCH_CHECK();
        hipLaunchKernelGGL(match_top2_rows_kernel, dim3(n_wg2, bt.n), dim3(512), lds_full, st, bt);
        NM_LAUNCH_CHECK();
        hipLaunchKernelGGL(match_finalize_kernel<2>, dim3(min((int)fg.x, bt.n >= 4 ? 32 : 512), bt.n), dim3(256), 0, st, bt);
    } else {
        hipLaunchKernelGGL(match_finalize_kernel<0>, fg, dim3(256), 0, st, bt);
    }
    NM_LAUNCH_CHECK();
    static_assert(FB_SPLIT <= MAX_CHUNKS, "fallback slices reuse the partial area");
    // few rows are ever listed (0-2 of 12k on SIFT data): a small grid drains fastest when the list is empty, and its
    // workgroups loop over the entries when it is not
    // (~1 000 workgroups whatever the number of pairs: with 16 chunk loops for each of 16 pairs the launch spent ~150 us
    // dispatching 16 384 workgroups that found their lists empty)
    hipLaunchKernelGGL(match_fallback_kernel, dim3(FB_SPLIT, max(1, FB_CHUNKS / bt.n), bt.n), dim3(256), 0, st, bt);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(match_fallback_merge_kernel, dim3(8, bt.n), dim3(256), 0, st, bt);
    NM_LAUNCH_CHECK();
    return 0;
}

static int run_fused(const float *A, int nA, const float *B, int nB, int mode, int index_offset, float ambiguity,
                     int *result, float *min1, int *idx1, float *min2, void *workspace, hipStream_t st)
{
    const MatchJob j{A, B, nA, nB, mode, index_offset, result, min1, min2, idx1, workspace, nullptr, nullptr};
    return run_fused_batch(&j, 1, ambiguity, st);
}

}  // namespace

extern "C" {

int nm_transpose_f32(float *odata, const float *idata, int width, int height, void *stream)
{
    if (width <= 0 || height <= 0) return 0;
    dim3 grid(nm_divup(width, 32), nm_divup(height, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, nm_stream(stream), odata, idata, width, height);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_bf_distance_f32(const float *A, int size_A, const float *B, int size_B, int sift_vector_size, float *result,
                       void *stream)
{
    if (sift_vector_size != DIM) return (int)hipErrorInvalidValue;
    if (size_A <= 0 || size_B <= 0) return 0;
    // rows of the (transposed) output = candidates j, columns = queries i: result[j * size_A + i]; A arrives k-major
    dim3 grid(nm_divup(size_A, XD_TILE), nm_divup(size_B, XD_TILE));
    hipLaunchKernelGGL((exact_distance_kernel<false, true>), grid, dim3(256), 0, nm_stream(stream), B, size_B, A, size_A,
                       result, (size_t)size_A, nullptr, 0);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_get_sift_matches_f32(const float *distance, int rows, int cols, int buffer_width, int *result, float ambiguity,
                            void *stream)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(set_matches_kernel, dim3(nm_divup(rows, 4)), dim3(256), 0, nm_stream(stream), result, distance,
                       rows, cols, buffer_width, ambiguity);
    NM_LAUNCH_CHECK();
    return 0;
}

// Upper bound for every call with nA' <= nA and nB' <= nB on the same workspace (the grid plan, hence the number of
// candidate chunks S <= MAX_CHUNKS, depends on the actual sizes).
size_t nm_sift_match_workspace_bytes(int nA, int nB) { return pair_workspace_bytes(nA, nB); }

int nm_sift_match_set_screen(int screen)
{
    if (screen < 0 || screen > 2) return (int)hipErrorInvalidValue;
    g_screen.store(screen, std::memory_order_relaxed);
    return 0;
}
int nm_sift_match_get_screen(void) { return match_screen(); }

int nm_sift_match_pairs_per_launch(int n_pairs)
{
    if (n_pairs <= 0) return 0;
    const int screen = match_screen(), xcd = nm_xcd_count(), n_cu = nm_cu_count();
    if (screen == 2) return n_pairs;
    const char *e = getenv("NM_TOP2_PAIR_XCD");
    const bool grouped = (!e || atoi(e)) && xcd > 1 && xcd <= 8 && n_cu % xcd == 0 && n_pairs >= xcd && n_pairs % xcd == 0;
    return grouped ? xcd : 1;
}

float nm_sift_match_distance_budget(void) { return DIST_C; }

int nm_sift_match_set_distance_mode(int mode)
{
    if (mode < 0 || mode > 1) return (int)hipErrorInvalidValue;
    g_dist_mode.store(mode, std::memory_order_relaxed);
    return 0;
}
int nm_sift_match_get_distance_mode(void) { return distance_mode(); }
// Diagnostics: entries of the last MFMA distance pass on `workspace` (same nA, nB) that were listed for the exact recomputation,
// and the capacity of the list (more listed than that = the whole matrix was filled by the exact kernel). Synchronises.
int nm_sift_match_distance_listed(const void *workspace, int nA, int nB, int *listed, int *capacity, void *stream)
{
    if (!workspace || !listed || nA <= 0 || nB <= 0) return (int)hipErrorInvalidValue;
    DistArgs a{};
    float *part = nullptr;
    *listed = -1;
    if (capacity) *capacity = 0;
    if (!distance_scratch(carve(const_cast<void *>(workspace), nA, nB), nA, a, &part)) return 0;
    if (capacity) *capacity = distance_list_cap(a.fix_cap, nA, nB);
    NM_RETURN_IF(hipStreamSynchronize(nm_stream(stream)));
    return (int)hipMemcpy(listed, a.fix_report, sizeof(int), hipMemcpyDeviceToHost);
}

size_t nm_sift_match_batch_workspace_bytes(int n, const int *nA, const int *nB)
{
    size_t total = 0;
    for (int k = 0; k < n; ++k) total += pair_workspace_bytes(nA ? nA[k] : 0, nB ? nB[k] : 0);
    return total;
}

int nm_sift_match_batch_f32(int n, const float *const *A, const int *nA, const float *const *B, const int *nB,
                            int *const *result, float ambiguity, void *workspace, void *stream)
{
    if (n <= 0) return 0;
    if (n > MATCH_MAX_BATCH || !A || !nA || !B || !nB || !result || !workspace) return (int)hipErrorInvalidValue;
    MatchJob jobs[MATCH_MAX_BATCH];
    char *ws = static_cast<char *>(workspace);
    for (int k = 0; k < n; ++k) {
        jobs[k] = MatchJob{A[k], B[k], nA[k], nB[k], 0, 0, result[k], nullptr, nullptr, nullptr, ws, nullptr, nullptr};
        ws += pair_workspace_bytes(nA[k], nB[k]);
    }
    return run_fused_batch(jobs, n, ambiguity, nm_stream(stream));
}

// Device-sized batch: what a live client needs -- the frame driver leaves its descriptor count on the DEVICE
// (d_num_items, siftfunctions.cu:165-178 keeps it on the host after a synchronisation per level), and this entry reads it
// there. Every grid and the workspace are laid out for (capA, capB); the work plan is made on the device for the real
// sizes. No host synchronisation, no allocation: detect -> match chains on a stream and can be captured in a HIP graph
// that is replayed on different frames.
size_t nm_sift_match_batch_dev_workspace_bytes(int n, int capA, int capB)
{
    return (size_t)(n > 0 ? n : 0) * pair_workspace_bytes(capA, capB);
}

int nm_sift_match_batch_dev_f32(int n, const float *const *A, const int *const *d_nA, const float *const *B,
                                const int *const *d_nB, int capA, int capB, int *const *result, float ambiguity,
                                void *workspace, void *stream)
{
    return nm_sift_match_batch_dev_phases_f32(NM_MATCH_PHASE_PREP | NM_MATCH_PHASE_SCREEN | NM_MATCH_PHASE_FINISH, n, A, d_nA, B,
                                              d_nB, capA, capB, result, ambiguity, workspace, stream);
}

int nm_sift_match_batch_dev_phases_f32(int phases, int n, const float *const *A, const int *const *d_nA, const float *const *B,
                                       const int *const *d_nB, int capA, int capB, int *const *result, float ambiguity,
                                       void *workspace, void *stream)
{
    if (n <= 0) return 0;
    if (phases <= 0 || phases > 7) return (int)hipErrorInvalidValue;
    if (n > MATCH_MAX_BATCH || !A || !d_nA || !B || !d_nB || !result || !workspace || capA <= 0 || capB <= 0)
        return (int)hipErrorInvalidValue;
    MatchJob jobs[MATCH_MAX_BATCH];
    char *ws = static_cast<char *>(workspace);
    for (int k = 0; k < n; ++k) {
        if (!A[k] || !B[k] || !d_nA[k] || !d_nB[k] || !result[k]) return (int)hipErrorInvalidValue;
        jobs[k] = MatchJob{A[k], B[k], capA, capB, 0, 0, result[k], nullptr, nullptr, nullptr, ws, d_nA[k], d_nB[k]};
        ws += pair_workspace_bytes(capA, capB);
    }
    return run_fused_batch(jobs, n, ambiguity, nm_stream(stream), phases);
}

int nm_sift_match_plan(int nA, int nB, int plan[10])
{
    if (!plan || nA < 0 || nB < 0 || nA >= MATCH_MAX_ROWS || nB >= MATCH_MAX_ROWS) return (int)hipErrorInvalidValue;
    const MatchPlan p = make_plan(nA, nB);
    plan[0] = p.qblocks; plan[1] = p.T; plan[2] = p.G; plan[3] = p.S; plan[4] = p.X; plan[5] = p.Gx;
    plan[6] = p.Tc; plan[7] = p.C; plan[8] = p.q_base; plan[9] = p.q_rem;
    return 0;
}

// HOST function for tests: the segments workgroup `wg` processes, in order, as rows (query block, first tile, tiles, slot,
// ends_block); returns their number (at most max_segments are written).
int nm_sift_match_plan_segments(int nA, int nB, int wg, int *segments, int max_segments)
{
    if (nA < 0 || nB < 0 || wg < 0 || nA >= MATCH_MAX_ROWS || nB >= MATCH_MAX_ROWS) return -1;
    const MatchPlan p = make_plan(nA, nB);
    if (wg >= p.G) return 0;
    const int xg = wg % p.X, vg = wg / p.X;
    const PlanGroup grp = plan_group(p, xg);
    const unit_t u_end = group_begin(grp, vg + 1);
    SegIter it;
    it.init(group_begin(grp, vg), u_end);
    int n = 0;
    unit_t u;
    while (it.next(p, grp, u)) {
        int pc, qbl, tt, Lc;
        plan_locate(p, grp, u, pc, qbl, tt, Lc);
        const int ntiles = (int)((Lc - tt) < (u_end - u) ? (Lc - tt) : (u_end - u));
        if (segments && n < max_segments) {
            int *r = segments + 5 * n;
            r[0] = grp.q0 + qbl; r[1] = pc * p.Tc + tt; r[2] = ntiles; r[3] = plan_slot(p, grp, pc, qbl, vg);
            r[4] = (pc == p.C - 1 && tt + ntiles == Lc) ? 1 : 0;
        }
        ++n;
    }
    return n;
}

int nm_sift_match_fallback_count(const void *workspace, int nA, int nB, int *host_count, void *stream)
{
    if (!workspace || !host_count || nA <= 0 || nB <= 0) return (int)hipErrorInvalidValue;
    MatchWs w = carve(const_cast<void *>(workspace), nA, nB);
    NM_RETURN_IF(hipMemcpyAsync(host_count, w.fb_count, sizeof(int), hipMemcpyDeviceToHost, nm_stream(stream)));
    return (int)hipStreamSynchronize(nm_stream(stream));
}

int nm_sift_match_second_pass_count(const void *workspace, int nA, int nB, int *host_count, void *stream)
{
    if (!workspace || !host_count || nA <= 0 || nB <= 0) return (int)hipErrorInvalidValue;
    MatchWs w = carve(const_cast<void *>(workspace), nA, nB);
    NM_RETURN_IF(hipMemcpyAsync(host_count, w.fb_count + 4, sizeof(int), hipMemcpyDeviceToHost, nm_stream(stream)));
    return (int)hipStreamSynchronize(nm_stream(stream));
}

int nm_sift_match_f32(const float *A, int nA, const float *B, int nB, float *distance, int *result, float ambiguity,
                      void *workspace, void *stream)
{
    if (nA <= 0 || nB <= 0) return 0;
    hipStream_t st = nm_stream(stream);
    if (distance) {                     // distance[i * nB + j]: rows = queries, columns = candidates
        if (!workspace) return (int)hipErrorInvalidValue;
        const int rc = run_distance(A, nA, B, nB, distance, carve(workspace, nA, nB), st);
        if (rc) return rc;
    }
    return run_fused(A, nA, B, nB, 0, 0, ambiguity, result, nullptr, nullptr, nullptr, workspace, st);
}

int nm_sift_match_shard_f32(const float *A, int nA, const float *B_shard, int nB_shard, int index_offset, float *min1,
                            int *idx1, float *min2, void *workspace, void *stream)
{
    if (nA > 0 && nB_shard <= 0) {      // nothing to scan: (inf, -1, min2 initial) loses against every real candidate
        hipLaunchKernelGGL(shard_neutral_kernel, dim3(nm_divup(nA, 256)), dim3(256), 0, nm_stream(stream), min1, idx1,
                           min2, nA);
        NM_LAUNCH_CHECK();
        return 0;
    }
    return run_fused(A, nA, B_shard, nB_shard, 1, index_offset, 0.f, nullptr, min1, idx1, min2, workspace,
                     nm_stream(stream));
}

// ---- native multi-GPU entry: shard -> ONE ncclAllGather of 12 B per row per rank -> merge ----
// RCCL is resolved at run time from the process (the caller created the communicator, so its RCCL is loaded already;
// torch ships its own copy): libnm_hip.so has no link-time dependency on librccl.
typedef int (*NmAllGatherFn)(const void *, void *, size_t, int, void *, hipStream_t);
static NmAllGatherFn nm_resolve_allgather()
{
    static NmAllGatherFn fn = [] {
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym) {
            void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) sym = dlsym(h, "ncclAllGather");
        }
        return reinterpret_cast<NmAllGatherFn>(sym);
    }();
    return fn;
}

int nm_sift_match_merge_packed_f32(const int *packed, int n_shards, int nA, int *result, float ambiguity, void *stream)
{
    if (nA <= 0 || n_shards <= 0) return 0;
    if (!packed || !result) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(match_merge_packed_kernel, dim3(nm_divup(nA, 256)), dim3(256), 0, nm_stream(stream), packed, n_shards,
                       nA, ambiguity, result);
    NM_LAUNCH_CHECK();
    return 0;
}

size_t nm_sift_match_allgather_workspace_bytes(int nA, int nB_shard, int n_ranks)
{
    if (nA < 0) nA = 0;
    if (n_ranks < 1) n_ranks = 1;
    return pair_workspace_bytes(nA, nB_shard) + align256((size_t)3 * nA * 4) + align256((size_t)n_ranks * 3 * nA * 4);
}

int nm_sift_match_allgather_f32(const float *A, int nA, const float *B_shard, int nB_shard, int index_offset, int n_ranks,
                                int *result, float ambiguity, void *workspace, void *nccl_comm, void *stream)
{
    if (nA <= 0) return 0;
    if (!A || !result || !workspace || n_ranks < 1 || (n_ranks > 1 && !nccl_comm)) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    char *base = static_cast<char *>(workspace) + pair_workspace_bytes(nA, nB_shard);
    int *mine = reinterpret_cast<int *>(base);
    int *gathered = reinterpret_cast<int *>(base + align256((size_t)3 * nA * 4));
    int rc = nm_sift_match_shard_f32(A, nA, B_shard, nB_shard, index_offset, reinterpret_cast<float *>(mine), mine + nA,
                                     reinterpret_cast<float *>(mine + 2 * (size_t)nA), workspace, stream);
    if (rc) return rc;
    const int *merged_from = mine;
    if (n_ranks > 1) {
        const NmAllGatherFn allgather = nm_resolve_allgather();
        if (!allgather) return (int)hipErrorNotSupported;                 // no RCCL in this process
        const int nrc = allgather(mine, gathered, (size_t)3 * nA, /* ncclInt32 */ 2, nccl_comm, st);
        if (nrc != 0) return (int)hipErrorUnknown;
        merged_from = gathered;
    }
    return nm_sift_match_merge_packed_f32(merged_from, n_ranks, nA, result, ambiguity, stream);
}

int nm_sift_match_merge_f32(const float *min1, const int *idx1, const float *min2, int n_shards, int nA, int *result,
                            float ambiguity, void *stream)
{
    if (nA <= 0 || n_shards <= 0) return 0;
    hipLaunchKernelGGL(match_merge_kernel, dim3(nm_divup(nA, 256)), dim3(256), 0, nm_stream(stream), min1, idx1, min2,
                       n_shards, nA, ambiguity, result);
    NM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

#ifdef NM_COARSE_STAMPS
extern "C" __attribute__((visibility("default"))) int nm_debug_coarse_stamps(unsigned long long *host_dst)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nm_coarse_stamps), sizeof(nm_coarse_stamps));
}
#endif
