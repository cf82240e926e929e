// nm_keypoint.hip -- DoG extrema, one-shot sub-pixel refinement, ordered (raster) stream compaction for gfx950.
// Replaces kernels/keypoint.cu:19-251 and the thrust::copy_if of sift/pyramidata.cu:84-91. The reference's texture
// fetches at (x+0.5,y+0.5) are exact texel loads (utils/cudatex2D.cu:15-19), so planes are read as plain arrays.
#include <algorithm>
#include <atomic>
#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "nm_keypoint.hpp"
#include "nm_detect_dev.hpp"
#include "../../include/nm_abi.h"

using nmfp::fma32;
using namespace nmdet;

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

// (Round 5, measured and removed: count_valid3_kernel + scan_counts3_kernel as ONE launch -- every workgroup publishes its count
// write-through, the last one to finish, found by an agent-scope ticket, scans. 24 000 tickets on one address cost 300 us per
// launch; with 16 units per workgroup the ticket is free, but the last workgroup's acquire + scan is a 15 us tail on every
// launch, however small -- 56 us at 1080p octave 0 against 15 + 8 for the two launches. profiles/r05_l_dropin_kernel_summary.txt)
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

// Exclusive scans of NL arrays of n ints each (in + l * stride -> out + l * stride) by ONE workgroup of 1024 threads; total[l]
// receives array l's sum (every thread). A thread owns `per` consecutive elements of every array: all its loads are issued
// before the first is used (the kernel is one workgroup's chain of memory round trips), the running sums go through lane
// shuffles inside a wave and one LDS word per wave and array across waves: one barrier instead of twenty per array.
// s: 16 * NL + 1 ints of LDS.
template <int NL>
__device__ __forceinline__ void block_exclusive_scan_1024(const int *__restrict__ in, int *__restrict__ out, int n, int stride,
                                                          int *s, int (&total)[NL])
{
    constexpr int MAXPER = 12;                             // elements per thread kept in registers (n <= 12 288)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + 1023) / 1024;
    const int beg = min(t * per, n), end = min(beg + per, n);
    int v[NL][MAXPER], sum[NL];
    const bool regs = per <= MAXPER;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        sum[l] = 0;
        if (regs) {
#pragma unroll
            for (int k = 0; k < MAXPER; ++k) v[l][k] = (beg + k < end) ? in[l * stride + beg + k] : 0;
#pragma unroll
            for (int k = 0; k < MAXPER; ++k) sum[l] += v[l][k];
        } else {
            for (int i = beg; i < end; ++i) sum[l] += in[l * stride + i];
        }
    }
    int incl[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        incl[l] = sum[l];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl[l], d);
            if (lane >= d) incl[l] += up;
        }
        if (lane == 63) s[l * 16 + wave] = incl[l];
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = s[l * 16 + w];
            if (w < wave) before += c;
            all += c;
        }
        total[l] = all;
        int run = before + incl[l] - sum[l];
        if (regs) {
#pragma unroll
            for (int k = 0; k < MAXPER; ++k)
                if (beg + k < end) { out[l * stride + beg + k] = run; run += v[l][k]; }
        } else {
            for (int i = beg; i < end; ++i) { const int c = in[l * stride + i]; out[l * stride + i] = run; run += c; }
        }
    }
    __syncthreads();                                       // s may be reused
}

__global__ __launch_bounds__(1024) void scan_counts_kernel(const int *__restrict__ counts, int *__restrict__ offsets,
                                                          int n, int *__restrict__ total_out)
{
    __shared__ int s[16];
    int total[1];
    block_exclusive_scan_1024<1>(counts, offsets, n, 0, s, total);
    if (threadIdx.x == 0 && total_out) *total_out = total[0];
}

__global__ __launch_bounds__(1024) void scan_counts3_kernel(NmCompact3 c)
{
    __shared__ int s[48];
    int total[3];
    block_exclusive_scan_1024<3>(c.counts, c.offsets, c.nb, c.nb, s, total);
    if (threadIdx.x < 3) c.totals[threadIdx.x] = total[threadIdx.x];
}

// ---- frame-driver kernels: detect 3 levels of one octave straight into per-unit staging, then scan + book-keeping,
//      then gather into the output-ordered keypoint list ----
// A unit is one image row of a segment of NM_DET_SEG_W = 248 pixels (units in raster order: u = y * nseg + seg). One workgroup
// per unit GROUP (DET_ROWS or DET_ROWS_TALL rows of a segment), four waves of 62 tested columns with their edge lanes as halo
// (nm_detect_dev.hpp). Every lane loads its own column of the window's rows x planes; horizontal neighbours come from the
// adjacent lanes by DPP wave shifts. The 26-neighbour strict extremum test is branch-free: per plane, max3/min3 of each row, then
//   is_max(level l) = c > max(M9[l], M9[l+2], M8[l+1]),  M9 = max of a plane's 3x3, M8 = the 3x3 without its centre.
// Only accepted candidates (rare) run the divergent sub-pixel refinement from global memory.
template <bool DENSE, bool LEV = false, bool MASKED = DENSE, int ROWS = DET_ROWS>
__global__ __launch_bounds__(256) void detect_stage_kernel(NmDetectArgs a)
{
    __shared__ DetectSmemT<ROWS> sm;
    if (DENSE && a.fill_blocks > 0 && (int)blockIdx.x >= a.det_blocks) {
        // the launch's reset part: -1 into what lies behind this octave's region in the three dense maps
        const size_t first = (size_t)a.ow * a.oh;
        const float4 inv = make_float4(-1.f, -1.f, -1.f, -1.f);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float4 *m = reinterpret_cast<float4 *>(a.dense[l]);
            for (size_t i = first + (size_t)(blockIdx.x - a.det_blocks) * 256 + threadIdx.x; i < a.reset_end[l];
                 i += (size_t)a.fill_blocks * 256)
                m[i] = inv;
        }
        return;
    }
    const int frame = blockIdx.y;
    DetectView v;
#pragma unroll
    for (int p = 0; p < 6; ++p) v.pl[p] = a.plane0[frame] ? a.plane0[frame] + p * a.plane_stride : a.api_planes[p];
    v.staging = a.staging[frame]; v.stage_stride = a.stage_stride; v.counts = a.counts[frame];
    v.dense = a.dense;
    v.mask = MASKED ? (DENSE ? a.mask : a.masks[frame]) : nullptr; v.mask_w = a.mask_w; v.mask_h = a.mask_h;
    v.ow = a.ow; v.oh = a.oh; v.peak = a.peak; v.edge = a.edge; v.xper = a.xper; v.sigma0 = a.sigma0;
    v.num_dogs = a.num_dogs; v.n_blocks = a.n_blocks; v.nseg = a.nseg;
    int blk = blockIdx.x;
    if (!DENSE && a.xcd_band > 0) {                      // NmDetectArgs::xcd_band
        blk = (blk & 7) * a.xcd_band + (blk >> 3);
        if (blk >= a.group_rows * a.nseg) return;        // the whole workgroup: padding of the last band
    }
    detect_stage_body<DENSE, LEV, MASKED, false, 1, ROWS>(v, blk, &sm);
}

__global__ __launch_bounds__(1024) void scan_book_kernel(NmScanArgs a)
{
    __shared__ int s[48];
    const int frame = blockIdx.x;
    int totals[3];
    block_exclusive_scan_1024<3>(a.counts[frame], a.offsets[frame], a.n_blocks, a.n_blocks, s, totals);
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

// (Round 5, measured and removed: the gather inside scan_book_kernel -- the workgroup that scanned a frame's counts copies the
// frame's staged keypoints itself, one dependent launch per octave fewer. 64-frame calls: no difference beyond the box noise
// (2 909 / 3 062 without, 2 942 / 2 921 with, same box); one frame: +29 us, a single workgroup copies the octave's thousands
// of keypoints alone.)
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

#ifndef NM_DET_TALL_MIN_DEFAULT
// 20-row unit groups of a launch from which it takes tall groups. Round 5: 2 048 (eight workgroups per CU). Round 6: 1 200 -- octave 2
// of a 64-frame 1080p call (1 792 such groups = 1 280 workgroups of 27 rows) 73 -> 52 us (profiles/r06_z_detect_tall_min.txt); a
// single frame's octave 0 (432) keeps the 5-row groups that fill the chip.
#define NM_DET_TALL_MIN_DEFAULT 1200
#endif
constexpr int DET_TALL_MIN_DEFAULT = NM_DET_TALL_MIN_DEFAULT;
static std::atomic<int> g_tall_min{DET_TALL_MIN_DEFAULT};

// One detection launch of the frame driver with ROWS image rows per unit group. bands: the unit groups are dealt to the XCDs in
// bands (NmDetectArgs::xcd_band) when every XCD gets at least eight.
template <int ROWS>
static void launch_detect_rows(NmDetectArgs &d, bool bands, hipStream_t stream)
{
    d.group_rows = nm_divup(d.oh, ROWS);
    const int groups = d.group_rows * d.nseg;
    d.xcd_band = (bands && groups >= 64) ? nm_divup(groups, 8) : 0;
    const dim3 grid(d.xcd_band ? 8 * d.xcd_band : groups, d.n);
    if (d.from_levels && d.any_mask) hipLaunchKernelGGL((detect_stage_kernel<false, true, true, ROWS>), grid, dim3(256), 0, stream, d);
    else if (d.from_levels) hipLaunchKernelGGL((detect_stage_kernel<false, true, false, ROWS>), grid, dim3(256), 0, stream, d);
    else if (d.any_mask) hipLaunchKernelGGL((detect_stage_kernel<false, false, true, ROWS>), grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((detect_stage_kernel<false, false, false, ROWS>), grid, dim3(256), 0, stream, d);
}

int nm_launch_detect_octave(const NmDetectArgs &d_in, const NmScanArgs &s, const NmGatherArgs &g, hipStream_t stream)
{
    NmDetectArgs d = d_in;
    if (d.n_blocks <= 0 || d.n <= 0) return 0;
    const bool prof = s.octave == 0;
    if (prof) nm_prof_begin(NM_PROF_DETECT_O0, stream);
    static const bool bands = [] { const char *e = getenv("NM_DETECT_XCD_BANDS"); return e ? atoi(e) != 0 : true; }();
    // tall unit groups when there are thousands of them even so (see DET_ROWS_TALL). Of the two tall heights the launch takes the
    // one under which the busiest XCD walks fewer segment rows, halo rows included: ceil(groups / 8) * (ROWS + 2) -- at 1080p
    // 27 rows divide into 320 groups, 40 per XCD and none of them partial (1 160 rows against the 1 188 of 432 groups of 20:
    // octave 0 of 64 frames 12.5 -> 11.9 us per frame; 24, 26, 28 rows are all SLOWER than 20: profiles/r06_z_detect_rows.txt).
    const long tall_groups = (long)d.nseg * nm_divup(d.oh, DET_ROWS_TALL) * d.n;
    if (d.from_levels && tall_groups >= g_tall_min.load(std::memory_order_relaxed)) {
        auto cost = [&](int rows) { return nm_divup(nm_divup(d.oh, rows) * d.nseg, 8) * (rows + 2); };
        static const bool rows2 = [] { const char *e = getenv("NM_DETECT_ROWS2"); return e ? atoi(e) != 0 : true; }();
        if (rows2 && cost(DET_ROWS_TALL2) < cost(DET_ROWS_TALL)) launch_detect_rows<DET_ROWS_TALL2>(d, bands, stream);
        else launch_detect_rows<DET_ROWS_TALL>(d, bands, stream);
    } else {
        launch_detect_rows<DET_ROWS>(d, bands, stream);
    }
    if (prof) nm_prof_end(NM_PROF_DETECT_O0, stream);
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
int nm_sift_set_detect_tall_min(int min_groups)
{
    return g_tall_min.exchange(min_groups < 0 ? DET_TALL_MIN_DEFAULT : min_groups);
}

size_t nm_find_keypoints3_compact_workspace_bytes(int width, int height)
{
    const size_t nb = (size_t)(height > 0 ? height : 1) * nm_divup(width > 0 ? width : 1, NM_DET_SEG_W);
    return 3 * nb * 256 * 16 + 2 * 3 * nb * sizeof(int) + 1024;
}

int nm_find_keypoints3_compact_f32(const float *const dog[5], int width, int height, float peak_threshold,
                                   float edge_threshold, float xper, float sigma_0, int num_dogs, int capacity, float *out,
                                   int *d_counts, void *workspace, void *stream)
{
    if (width <= 0 || height <= 0 || capacity <= 0) return 0;
    if (!dog || !out || !d_counts || !workspace) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    const int nseg = nm_divup(width, NM_DET_SEG_W), n_blocks = height * nseg;
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
    for (int i = 0; i < 5; ++i) { if (!dog[i]) return (int)hipErrorInvalidValue; d.api_planes[i] = dog[i]; }
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
    return nm_find_keypoints3_reset_f32(dog, mask, mask_width, mask_height, width, height, peak_threshold, edge_threshold, xper,
                                        sigma_0, num_dogs, result, nullptr, stream);
}

// The same launch, which ALSO resets entries [width * height, reset_end[l]) of result[l] to -1 (reset_end NULL or <= width *
// height: nothing): compute_keypoints' per-octave reset of the dense maps (thrust::fill, sift/siftfunctions.cu:120-121) without
// launches of its own -- the detection part writes every entry of the region, so only what an earlier, larger octave left
// behind it needs the reset.
int nm_find_keypoints3_reset_f32(const float *const dog[5], const float *mask, int mask_width, int mask_height, int width,
                                 int height, float peak_threshold, float edge_threshold, float xper, float sigma_0, int num_dogs,
                                 float *const result[3], const size_t reset_end[3], void *stream)
{
    if (width <= 0 || height <= 0) return 0;
    if (!dog || !result) return (int)hipErrorInvalidValue;
    NmDetectArgs d{};
    d.n = 1; d.ow = width; d.oh = height; d.peak = peak_threshold; d.edge = edge_threshold; d.xper = xper;
    d.sigma0 = sigma_0; d.num_dogs = num_dogs;
    d.nseg = nm_divup(width, NM_DET_SEG_W); d.n_blocks = height * d.nseg;
    for (int i = 0; i < 5; ++i) { if (!dog[i]) return (int)hipErrorInvalidValue; d.api_planes[i] = dog[i]; }
    for (int l = 0; l < 3; ++l) { if (!result[l]) return (int)hipErrorInvalidValue; d.dense[l] = result[l]; }
    d.mask = mask; d.mask_w = mask_width; d.mask_h = mask_height;
    d.det_blocks = d.nseg * nm_divup(height, DET_ROWS);
    size_t most = 0;
    const size_t region = (size_t)width * height;
    for (int l = 0; l < 3; ++l) {
        d.reset_end[l] = reset_end ? reset_end[l] : 0;
        if (d.reset_end[l] > region) most = std::max(most, d.reset_end[l] - region);
    }
    d.fill_blocks = (int)std::min<size_t>((most + 2047) / 2048, 2048);          // >= 8 entries per thread, at most 2048 workgroups
    hipLaunchKernelGGL(detect_stage_kernel<true>, dim3(d.det_blocks + d.fill_blocks, 1), dim3(256), 0, nm_stream(stream), d);
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
