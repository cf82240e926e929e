// nm_tail.hip -- the octave tail of the frame driver as ONE persistent launch (gfx950).
//
// A 1080p frame has six octaves; octaves 2..5 (480 x 270 and below) hold 6 % of its pixels but used to be 32 of its ~55
// dependent launches (5 Gaussian launches + detect + scan + gather per octave): pure launch latency for almost no work
// (profiles/r03_e_frame_timeline_batch1.txt). Inside a launch a hand-off through memory costs as much as a launch
// boundary (MI355X: a drained write-through publish + poll + acquire is 3-5 us), so the tail is organised around doing
// WITHOUT hand-offs where the data fits a CU's 160 KB of LDS:
//   * CONV_A (levels 1..3 of a 64 x 32 tile, or all five levels of a whole plane that fits): the input region with the
//     cumulative halo of the fused levels is loaded ONCE, every level is computed in LDS (row pass into a second LDS plane,
//     column pass back in place), each level's own pixels are written to its global plane on the way, level 3 also decimated
//     into the next octave's level 0 (kernels/downsample.cu:6-17). The halo is recomputed per tile instead of exchanged.
//   * CONV_B (levels 4..5 of a tile) starts from level 3, which its neighbours must have finished.
//   * gradient planes of levels 1..3 (kernels/cudamath.cu:38-54) come from the LDS copy of the level while it is an input
//     (tiles), or from GRAD items off the critical path (whole planes).
//   * DETECT = the frame driver's extrema + refinement body (nm_detect_dev.hpp) on one unit group, reading the six levels.
//   * tail_scan_kernel, a second small launch on the detection stream (one workgroup per frame): per octave the scan of the
//     per-unit counts, the book-keeping of scan_book_kernel (empty-level rule of sift/siftfunctions.cu:145,160, capacity rule
//     of :165-169) and the ordered gather (sift/pyramidata.cu:84-91). It continues the book of octave T - 1, whose detection
//     runs on that stream BESIDE the tail launch: were the scans items of the tail launch, it would have to wait for it.
// Arithmetic is the fp spec's: sum = fma(x[k], w[r - k], sum) from +0 for k = -r..r, rows then columns, zero padding,
// fp32 intermediate (kernels/convolution.cu:66-72,126-133) -- the same sequence as conv_pk_kernel / conv_sep_kernel /
// the oracle, so every plane is bit-identical to the per-octave launches'.
//
// Scheduling: the items of all frames form ONE list in a topological order (an item only depends on items with SMALLER
// numbers); persistent workgroups draw tickets from an agent-scope counter and wait, where needed, for per-(frame, octave)
// completion counters. The holder of the smallest unfinished ticket never waits, so the launch completes with ANY number of
// resident workgroups -- nothing depends on dispatch order, co-residency or XCD placement, and several such launches may run
// side by side (bench.py drives four detect streams). Hand-offs follow nm_devmem.hpp (write-through payload, drained, one
// counter add; relaxed poll, one agent acquire, barrier). The launch leaves its state words zero (the last workgroup out
// cleans up), so it needs no memset node and replays from a HIP graph.
#include "nm_tail.hpp"

#include <algorithm>
#include <cstdlib>

#include "../../include/nm_abi.h"
#include "nm_detect_dev.hpp"
#include "nm_devmem.hpp"
#include "nm_grad_dev.hpp"

namespace {

using namespace nmdev;
using nmfp::fma32;

constexpr int NT = 1024;                 // threads per workgroup: 16 waves, the only workgroup of its CU (the LDS frames)
constexpr int NQ = NT / 256;             // detection unit groups per item (the detection body is written for 256 threads)
constexpr int TW = 64, TH = 32;          // output tile of a tiled octave
constexpr int PAD = 13;                  // zero border kept around the image inside an LDS frame (>= the largest radius)
constexpr int SLACK = 8;                 // the 8-output tasks of a pass may overrun their region by 7
constexpr int GRAD_BAND = 16;            // rows per gradient item of a whole plane

// LDS frame of a conv item: image coordinates [fx0, fx0 + fw) x [fy0, fy0 + fh), pitch fp (odd: lanes that walk down a
// column of tasks hit different banks). Entries outside the image are zero and stay zero.
struct Frame { int fx0, fy0, fw, fh, fp; };

__device__ __forceinline__ int *counter_of(const NmTailArgs &a, int f, int slot, int k)
{
    return a.state + NM_TAIL_STATE_HEAD + ((f * NM_TAIL_MAX_OCT + slot) << 2) + k;
}

// (Packed v_pk_fma_f32 forms of both passes -- two rows / two columns half a region apart per register pair, taps in scalar
// registers, 8 or 4 outputs per task -- were built and measured on MI355X: bit-identical, but no faster (level step 4.3 -> 4.3-7
// us). In this one big kernel the windows of register pairs push the allocation over the 128 VGPRs a 1024-thread workgroup may
// have, and the spills cost more than the halved FMA count returns; the detection items, which share the allocation, went
// from 20 to 46 us.)
// rows [ry0, ry1) x columns [cx0, cx1) of the row pass: M[row][x] = sum_k A[row][x + k] w[R - k]
template <int R>
__device__ __forceinline__ void row_pass(const float *A, float *M, const Frame &F, const float *__restrict__ taps, int ry0,
                                         int ry1, int cx0, int cx1)
{
    float w[2 * R + 1];
#pragma unroll
    for (int i = 0; i <= 2 * R; ++i) w[i] = taps[i];
    const int nrows = ry1 - ry0, ntx = (cx1 - cx0 + 7) >> 3;
    for (int task = threadIdx.x; task < nrows * ntx; task += NT) {
        const int g = task / nrows, row = ry0 + (task - g * nrows);
        const int c = cx0 + 8 * g;
        const float *src = A + (row - F.fy0) * F.fp + (c - R - F.fx0);
        float v[8 + 2 * R];
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; ++j) v[j] = src[j];
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
        for (int k = -R; k <= R; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fma32(v[i + R + k], w[R - k], o[i]);
        }
        float *dst = M + (row - F.fy0) * F.fp + (c - F.fx0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c + i < cx1) dst[i] = o[i];
    }
}

// region [cx0, cx1) x [cy0, cy1) of the column pass: level value = sum_k M[y + k][x] w[R - k], stored back into A (the
// level becomes the next input; write_out sends the item's own pixels to the global plane afterwards)
template <int R>
__device__ __forceinline__ void col_pass(float *A, const float *M, const Frame &F, const float *__restrict__ taps, int cx0,
                                         int cx1, int cy0, int cy1)
{
    float w[2 * R + 1];
#pragma unroll
    for (int i = 0; i <= 2 * R; ++i) w[i] = taps[i];
    const int ncols = cx1 - cx0, nty = (cy1 - cy0 + 7) >> 3;
    for (int task = threadIdx.x; task < ncols * nty; task += NT) {
        const int g = task / ncols, x = cx0 + (task - g * ncols);
        const int y = cy0 + 8 * g;
        const float *src = M + (y - R - F.fy0) * F.fp + (x - F.fx0);
        float v[8 + 2 * R];
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; ++j) v[j] = src[j * F.fp];
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
        for (int k = -R; k <= R; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fma32(v[i + R + k], w[R - k], o[i]);
        }
        float *dst = A + (y - F.fy0) * F.fp + (x - F.fx0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (y + i < cy1) dst[i * F.fp] = o[i];
    }
}

// The item's own pixels [x0, x1) x [y0, y1) of the level held in A -> its global plane, write-through (other workgroups of
// this launch read it), 16 bytes per lane where the plane's rows allow; `down`: the same pixels decimated by 2 into the next
// octave's level 0 (kernels/downsample.cu:6-17: every other pixel of every other row).
__device__ __forceinline__ void write_out(const float *A, const Frame &F, int x0, int x1, int y0, int y1, float *plane, int ow,
                                          int oh, float *down)
{
    const int tw = x1 - x0, th = y1 - y0;
    if (((ow | tw | x0) & 3) == 0) {
        const int nq = tw >> 2;
        for (int i = threadIdx.x; i < nq * th; i += NT) {
            const int r = i / nq, x = x0 + 4 * (i - r * nq), y = y0 + r;
            const float *src = A + (y - F.fy0) * F.fp + (x - F.fx0);
            store_f4_agent(reinterpret_cast<float4 *>(plane + (size_t)y * ow + x), make_float4(src[0], src[1], src[2], src[3]));
        }
    } else {
        for (int i = threadIdx.x; i < tw * th; i += NT) {
            const int r = i / tw, x = x0 + (i - r * tw), y = y0 + r;
            store_f32_agent(plane + (size_t)y * ow + x, A[(y - F.fy0) * F.fp + (x - F.fx0)]);
        }
    }
    if (down) {
        const int dw = ow >> 1, dh = oh >> 1;
        const int dx0 = (x0 + 1) >> 1, dx1 = min((x1 + 1) >> 1, dw), dy0 = (y0 + 1) >> 1, dy1 = min((y1 + 1) >> 1, dh);
        const int nx = dx1 - dx0, ny = dy1 - dy0;
        if (nx > 0 && ny > 0) {
            if (((dw | nx | dx0) & 3) == 0) {
                const int nq = nx >> 2;
                for (int i = threadIdx.x; i < nq * ny; i += NT) {
                    const int r = i / nq, dx = dx0 + 4 * (i - r * nq), dy = dy0 + r;
                    const float *src = A + (2 * dy - F.fy0) * F.fp + (2 * dx - F.fx0);
                    store_f4_agent(reinterpret_cast<float4 *>(down + (size_t)dy * dw + dx), make_float4(src[0], src[2], src[4], src[6]));
                }
            } else {
                for (int i = threadIdx.x; i < nx * ny; i += NT) {
                    const int r = i / nx, dx = dx0 + (i - r * nx), dy = dy0 + r;
                    store_f32_agent(down + (size_t)dy * dw + dx, A[(2 * dy - F.fy0) * F.fp + (2 * dx - F.fx0)]);
                }
            }
        }
    }
}

template <int R>
__device__ __forceinline__ void level_step(float *A, float *M, const Frame &F, const float *__restrict__ taps, int rx0,
                                           int rx1, int ry0, int ry1, int py0, int py1, int x0, int x1, int y0, int y1,
                                           float *plane, int ow, int oh, float *down)
{
    row_pass<R>(A, M, F, taps, py0, py1, rx0, rx1);
    __syncthreads();
    col_pass<R>(A, M, F, taps, rx0, rx1, ry0, ry1);
    __syncthreads();
    write_out(A, F, x0, x1, y0, y1, plane, ow, oh, down);          // reads A; the next pass that WRITES A is behind a barrier
}

// gradient of the level held in A for the item's own pixels (border pixels = (0, 0): Q4 of SURVEY.md 8(a))
__device__ __forceinline__ void grad_from_lds(const float *A, const Frame &F, int x0, int x1, int y0, int y1, int ow, int oh,
                                              float2 *__restrict__ gplane)
{
    const int tw = x1 - x0, n = tw * (y1 - y0);
    for (int i = threadIdx.x; i < n; i += NT) {
        const int ry = i / tw, x = x0 + (i - ry * tw), y = y0 + ry;
        float2 g = make_float2(0.f, 0.f);
        if (x >= 1 && x < ow - 1 && y >= 1 && y < oh - 1) {
            const float *c = A + (y - F.fy0) * F.fp + (x - F.fx0);
            g = nmgrad::grad_of(c[-1], c[1], c[-F.fp], c[F.fp]);
        }
        gplane[(size_t)y * ow + x] = g;
    }
}

// One conv item: levels la..lb of the rectangle [x0, x1) x [y0, y1) of octave oc of frame fr.
// ph: NULL, or 12 words of the diagnostic trace that receive the clock after each phase of the item
#define NM_TAIL_STAMP(k) do { if (ph && threadIdx.x == 0) ph[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ void conv_item(const NmTailArgs &a, const NmTailOct &oc, const NmTailFrame &fr, int f, int slot, int x0, int x1,
                          int y0, int y1, int la, int lb, float *lds, unsigned long long *ph)
{
    const int ow = oc.ow, oh = oc.oh;
    int H = 0;                                             // cumulative halo of the fused levels
    for (int l = la; l <= lb; ++l) H += a.radii[l - 1];
    Frame F;
    F.fx0 = max(x0 - H, -PAD); F.fy0 = max(y0 - H, -PAD);
    F.fw = min(x1 + H, ow + PAD) - F.fx0 + SLACK;
    F.fh = min(y1 + H, oh + PAD) - F.fy0 + SLACK;
    F.fp = F.fw | 1;
    const int plane_f = (F.fp * F.fh + 3) & ~3;
    float *A = lds, *M = lds + plane_f;
    {   // zero both planes (padding, slack and everything a pass may read before it was written)
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < plane_f / 2; i += NT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    NM_TAIL_STAMP(0);
    if (ph && threadIdx.x == 0 && lb - la < 4) ph[10] = __builtin_amdgcn_s_memtime();       // shader clock, for the clock rate
    {   // input level la - 1 on the image part of the halo region
        const int rx0 = max(x0 - H, 0), rx1 = min(x1 + H, ow), ry0 = max(y0 - H, 0), ry1 = min(y1 + H, oh);
        const int rw = rx1 - rx0, n = rw * (ry1 - ry0);
        const float *src = fr.lev[slot][la - 1];
        for (int base = 0; base < n; base += 4 * NT) {     // four loads in flight per thread before the first LDS store
            float v[4];
            int dst[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * NT + (int)threadIdx.x;
                dst[k] = -1;
                v[k] = 0.f;
                if (i < n) {
                    const int r = i / rw, cx = rx0 + (i - r * rw), cy = ry0 + r;
                    v[k] = src[(size_t)cy * ow + cx];
                    dst[k] = (cy - F.fy0) * F.fp + (cx - F.fx0);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (dst[k] >= 0) A[dst[k]] = v[k];
        }
    }
    __syncthreads();
    NM_TAIL_STAMP(1);
    int h = H;                                             // halo of the level held in A
    for (int l = la; l <= lb; ++l) {
        const int R = a.radii[l - 1];
        const float *taps = a.taps[l - 1];
        if (!oc.whole && l - 1 >= 1 && l - 1 <= 3)          // gradient of the INPUT level (plane l - 2 of the octave's three)
            grad_from_lds(A, F, x0, x1, y0, y1, ow, oh, reinterpret_cast<float2 *>(fr.grad[slot]) + (size_t)(l - 2) * ow * oh);
        const int hn = h - R;                               // halo of level l
        const int rx0 = max(x0 - hn, 0), rx1 = min(x1 + hn, ow), ry0 = max(y0 - hn, 0), ry1 = min(y1 + hn, oh);
        const int py0 = max(y0 - h, 0), py1 = min(y1 + h, oh);              // rows of the row pass
        float *plane = fr.lev[slot][l];
        float *down = (l == 3 && oc.decimate) ? fr.lev[slot + 1][0] : nullptr;
        switch (R) {
            case 5: level_step<5>(A, M, F, taps, rx0, rx1, ry0, ry1, py0, py1, x0, x1, y0, y1, plane, ow, oh, down); break;
            case 7: level_step<7>(A, M, F, taps, rx0, rx1, ry0, ry1, py0, py1, x0, x1, y0, y1, plane, ow, oh, down); break;
            case 8: level_step<8>(A, M, F, taps, rx0, rx1, ry0, ry1, py0, py1, x0, x1, y0, y1, plane, ow, oh, down); break;
            case 10: level_step<10>(A, M, F, taps, rx0, rx1, ry0, ry1, py0, py1, x0, x1, y0, y1, plane, ow, oh, down); break;
            default: level_step<13>(A, M, F, taps, rx0, rx1, ry0, ry1, py0, py1, x0, x1, y0, y1, plane, ow, oh, down); break;
        }
        h = hn;
        NM_TAIL_STAMP(2 + 2 * (l - la));
        if (l == 3) publish_add(counter_of(a, f, slot, 0), 1);          // levels 1..3 (+ the seed of octave o + 1) are out
        NM_TAIL_STAMP(3 + 2 * (l - la));
    }
    if (ph && threadIdx.x == 0 && lb - la < 4) ph[11] = __builtin_amdgcn_s_memtime();
    if (lb == 5) publish_add(counter_of(a, f, slot, 1), 1);
}

// gradient item of a whole plane: level m (1..3), rows [band * GRAD_BAND, ...), straight from the global level plane
__device__ void grad_item(const NmTailOct &oc, const NmTailFrame &fr, int slot, int idx)
{
    const int bands = (oc.oh + GRAD_BAND - 1) / GRAD_BAND;
    const int m = 1 + idx / bands, band = idx - (m - 1) * bands;
    const int ow = oc.ow, oh = oc.oh, y0 = band * GRAD_BAND, y1 = min(y0 + GRAD_BAND, oh);
    const float *src = fr.lev[slot][m];
    float2 *gplane = reinterpret_cast<float2 *>(fr.grad[slot]) + (size_t)(m - 1) * ow * oh;
    const int n = ow * (y1 - y0);
    for (int i = threadIdx.x; i < n; i += NT) {
        const int ry = i / ow, x = i - ry * ow, y = y0 + ry;
        float2 g = make_float2(0.f, 0.f);
        if (x >= 1 && x < ow - 1 && y >= 1 && y < oh - 1) {
            const size_t c = (size_t)y * ow + x;
            g = nmgrad::grad_of(src[c - 1], src[c + 1], src[c - ow], src[c + ow]);
        }
        gplane[(size_t)y * ow + x] = g;
    }
}

// The book-keeping of the tail octaves of ONE frame by one workgroup (scan_book_kernel + gather_stage_kernel of
// nm_keypoint.hip for all of them at once): every octave's per-unit counts are loaded into LDS in one batch, each
// (octave, level) list is scanned by a wave of its own (chunks of 64 by lane shuffles: no workgroup barrier inside), one
// thread chains the book through the octaves (empty-level rule of sift/siftfunctions.cu:145,160, capacity rule of
// :165-169), and all output slots of all (octave, level) lists are gathered together (sift/pyramidata.cu:84-91: raster
// order). li: ints of LDS: the offsets, then 16 words per (octave, level).
template <int NTH>
__device__ void scan_gather_all(const NmTailArgs &a, const NmTailFrame &fr, int f, int *li)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_lists = 3 * a.n_oct;
    int base_of[NM_TAIL_MAX_OCT + 1];                     // first LDS word of octave slot j's 3 x n_blocks offsets
    base_of[0] = 0;
#pragma unroll
    for (int j = 0; j < NM_TAIL_MAX_OCT; ++j) base_of[j + 1] = base_of[j] + (j < a.n_oct ? 3 * a.oct[j].n_blocks : 0);
    int *s_off = li, *s_tot = li + base_of[NM_TAIL_MAX_OCT];     // s_tot[3 j + l]: total; [32 + ..]: lvl_n; [64 + ..]: lvl_base
    // the book's running count after octave T - 1 (an earlier launch of this stream): requested now, used after the scans
    const int items_before = (tid == 0 && a.oct[0].o != 0) ? fr.book->num_items : 0;
    // status of the tail launch this scan follows on its stream (state[3], nm_tail.hpp): non-zero = a wait timed out and the
    // items behind it were drained without working, so the staged lists and counts of the tail octaves are not this frame's
    const int tail_failed = (tid == 0) ? load_i32_agent(a.state + 3) : 0;
    {   // every octave's counts with ALL of a thread's loads in flight before the first LDS store: this launch is a chain of
        // memory round trips on the critical path of a single-frame call (a loop of load -> store pairs made it eleven of them)
        constexpr int MAXK = 12;
        const int n_all = base_of[NM_TAIL_MAX_OCT];
        if (n_all <= MAXK * NTH) {
            int v[MAXK];
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
                const int idx = tid + k * NTH;
                v[k] = 0;
                if (idx < n_all) {
                    const int *cp = fr.counts[0];           // select chain over compile-time slots: the pointers stay scalar loads
                    int b0 = 0;
#pragma unroll
                    for (int jj = 1; jj < NM_TAIL_MAX_OCT; ++jj)
                        if (jj < a.n_oct && idx >= base_of[jj]) { cp = fr.counts[jj]; b0 = base_of[jj]; }
                    v[k] = cp[idx - b0];
                }
            }
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (tid + k * NTH < n_all) s_off[tid + k * NTH] = v[k];
        } else {
            for (int j = 0; j < a.n_oct; ++j) {
                const int *counts = fr.counts[j];
                const int n = 3 * a.oct[j].n_blocks;
                for (int i = tid; i < n; i += NTH) s_off[base_of[j] + i] = counts[i];
            }
        }
    }
    __syncthreads();
    for (int q = wave; q < n_lists; q += NTH / 64) {       // exclusive scan of list q = (slot j, level l) by this wave
        const int j = q / 3, l = q - 3 * j, nb = a.oct[j].n_blocks;
        int *v = s_off + base_of[j] + l * nb;
        int run = 0;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int i = c0 + lane;
            const int c = i < nb ? v[i] : 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            if (i < nb) v[i] = run + incl - c;
            run += __shfl(incl, 63);
        }
        if (lane == 0) s_tot[q] = run;
    }
    __syncthreads();
    if (tid == 0) {
        NmFrameBook *b = fr.book;
        int num_items = items_before;
        for (int j = 0; j < a.n_oct; ++j) {
            const int o = a.oct[j].o;
            b->oct_base[o] = num_items;
            bool live = true;                             // sift/siftfunctions.cu:145,160: an empty level ends the octave
            for (int l = 0; l < 3; ++l) {
                const int cnt = (live && !tail_failed) ? s_tot[3 * j + l] : 0;
                if (cnt == 0) live = false;
                int n = cnt;
                if (n + num_items > a.capacity) n = a.capacity - num_items;       // siftfunctions.cu:165-169
                if (n < 0) n = 0;
                b->lvl_count[o][l] = cnt; b->lvl_base[o][l] = num_items; b->lvl_n[o][l] = n;
                s_tot[32 + 3 * j + l] = n; s_tot[64 + 3 * j + l] = num_items;
                num_items += n;
            }
            b->oct_base[o + 1] = num_items;
        }
        b->num_items = num_items;
        // a failed tail launch: the book stays consistent (the tail octaves are empty), the caller's count says -1 = invalid
        if (a.d_num_items[f]) *a.d_num_items[f] = tail_failed ? -1 : num_items;
        s_tot[96] = num_items - s_tot[64];                // keypoints of the tail octaves (their output slots are contiguous)
    }
    __syncthreads();
    float4 *out = reinterpret_cast<float4 *>(a.kpts[f]);
    const int first = s_tot[64], total = s_tot[96];
    for (int p = tid; p < total; p += NTH) {
        const int slot_out = first + p;
        int q = 0;                                        // the list that holds output slot `slot_out`
        while (q + 1 < n_lists && slot_out >= s_tot[64 + q + 1]) ++q;
        while (s_tot[32 + q] == 0 && q + 1 < n_lists) ++q;         // (lists of zero length share their successor's base)
        const int j = q / 3, l = q - 3 * j, nb = a.oct[j].n_blocks, pos = slot_out - s_tot[64 + q];
        const int *off = s_off + base_of[j] + l * nb;
        int lo = 0, hi = nb;                              // last unit whose exclusive offset is <= pos
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= pos) lo = mid; else hi = mid;
        }
        const float4 *st = reinterpret_cast<const float4 *>(fr.staging[j]) + (size_t)l * fr.stage_stride[j];
        out[slot_out] = st[(size_t)lo * 256 + (pos - off[lo])];
    }
}

// NQ unit groups per item, one per 256-thread quarter of the workgroup (idx-th item: groups NQ idx .. NQ idx + NQ - 1)
template <bool MASKED>
__device__ __forceinline__ void detect_item(const NmTailArgs &a, const NmTailOct &oc, const NmTailFrame &fr, int slot, int f,
                                            int idx, nmdet::DetectSmem *sm)
{
    const int groups = oc.nseg * ((oc.oh + nmdet::DET_ROWS - 1) / nmdet::DET_ROWS);
    const int want = NQ * idx + (int)(threadIdx.x >> 8);
    const bool active = want < groups;
    const int blk = active ? want : groups - 1;
    nmdet::DetectView v;
#pragma unroll
    for (int p = 0; p < 6; ++p) v.pl[p] = fr.lev[slot][p];
    v.staging = fr.staging[slot]; v.stage_stride = fr.stage_stride[slot]; v.counts = fr.counts[slot];
    v.dense = nullptr;
    v.mask = MASKED ? a.masks[f] : nullptr; v.mask_w = a.mask_w; v.mask_h = a.mask_h;
    v.ow = oc.ow; v.oh = oc.oh; v.peak = a.peak; v.edge = a.edge; v.xper = oc.xper; v.sigma0 = a.sigma0;
    v.num_dogs = a.num_dogs; v.n_blocks = oc.n_blocks; v.nseg = oc.nseg;
    // (Touching the group's 7 rows x 6 levels with all loads in flight first -- the body fetches row by row, one dependent
    // round trip each, from planes other CUs wrote through to memory a moment ago -- was measured: 31 instead of 20 us per item.)
    nmdet::detect_stage_body<false, true, MASKED, false, NQ>(v, blk, sm, active);
}

__global__ __launch_bounds__(NT) void tail_kernel(NmTailArgs a)
{
    // all LDS is dynamic (a static variable would shift the dynamic base off its 16-byte alignment): the items' scratch first,
    // then two control words
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int &s_ticket = reinterpret_cast<int *>(lds + (a.lds_bytes >> 2))[0];
    int &s_flag = reinterpret_cast<int *>(lds + (a.lds_bytes >> 2))[1];
    int &s_err = reinterpret_cast<int *>(lds + (a.lds_bytes >> 2))[2];
    int *const state = a.state;
    const int total = a.n * a.items_per_frame;
    __builtin_amdgcn_s_setprio(2);                          // a latency chain beside the other stream's throughput kernels
    for (;;) {
        // the sticky error word is read ONCE per item, by the thread that draws the ticket: every wave of the workgroup then
        // takes the same branch below (a per-thread load could split the waves between `continue` and an item's barriers)
        if (threadIdx.x == 0) { s_ticket = add_i32_agent(state + 0, 1); s_err = load_i32_agent(state + 2); }
        __syncthreads();
        const int t = s_ticket;
        const bool failed = s_err != 0;
        __syncthreads();
        if (t >= total) break;
        const unsigned long long tr0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        int si = 0;
        while (si + 1 < a.n_seg && t >= a.n * a.seg[si + 1].first_per_frame) ++si;
        const int kind = a.seg[si].kind, slot = a.seg[si].slot, per_frame = a.seg[si].per_frame;
        const int local = t - a.n * a.seg[si].first_per_frame;
        const int f = local / per_frame, idx = local - f * per_frame;
        const NmTailOct &oc = a.oct[slot];
        const NmTailFrame &fr = a.fr[f];
        // what the item waits for (always items with smaller tickets)
        const int *c0 = nullptr, *c1 = nullptr;
        int t0 = 0, t1 = 0;
        if (kind == NM_TAIL_CONV_A) {
            if (slot > 0) { c0 = counter_of(a, f, slot - 1, 0); t0 = a.oct[slot - 1].n_a; }
        } else if (kind == NM_TAIL_CONV_B || kind == NM_TAIL_GRAD) {
            c0 = counter_of(a, f, slot, 0); t0 = oc.n_a;
        } else {                                           // NM_TAIL_DETECT
            c0 = counter_of(a, f, slot, 1); t0 = oc.b_target;
        }
        if (c0 || c1) {
            if (!wait_counters(c0, t0, c1, t1, state + 2, &s_flag)) continue;      // error: drain the tickets without working
        } else if (failed) {
            continue;
        }
        const unsigned long long tr1 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (kind == NM_TAIL_CONV_A || kind == NM_TAIL_CONV_B) {
            int x0 = 0, x1 = oc.ow, y0 = 0, y1 = oc.oh, la = 1, lb = 5;
            if (!oc.whole) {
                const int ty = idx / oc.tiles_x, tx = idx - ty * oc.tiles_x;
                x0 = tx * TW; x1 = min(x0 + TW, oc.ow); y0 = ty * TH; y1 = min(y0 + TH, oc.oh);
                la = (kind == NM_TAIL_CONV_A) ? 1 : 4; lb = (kind == NM_TAIL_CONV_A) ? 3 : 5;
            }
            conv_item(a, oc, fr, f, slot, x0, x1, y0, y1, la, lb, lds,
                      a.trace ? a.trace + 4 * (size_t)total + 12 * (size_t)t : nullptr);
        } else if (kind == NM_TAIL_GRAD) {
            grad_item(oc, fr, slot, idx);
        } else {
            // staging lists and counts go to the scan launch behind this one: plain stores, nothing to publish
            nmdet::DetectSmem *sm = reinterpret_cast<nmdet::DetectSmem *>(lds);
            if (a.any_mask) detect_item<true>(a, oc, fr, slot, f, idx, sm); else detect_item<false>(a, oc, fr, slot, f, idx, sm);
        }
        __syncthreads();                                   // the LDS is reused by the next item
        if (a.trace && threadIdx.x == 0) {                 // 100 MHz clock: ticket drawn, inputs ready, item done
            unsigned long long *r = a.trace + 4 * (size_t)t;
            r[0] = (unsigned long long)kind | ((unsigned long long)slot << 8) | ((unsigned long long)f << 16) |
                   ((unsigned long long)idx << 24) | ((unsigned long long)blockIdx.x << 48);
            r[1] = tr0; r[2] = tr1; r[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
    // the last workgroup out leaves the state words zero for the next launch (or replay) on these arenas -- the sticky error
    // word included: it moves to state[3], the STATUS of this launch (0 / 1), which the scan launch behind this one turns into
    // num_items = -1 for every frame of the call and nm_sift_arena_tail_status reports to the host; the next launch starts clean
    if (threadIdx.x == 0) s_flag = (add_i32_agent(state + 1, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (s_flag) {
        for (int i = NM_TAIL_STATE_HEAD + threadIdx.x; i < NM_TAIL_STATE_INTS; i += NT) store_i32_agent(state + i, 0);
        if (threadIdx.x == 0) {
            store_i32_agent(state + 3, load_i32_agent(state + 2) != 0 ? 1 : 0);
            store_i32_agent(state + 2, 0);
            store_i32_agent(state + 0, 0); store_i32_agent(state + 1, 0);
        }
    }
}

// The book-keeping scan + ordered gather of the tail octaves, one workgroup per frame walking the octaves in order (the
// book's running count chains them); a launch of its own on the detection stream, behind the detection of octave T - 1 and
// behind the tail launch.
constexpr int SCAN_NT = 256;             // a few hundred counts and keypoints per frame: four waves find a CU at once
__global__ __launch_bounds__(SCAN_NT) void tail_scan_kernel(NmTailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // four waves on the critical path of a single-frame call, sharing their CU with the description kernel of the other
    // stream (16 waves of dense VALU work): without the raised issue priority they take 39 us for what they do alone in 8
    __builtin_amdgcn_s_setprio(3);
    scan_gather_all<SCAN_NT>(a, a.fr[blockIdx.x], blockIdx.x, reinterpret_cast<int *>(lds));
}

}  // namespace

// LDS an item may need: two planes of the largest conv frame, the detection body's scratch, the scan's offsets
// halo = what surrounds the item's rectangle inside the frame: the cumulative radius of the fused levels for a tile in the
// interior, at most PAD (zeros) where the rectangle touches the image border -- a whole plane touches it everywhere
static int conv_frame_bytes(int w, int h, int halo)
{
    const int fw = w + 2 * halo + SLACK, fh = h + 2 * halo + SLACK;
    const int fp = fw | 1;
    return 2 * (((fp * fh) + 3) & ~3) * 4;
}

bool nm_tail_plan(NmTailArgs &a, int width, int height, int num_octaves, int T, const int radii[5])
{
    static const int lds_budget = 144 * 1024;             // of the CU's 160 KB
    a.n_oct = num_octaves - T; a.T = T;
    if (T < 1 || a.n_oct < 1 || a.n_oct > NM_TAIL_MAX_OCT) return false;
    static const int want[5] = {5, 7, 8, 10, 13};           // the SIFT sigmas' radii (sift/siftparams.h:50, pyramidata.cu:108)
    for (int i = 0; i < 5; ++i) {
        if (radii[i] != want[i]) return false;
        a.radii[i] = radii[i];
    }
    const int HA = radii[0] + radii[1] + radii[2], HB = radii[3] + radii[4], HW = HA + HB;
    int lds = NQ * (int)sizeof(nmdet::DetectSmem);
    a.scan_lds_bytes = 0;
    for (int j = 0; j < a.n_oct; ++j) {
        NmTailOct &oc = a.oct[j];
        oc.o = T + j; oc.ow = width >> oc.o; oc.oh = height >> oc.o; oc.xper = (float)(1 << oc.o);
        if (oc.ow < 1 || oc.oh < 1) return false;
        const int whole_bytes = conv_frame_bytes(oc.ow, oc.oh, std::min(HW, PAD));
        // a whole plane as ONE item only while its frame is no larger than a tile's (60 x 33 at 1080p): the launch reserves
        // the largest frame for every workgroup, and what a workgroup of this launch holds, the kernels of other streams on
        // the same CU cannot have
        oc.whole = whole_bytes <= conv_frame_bytes(TW, TH, HB) ? 1 : 0;
        oc.tiles_x = nm_divup(oc.ow, TW); oc.tiles_y = nm_divup(oc.oh, TH);
        oc.n_a = oc.whole ? 1 : oc.tiles_x * oc.tiles_y;
        oc.n_b = oc.whole ? 0 : oc.n_a;
        oc.b_target = oc.whole ? 1 : oc.n_b;
        oc.nseg = nm_divup(oc.ow, NM_DET_SEG_W);
        oc.n_det = nm_divup(oc.nseg * nm_divup(oc.oh, nmdet::DET_ROWS), NQ);
        oc.n_blocks = oc.oh * oc.nseg;
        oc.n_grad = oc.whole ? 3 * nm_divup(oc.oh, GRAD_BAND) : 0;
        oc.decimate = (oc.o + 1 < num_octaves) ? 1 : 0;
        lds = std::max(lds, oc.whole ? whole_bytes : std::max(conv_frame_bytes(TW, TH, HA), conv_frame_bytes(TW, TH, HB)));
        a.scan_lds_bytes += 3 * oc.n_blocks * 4;
    }
    a.scan_lds_bytes += 128 * 4;
    if (a.scan_lds_bytes > 60 * 1024) return false;
    a.lds_bytes = (lds + 255) & ~255;
    if (a.lds_bytes + NM_TAIL_LDS_CTRL > lds_budget) return false;   // the launch asks for the items' scratch + the control words
    // segments in topological order: sorted by the step at which their inputs exist (A of slot j: 3 j + 3, B: 3 j + 5,
    // GRAD: 3 j + 4, DETECT: 3 j + 6); every dependency of a segment has a smaller key
    struct Key { int key, kind, slot, count; } keys[4 * NM_TAIL_MAX_OCT];
    int nk = 0;
    for (int j = 0; j < a.n_oct; ++j) {
        const NmTailOct &oc = a.oct[j];
        keys[nk++] = {3 * j + 3, NM_TAIL_CONV_A, j, oc.n_a};
        if (oc.n_grad) keys[nk++] = {3 * j + 4, NM_TAIL_GRAD, j, oc.n_grad};
        if (oc.n_b) keys[nk++] = {3 * j + 5, NM_TAIL_CONV_B, j, oc.n_b};
        keys[nk++] = {3 * j + 6, NM_TAIL_DETECT, j, oc.n_det};
    }
    for (int i = 1; i < nk; ++i)                         // insertion sort, stable; on equal keys the deeper octave's A first
        for (int k = i; k > 0 && (keys[k].key < keys[k - 1].key ||
                                  (keys[k].key == keys[k - 1].key && keys[k].kind == NM_TAIL_CONV_A && keys[k - 1].kind != NM_TAIL_CONV_A)); --k)
            std::swap(keys[k], keys[k - 1]);
    a.n_seg = nk;
    int first = 0;
    for (int i = 0; i < nk; ++i) {
        a.seg[i].kind = keys[i].kind; a.seg[i].slot = keys[i].slot; a.seg[i].per_frame = keys[i].count;
        a.seg[i].first_per_frame = first;
        first += keys[i].count;
    }
    a.items_per_frame = first;
    return true;
}

int nm_launch_tail(const NmTailArgs &a, hipStream_t stream)
{
    if (a.n <= 0 || a.items_per_frame <= 0) return 0;
    // per call, like the matcher's launches: the attribute belongs to the (function, device) pair and a process may drive
    // several devices; lds_bytes + NM_TAIL_LDS_CTRL is what the launch asks for and what nm_tail_plan budgeted
    NM_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void *>(tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     a.lds_bytes + NM_TAIL_LDS_CTRL));
    // persistent workgroups: about as many as items can run side by side (the widest segment of a frame is its first), never
    // more than one per CU; a workgroup that waits for a counter still holds its CU's LDS
    const int total = a.n * a.items_per_frame;
    const int widest = a.seg[0].per_frame + (a.n_seg > 1 ? a.seg[1].per_frame / 2 : 0);
    // ... and never more than 3/8 of the CUs: a tail workgroup owns its CU (1 024 threads x 128 registers: the whole register file),
    // so what runs BESIDE the launch -- octave T - 1's detection, then orientation + descriptors of the large octaves -- has only the
    // CUs it leaves. Round 6, one 1080p pair at batch 1 (two frames: 216 workgroups uncapped), us per pair at a cap of
    // none / 192 / 160 / 128 / 96 / 64 / 48 / 32: 541 / 532 / 524 / 515 / 509 / 506 / 519 / 584; one frame (108 uncapped) at
    // none / 96 / 64 / 48: 270 / 268 / 276 / 288 (profiles/r06_zz_tail_grid_cap.txt).
    const int grid = std::max(1, std::min(std::min(std::min(total, nm_cu_count()), a.n * widest), nm_cu_count() * 3 / 8));
    hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(NT), a.lds_bytes + NM_TAIL_LDS_CTRL, stream, a);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_launch_tail_scan(const NmTailArgs &a, hipStream_t stream)
{
    if (a.n <= 0 || a.n_oct <= 0) return 0;
    hipLaunchKernelGGL(tail_scan_kernel, dim3(a.n), dim3(SCAN_NT), a.scan_lds_bytes, stream, a);
    NM_LAUNCH_CHECK();
    return 0;
}
