// nm_tail.hpp -- argument blocks of the octave-tail kernel (internal, not C ABI): ONE persistent launch that runs, for every
// frame of a batched call, everything the frame driver does for the octaves >= T (480 x 270 and below at 1080p): the five
// Gaussian levels with their gradient planes and the decimated seed of the next octave, the three-level extrema search with
// sub-pixel refinement, the per-unit counts, the book-keeping scan and the ordered gather (reference orchestration:
// sift/siftfunctions.cu:42-181, sift/pyramidata.cu:84-91). See nm_tail.hip.
#pragma once
#include "nm_common.hpp"

#define NM_TAIL_MAX_OCT 8            // octaves a tail launch can cover (a 1080p frame has 4 of them at T = 2)

struct NmTailOct {
    int o;                           // octave index
    int ow, oh;
    float xper;
    int whole;                       // 1: ONE item computes all five levels of the whole plane in LDS; 0: 64 x 32 output tiles
    int tiles_x, tiles_y;
    int n_a;                         // items that produce levels 1..3 (+ the decimated seed): tiles, or 1
    int n_b;                         // items that produce levels 4..5: tiles, or 0 (the whole-plane item continues)
    int b_target;                    // value of the b_done counter when levels 4..5 are complete (n_b, or 1)
    int nseg, n_det;                 // detection: NM_DET_SEG_W-pixel segments per row, unit groups (DET_ROWS rows x one segment)
    int n_blocks;                    // units = oh * nseg
    int n_grad;                      // whole planes: gradient items (3 levels x row bands); tiles compute it inline
    int decimate;                    // level 3 seeds octave o + 1
};

// One frame's buffers for the tail octaves, indexed by SLOT (octave - T). Kept by value inside the kernel arguments (the
// tail serves calls of at most NM_TAIL_MAX_FRAMES frames): an item reads its plane pointers from the kernel-argument segment
// instead of paying a dependent memory round trip for a device-resident table first.
#define NM_TAIL_MAX_FRAMES 2
struct NmTailFrame {
    float *lev[NM_TAIL_MAX_OCT][6];
    float *grad[NM_TAIL_MAX_OCT];
    float *staging[NM_TAIL_MAX_OCT];
    size_t stage_stride[NM_TAIL_MAX_OCT];
    int *counts[NM_TAIL_MAX_OCT];
    NmFrameBook *book;
};

enum { NM_TAIL_CONV_A = 0, NM_TAIL_CONV_B = 1, NM_TAIL_DETECT = 2, NM_TAIL_SCAN = 3, NM_TAIL_GRAD = 4 };
struct NmTailSeg {
    int kind, slot;                  // slot = octave - T
    int per_frame;                   // items of this segment per frame
    int first_per_frame;             // items per frame in all earlier segments (first ticket = n * first_per_frame)
};

struct NmTailArgs {
    int n, n_oct, T;
    NmTailOct oct[NM_TAIL_MAX_OCT];
    NmTailFrame fr[NM_TAIL_MAX_FRAMES];
    float *kpts[NM_TAIL_MAX_FRAMES];
    int *d_num_items[NM_TAIL_MAX_FRAMES];
    const float *masks[NM_TAIL_MAX_FRAMES];
    int any_mask, mask_w, mask_h;
    float peak, edge, sigma0;
    int num_dogs, capacity;
    const float *taps[5];
    int radii[5];
    int *state;                      // NM_TAIL_STATE_INTS ints, all zero between launches (the launch cleans up after itself)
    int n_seg;
    NmTailSeg seg[4 * NM_TAIL_MAX_OCT];
    int items_per_frame;
    int lds_bytes, scan_lds_bytes;
    unsigned long long *trace;       // diagnostic (NM_TAIL_TRACE=1 at arena creation) or NULL: 4 words per ticket
};
static_assert(sizeof(NmTailArgs) <= 4096, "kernel arguments are limited to 4 KB");

// state words: [0] ticket, [1] workgroups that have left, [2] sticky error of the RUNNING launch (a wait that hit its spin
// limit; the items behind it are drained without working), [3] status of the LAST COMPLETED launch (the last workgroup out
// moves [2] there and zeroes [2]), then per (frame, slot) four counters. [0], [1], [2] and the counters are zero between launches.
#define NM_TAIL_LDS_CTRL 16          // bytes of control words behind the items' LDS scratch (ticket, flag, error)
#define NM_TAIL_STATE_HEAD 4
#define NM_TAIL_STATE_INTS (NM_TAIL_STATE_HEAD + NM_TAIL_MAX_FRAMES * NM_TAIL_MAX_OCT * 4)

// Host side: plans the launch for one geometry (false: this geometry / these radii are not covered, use the per-octave launches).
bool nm_tail_plan(NmTailArgs &a, int width, int height, int num_octaves, int T, const int radii[5]);
int nm_launch_tail(const NmTailArgs &a, hipStream_t stream);          // Gaussian levels, gradients, detection (caller's stream)
int nm_launch_tail_scan(const NmTailArgs &a, hipStream_t stream);     // book-keeping scans + gathers (detection stream)
