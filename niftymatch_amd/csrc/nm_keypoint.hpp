// nm_keypoint.hpp -- argument blocks of the frame driver's per-octave detection kernels (internal, not C ABI).
#pragma once
#include "nm_common.hpp"

// Per-frame pointers are arrays over the frames of a batched call (NM_MAX_BATCH); blockIdx selects the frame.
struct NmDetectArgs {
    int n;                                  // frames
    // The planes a frame's detection reads: the octave's DoG planes 0..4, or -- from_levels -- its Gaussian levels 0..5 (DoG i =
    // lev[i + 1] - lev[i] is then formed on the fly, the same fp32 subtraction the pyramid kernel stores). 64 x 11 pointers do
    // not fit the 4 KB of kernel arguments, and a pointer table in device memory puts a dependent load at the head of every
    // workgroup (measured: +30 % on the kernel): the frame driver passes ONE pointer per frame and a stride. API path (one
    // frame, caller's planes): plane0[0] = NULL and the pointers travel in api_planes.
    const float *plane0[NM_MAX_BATCH];      // frame driver: the frame's plane 0; plane p = plane0 + p * plane_stride (the arena
    size_t plane_stride;                    // allocates an octave's levels / DoG planes as one block)
    const float *api_planes[6];
    int from_levels;
    int ow, oh;
    float peak, edge, xper, sigma0;
    int num_dogs;
    float *staging[NM_MAX_BATCH];   // 3 x stage_stride float4: block b of level l writes its survivors at [l][b*256 ...]
    size_t stage_stride;   // in float4 elements
    int *counts[NM_MAX_BATCH];      // 3 x n_blocks
    int n_blocks;          // oh * nseg units: a unit is one 256-pixel segment of one row, units in raster order
    int nseg;              // ceil(ow / 256)
    // API path (nm_find_keypoints3_f32, one frame): every pixel of the three dense float4 maps is written (an accepted
    // keypoint or -1) instead of the staging lists; optional full-resolution mask (keypoint.cu:204-224)
    float *dense[3];
    const float *mask;
    int mask_w, mask_h;
    // frame driver (nm_sift_arena_set_mask): optional full-resolution mask per frame, mask_w x mask_h as above
    const float *masks[NM_MAX_BATCH];
    int any_mask;
    // API path (nm_find_keypoints3_reset_f32): the launch also resets entries [ow * oh, reset_end[l]) of dense[l] to -1 -- what an
    // earlier, larger octave left behind the region this octave writes -- with fill_blocks extra workgroups behind the
    // det_blocks that detect (the reference resets the whole maps with thrust::fill per octave, siftfunctions.cu:120-121)
    size_t reset_end[3];
    int det_blocks, fill_blocks;
    // frame driver: unit groups (raster order: group row, segment) dealt to the XCDs in BANDS of xcd_band consecutive groups (0:
    // blockIdx.x is the group). The hardware deals workgroups round-robin over the 8 XCDs; with grid.x = 8 * xcd_band, workgroup b
    // is slot b / 8 of XCD b % 8 and takes group (b % 8) * xcd_band + b / 8: the cache lines two neighbouring segments share (a
    // segment is 248 pixels: 992 B, 7.75 lines) and the halo rows of two neighbouring group rows are then fetched by ONE L2, and
    // every XCD gets the same number of groups to within one.
    int xcd_band, group_rows;
};

struct NmScanArgs {
    int n;
    const int *counts[NM_MAX_BATCH];
    int *offsets[NM_MAX_BATCH];
    int n_blocks;
    int octave;
    int capacity;
    NmFrameBook *book[NM_MAX_BATCH];
    int *d_num_items[NM_MAX_BATCH];      // optional mirror of book->num_items
};

struct NmGatherArgs {
    int n;
    const float *staging[NM_MAX_BATCH];
    size_t stage_stride;
    const int *counts[NM_MAX_BATCH];
    const int *offsets[NM_MAX_BATCH];
    int n_blocks;
    int octave;
    int capacity;          // upper bound of any level's output count
    const NmFrameBook *book[NM_MAX_BATCH];
    float *kpts[NM_MAX_BATCH];           // output-ordered float4 lists
};

static_assert(sizeof(NmDetectArgs) <= 4096 && sizeof(NmScanArgs) <= 4096 && sizeof(NmGatherArgs) <= 4096,
              "kernel arguments are limited to 4 KB: lower NM_MAX_BATCH");

int nm_launch_detect_octave(const NmDetectArgs &d, const NmScanArgs &s, const NmGatherArgs &g, hipStream_t stream);
