// nm_keypoint.hpp -- argument blocks of the frame driver's per-octave detection kernels (internal, not C ABI).
#pragma once
#include "nm_common.hpp"

struct NmDetectArgs {
    const float *dog[5];   // DoG planes 0..4 of the octave
    int ow, oh;
    float peak, edge, xper, sigma0;
    int num_dogs;
    float *staging;        // 3 x stage_stride float4: block b of level l writes its survivors at [l][b*256 ...]
    size_t stage_stride;   // in float4 elements
    int *counts;           // 3 x n_blocks
    int n_blocks;          // oh * nseg units: a unit is one 256-pixel segment of one row, units in raster order
    int nseg;              // ceil(ow / 256)
};

struct NmScanArgs {
    const int *counts;
    int *offsets;
    int n_blocks;
    int octave;
    int capacity;
    NmFrameBook *book;
    int *d_num_items;      // optional mirror of book->num_items
};

struct NmGatherArgs {
    const float *staging;
    size_t stage_stride;
    const int *counts;
    const int *offsets;
    int n_blocks;
    int octave;
    const NmFrameBook *book;
    float *kpts;           // output-ordered float4 list
};

int nm_launch_detect_octave(const NmDetectArgs &d, const NmScanArgs &s, const NmGatherArgs &g, hipStream_t stream);
