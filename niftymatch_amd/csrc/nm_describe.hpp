// nm_describe.hpp -- argument block of the frame driver's orientation/descriptor kernels (internal, not C ABI).
#pragma once
#include "nm_common.hpp"

struct NmOctGeom {
    int ow, oh;
    float xper;
};

// Per-frame pointers are arrays over the frames of a batched call; blockIdx.y selects the frame.
struct NmDescribeArgs {
    int n;                                       // frames
    NmOctGeom geom[20];
    const float *grad0[NM_MAX_BATCH];            // per frame: the arena's gradient block; octave o's three float2 planes (gradient
    size_t grad_off[20];                         // levels 0..2, stride ow*oh) start grad_off[o] floats into it (64 x 20 pointers
                                                 // do not fit the 4 KB of kernel arguments; a table in device memory would put a
                                                 // dependent load in front of every keypoint)
    int num_octaves;
    int num_dogs;
    int o_begin, o_end;                          // this launch covers the keypoints of octaves [o_begin, o_end)
    const NmFrameBook *book[NM_MAX_BATCH];
    const float *kpts[NM_MAX_BATCH];   // float4, output order
    float *orients[NM_MAX_BATCH];      // float2, output order
    float *desc[NM_MAX_BATCH];         // capacity x 128
    float *x[NM_MAX_BATCH], *y[NM_MAX_BATCH];
};

static_assert(sizeof(NmDescribeArgs) <= 4096, "kernel arguments are limited to 4 KB: lower NM_MAX_BATCH");

int nm_launch_frame_describe(const NmDescribeArgs &a, hipStream_t stream);
