// nm_describe.hpp -- argument block of the frame driver's orientation/descriptor kernels (internal, not C ABI).
#pragma once
#include "nm_common.hpp"

struct NmOctGeom {
    const float *grad;   // float2 planes of gradient levels 0..2 of this octave, stride ow*oh
    int ow, oh;
    float xper;
};

struct NmDescribeArgs {
    NmOctGeom geom[20];
    int num_octaves;
    int num_dogs;
    const NmFrameBook *book;
    const float *kpts;   // float4, output order
    float *orients;      // float2, output order
    float *desc;         // capacity x 128
    float *x, *y;
};

int nm_launch_frame_describe(const NmDescribeArgs &a, hipStream_t stream);
