// siftparams_dump.cpp -- TEST INFRASTRUCTURE. Prints every field of SiftParams(width, height) for a list of geometries as
// JSON, floats as their IEEE-754 bit patterns. The header is chosen on the command line:
//   -DSIFTPARAMS_HEADER='"/root/reference/src/gpu/sift/siftparams.h"'   the REFERENCE's own header, compiled unmodified
//                                                                       (plain C++: <cmath>, <vector>; the one part of
//                                                                       the reference's hot path that builds here)
//   -DSIFTPARAMS_HEADER='"../niftymatch_amd/nm/siftparams.h"'           the product's drop-in header
// The reference build's output is committed as tests/golden/siftparams_ref.json (a fixture: data, not source); the CPU
// tests hold the oracle's nmo_sift_params and the product header against it (tests/test_siftparams_pinned.py).
// No reference source is copied: this file only names the public fields (sift/siftparams.h:54-98).
#include SIFTPARAMS_HEADER

#include <cstdio>
#include <cstring>

static unsigned bits(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main()
{
    static const int geo[][2] = {{1920, 1080}, {640, 480}, {128, 96}, {160, 120}, {3840, 2160}, {1280, 720}, {256, 192},
                                 {320, 240}, {60, 33}, {31, 500}, {1916, 1076}, {1, 1}};
    const int n = (int)(sizeof(geo) / sizeof(geo[0]));
    std::printf("[\n");
    for (int g = 0; g < n; ++g) {
        const SiftParams p(geo[g][0], geo[g][1]);
        std::printf(" {\"width\": %d, \"height\": %d, \"num_octaves\": %d, \"num_dog_levels\": %d, \"level_max\": %d, "
                    "\"level_min\": %d, \"sigma_d_0\": %u, \"sigma_k\": %u, \"sigma_0\": %u, \"sigma_n\": %u, "
                    "\"base_smooth\": %u, \"peak_threshold\": %u, \"edge_threshold\": %u, \"sigmas\": [",
                    p._width, p._height, p._num_octaves, p._num_dog_levels, p._level_max, p._level_min, bits(p._sigma_d_0),
                    bits(p._sigma_k), bits(p._sigma_0), bits(p._sigma_n), bits(p._base_smooth), bits(p._peak_threshold),
                    bits(p._edge_threshold));
        for (size_t i = 0; i < p._sigmas.size(); ++i) std::printf("%s%u", i ? ", " : "", bits(p._sigmas[i]));
        std::printf("]}%s\n", g + 1 < n ? "," : "");
    }
    std::printf("]\n");
    return 0;
}
