// nm_client.cpp -- example client of the drop-in C++ API (see include/nm_client.h): the frame loop an application
// written against NiftyMatch owns, using only the public headers.
#include "../../../include/nm_client.h"

#include <cmath>
#include <cstring>
#include <exception>
#include <iostream>
#include <vector>

#include "../convolution.h"
#include "../downsample.h"
#include "../ransac.h"
#include "../siftfunctions.h"

extern "C" int nm_client_detect_describe(const float *gray, int width, int height, int capacity, float *desc, float *x,
                                         float *y)
{
    try {
        SiftParams params(width, height);
        PyramidData py(params);
        SiftData out(capacity);
        const size_t npix = (size_t)width * height;
        nm::device_vector<float> d_gray(std::vector<float>(gray, gray + npix));

        convolve<float>(py._octave[0].data(), d_gray.data(), py._buffer.data(), width, height, py._base_kernel.data(),
                        py._base_radius);
        for (int o = 0; o < params._num_octaves; ++o) {
            const int ow = width >> o, oh = height >> o;
            if (o > 0)   // level 3 has twice the base sigma: it seeds the next octave
                downsample_by_2<float>(py._octave[0].data(), ow, oh, py._octave[3].data(), width >> (o - 1),
                                       height >> (o - 1));
            for (int i = 1; i < py._num_octaves; ++i)
                convolve<float>(py._octave[i].data(), py._octave[i - 1].data(), py._buffer.data(), ow, oh,
                                py._kernels[i - 1].data(), py._kernel_radii[i - 1]);
            compute_dog(py, ow, oh);
            compute_gradients(py, params, ow, oh);
            compute_keypoints(py, params, o, ow, oh);
            compute_orientations(py, params, o, ow, oh);
            compute_descriptors(py, params, o, ow, oh, out);
        }
        const int n = out._num_items;
        if (n > 0) {
            std::vector<float> h = out._desc.to_host();
            std::memcpy(desc, h.data(), (size_t)n * 128 * sizeof(float));
            h = out._x.to_host(); std::memcpy(x, h.data(), (size_t)n * sizeof(float));
            h = out._y.to_host(); std::memcpy(y, h.data(), (size_t)n * sizeof(float));
        }
        return n;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}

extern "C" int nm_client_match(const float *A, int nA, const float *B, int nB, float *distance, int *result,
                               float ambiguity)
{
    try {
        SiftData a(nA), b(nB);
        a._desc = nm::device_vector<float>(std::vector<float>(A, A + (size_t)nA * 128));
        b._desc = nm::device_vector<float>(std::vector<float>(B, B + (size_t)nB * 128));
        a._match_indexes = nm::device_vector<int>(std::vector<int>(result, result + nA));
        a._num_items = nA; b._num_items = nB;
        nm::device_vector<float> d_dist(distance ? (size_t)nA * nB : 0);
        compute_sift_matches(&a, &b, distance ? d_dist.data() : nullptr, ambiguity);
        std::vector<int> r = a._match_indexes.to_host();
        std::memcpy(result, r.data(), (size_t)nA * sizeof(int));
        if (distance) {
            std::vector<float> h = d_dist.to_host();
            std::memcpy(distance, h.data(), h.size() * sizeof(float));
        }
        return 0;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}

extern "C" int nm_client_ransac(int model, const float *src_x, const float *src_y, const float *dst_x, const float *dst_y,
                                int n, float inlier_threshold, int iterations, unsigned int seed, float *H)
{
    try {
        nm::device_vector<float> sx(std::vector<float>(src_x, src_x + n)), sy(std::vector<float>(src_y, src_y + n));
        nm::device_vector<float> dx(std::vector<float>(dst_x, dst_x + n)), dy(std::vector<float>(dst_y, dst_y + n));
        nm::device_vector<float> h(9, 0.f);
        nm_ransac_seed(seed);
        bool ok = false;
        if (model == 0) ok = ransac_translation(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        else if (model == 1) ok = ransac_similarity(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        else ok = ransac_homography(sx.data(), sy.data(), dx.data(), dy.data(), n, n, inlier_threshold, iterations, h.data());
        nm_ransac_seed(0);
        std::vector<float> hh = h.to_host();
        std::memcpy(H, hh.data(), 9 * sizeof(float));
        return ok ? 1 : 0;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return -1;
    }
}
