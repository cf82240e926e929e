// ransac.cpp -- host side of the RANSAC entry points (reference: src/gpu/kernels/ransac.cu:523-694): filter the valid
// points, draw the sample indices on the host with std::mt19937 + std::uniform_int_distribution, evaluate all
// hypotheses on the device, return the one with the most inliers.
#include "../ransac.h"

#include <iostream>
#include <random>
#include <vector>

#include "../../../include/nm_abi.h"
#include "../device_vector.h"
#include "../exception.h"

static unsigned int g_ransac_seed = 0;

extern "C" void nm_ransac_seed(unsigned int seed) { g_ransac_seed = seed; }

static bool ransac_impl(int model, int samples, size_t min_points, float *src_x, float *src_y, float *dst_x,
                        float *dst_y, const int src_size, float inlier_threshold, int iterations, float *homography,
                        hipStream_t stream)
{
    if (src_size <= 0 || iterations <= 0) return false;
    std::vector<float> host_x((size_t)src_size);
    nm_check((int)hipStreamSynchronize(stream), "RANSAC launch failed");
    nm_check((int)hipMemcpy(host_x.data(), src_x, host_x.size() * sizeof(float), hipMemcpyDeviceToHost), "RANSAC D2H");
    std::vector<int> valid;
    for (int i = 0; i < src_size; ++i)
        if (host_x[i] >= 0) valid.push_back(i);
    if (valid.size() < min_points) {                       // "Not enough points" (ransac.cu:537-541)
        std::cout << valid.size() << std::endl;
        std::cout.flush();
        return false;
    }
    std::random_device seeder;
    std::mt19937 engine(g_ransac_seed ? g_ransac_seed : seeder());
    std::uniform_int_distribution<int> dist(0, (int)valid.size() - 1);
    std::vector<int> rand_list((size_t)iterations * samples);
    for (size_t c = 0; c < rand_list.size(); ++c) rand_list[c] = valid[dist(engine)];

    nm::device_vector<int> d_rand(rand_list);
    nm::device_vector<float> d_h((size_t)iterations * 9, 0.f);
    nm::device_vector<int> d_inl((size_t)iterations, 0);
    nm_check(nm_ransac_f32(model, src_x, src_y, dst_x, dst_y, src_size, d_rand.data(), iterations, inlier_threshold,
                           d_h.data(), d_inl.data(), homography, nullptr, stream),
             "RANSAC launch failed");
    nm_check((int)hipStreamSynchronize(stream), "RANSAC launch failed");      // the temporaries die here
    return true;
}

bool ransac_translation(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int,
                        float inlier_threshold, int iterations, float *homography, hipStream_t stream)
{
    return ransac_impl(0, 1, 2, src_x, src_y, dst_x, dst_y, src_size, inlier_threshold, iterations, homography, stream);
}

bool ransac_similarity(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int,
                       float inlier_threshold, int iterations, float *homography, hipStream_t stream)
{
    return ransac_impl(1, 2, 2, src_x, src_y, dst_x, dst_y, src_size, inlier_threshold, iterations, homography, stream);
}

bool ransac_homography(float *src_x, float *src_y, float *dst_x, float *dst_y, const int src_size, const int,
                       float inlier_threshold, int iterations, float *homography, hipStream_t stream)
{
    return ransac_impl(2, 4, 4, src_x, src_y, dst_x, dst_y, src_size, inlier_threshold, iterations, homography, stream);
}
