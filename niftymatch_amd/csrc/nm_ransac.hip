// nm_ransac.hip -- RANSAC hypothesis evaluation for gfx950: replaces translation_kernel,
// similarity_transformation_kernel, homography_kernel (kernels/ransac.cu:430-521) and the thrust::max_element +
// copy of the host functions (:523-694).
// The reference runs one thread per hypothesis and lets it scan every point serially. Here the model fit is one
// thread per hypothesis (it is a serial 9x9 Jacobi), but the inlier count is one WAVEFRONT per hypothesis: lanes
// stride over the points (coalesced), ballots are popcounted, no atomics. The best hypothesis is the first maximum.
#include "nm_common.hpp"
#include "nm_ransac_math.hpp"
#include "../../include/nm_abi.h"

namespace {

template <int MODEL>
__global__ __launch_bounds__(64) void ransac_model_kernel(const float *__restrict__ sx, const float *__restrict__ sy,
                                                         const float *__restrict__ dx, const float *__restrict__ dy,
                                                         const int *__restrict__ rand_list, int iterations,
                                                         float *__restrict__ homographies, int *__restrict__ inliers)
{
    constexpr int NS = (MODEL == 0) ? 1 : (MODEL == 1) ? 2 : 4;
    const int it = blockIdx.x * 64 + threadIdx.x;
    if (it >= iterations) return;
    float H[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = 0.f;
    int ri[NS];
    bool dup = false;
#pragma unroll
    for (int a = 0; a < NS; ++a) ri[a] = rand_list[(size_t)it * NS + a];
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int b = a + 1; b < NS; ++b) dup = dup || (ri[a] == ri[b]);
    if (!dup) {
        float px[NS], py[NS], qx[NS], qy[NS];
#pragma unroll
        for (int a = 0; a < NS; ++a) { px[a] = sx[ri[a]]; py[a] = sy[ri[a]]; qx[a] = dx[ri[a]]; qy[a] = dy[ri[a]]; }
        if (MODEL == 0) nmr_fit_translation(px, py, qx, qy, H);
        else if (MODEL == 1) nmr_fit_similarity(px, py, qx, qy, H);
        else nmr_fit_homography(px, py, qx, qy, H);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) homographies[(size_t)it * 9 + k] = H[k];
    inliers[it] = dup ? -1 : 0;                    // -1 marks a skipped hypothesis for the counting pass
}

__global__ __launch_bounds__(256) void ransac_inlier_kernel(const float *__restrict__ sx, const float *__restrict__ sy,
                                                           const float *__restrict__ dx, const float *__restrict__ dy,
                                                           int n, const float *__restrict__ homographies, int iterations,
                                                           float thr, int *__restrict__ inliers)
{
    const int lane = threadIdx.x & 63;
    for (int it = blockIdx.x * 4 + (threadIdx.x >> 6); it < iterations; it += gridDim.x * 4) {
        if (inliers[it] < 0) {                     // repeated sample index: count 0, as the reference's untouched zero
            if (lane == 0) inliers[it] = 0;
            continue;
        }
        float H[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] = homographies[(size_t)it * 9 + k];
        int cnt = 0;
        for (int i = lane; i < n; i += 64) {
            const float x = sx[i];
            const bool in = (x >= 0.f) && nmr_is_inlier(H, x, sy[i], dx[i], dy[i], thr);
            cnt += in ? 1 : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0) inliers[it] = cnt;
    }
}

// first maximum of inliers[0..iterations) + copy of its homography (thrust::max_element semantics)
__global__ __launch_bounds__(1024) void ransac_select_kernel(const int *__restrict__ inliers, int iterations,
                                                            const float *__restrict__ homographies,
                                                            float *__restrict__ H_best, int *__restrict__ position)
{
    __shared__ int s_val[1024], s_pos[1024];
    int bv = -0x7fffffff, bp = 0x7fffffff;
    for (int i = threadIdx.x; i < iterations; i += 1024) {
        const int v = inliers[i];
        if (v > bv) { bv = v; bp = i; }            // ascending i per thread: keeps the first maximum
    }
    s_val[threadIdx.x] = bv; s_pos[threadIdx.x] = bp;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            const int ov = s_val[threadIdx.x + d], op = s_pos[threadIdx.x + d];
            if (ov > s_val[threadIdx.x] || (ov == s_val[threadIdx.x] && op < s_pos[threadIdx.x])) {
                s_val[threadIdx.x] = ov; s_pos[threadIdx.x] = op;
            }
        }
        __syncthreads();
    }
    const int pos = s_pos[0];
    if (threadIdx.x < 9) H_best[threadIdx.x] = homographies[(size_t)pos * 9 + threadIdx.x];
    if (threadIdx.x == 0 && position) *position = pos;
}

}  // namespace

extern "C" int nm_ransac_f32(int model, const float *src_x, const float *src_y, const float *dst_x, const float *dst_y,
                             int num_pts, const int *rand_list, int iterations, float inlier_threshold,
                             float *homographies, int *inliers, float *H_best, int *d_position, void *stream)
{
    if (model < 0 || model > 2 || iterations <= 0 || num_pts <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = nm_stream(stream);
    const dim3 grid(nm_divup(iterations, 64));
    if (model == 0)
        hipLaunchKernelGGL(ransac_model_kernel<0>, grid, dim3(64), 0, st, src_x, src_y, dst_x, dst_y, rand_list, iterations, homographies, inliers);
    else if (model == 1)
        hipLaunchKernelGGL(ransac_model_kernel<1>, grid, dim3(64), 0, st, src_x, src_y, dst_x, dst_y, rand_list, iterations, homographies, inliers);
    else
        hipLaunchKernelGGL(ransac_model_kernel<2>, grid, dim3(64), 0, st, src_x, src_y, dst_x, dst_y, rand_list, iterations, homographies, inliers);
    NM_LAUNCH_CHECK();
    const int blocks = min(nm_divup(iterations, 4), 2048);
    hipLaunchKernelGGL(ransac_inlier_kernel, dim3(blocks), dim3(256), 0, st, src_x, src_y, dst_x, dst_y, num_pts,
                       homographies, iterations, inlier_threshold, inliers);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(1024), 0, st, inliers, iterations, homographies, H_best, d_position);
    NM_LAUNCH_CHECK();
    return 0;
}
