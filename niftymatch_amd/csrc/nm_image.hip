// nm_image.hip -- the element-wise stages on either side of the SIFT path (N3 front end, N1's gather):
// BGRA -> gray / channel extract / channel put / alpha fill (kernels/bgra_2_gray.cu:9-113), float -> uchar cast
// (kernels/cast.cu:8-36), uchar4 decimation (kernels/downsample.cu:6-32) and the match gather align_points
// (kernels/ransac.cu:29-59). All are HBM-bound streaming kernels: one 16-byte or 4-byte access per lane, grid-stride.
#include "nm_common.hpp"
#include "../../include/nm_abi.h"

namespace {

__device__ __forceinline__ float gray_of(uchar4 p)
{
    // 0.07*B + 0.72*G + 0.21*R in double (the literals are double), narrowed to float (bgra_2_gray.cu:16);
    // contraction written out: fma(0.21, R, fma(0.07, B, 0.72*G))
    const double b = (double)(int)p.x, g = (double)(int)p.y, r = (double)(int)p.z;
    return (float)__builtin_fma(0.21, r, __builtin_fma(0.07, b, 0.72 * g));
}

__global__ __launch_bounds__(256) void grayscale_kernel(const uchar4 *__restrict__ bgra, float *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = gray_of(bgra[i]);
}

__global__ __launch_bounds__(256) void extract_channel_kernel(const uchar4 *__restrict__ bgra, float *__restrict__ out,
                                                             size_t n, int channel)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const uchar4 p = bgra[i];
        const unsigned char v = channel == 0 ? p.x : channel == 1 ? p.y : channel == 2 ? p.z : p.w;
        out[i] = (float)v;
    }
}

__global__ __launch_bounds__(256) void put_channel_kernel(uchar4 *__restrict__ bgra, const float *__restrict__ in,
                                                         size_t n, int channel)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        uchar4 p = bgra[i];
        const unsigned char v = (unsigned char)in[i];
        if (channel == 0) p.x = v;
        else if (channel == 1) p.y = v;
        else if (channel == 2) p.z = v;
        else if (channel == 3) p.w = 255;          // the reference overwrites alpha with 255 (bgra_2_gray.cu:68-69)
        bgra[i] = p;
    }
}

__global__ __launch_bounds__(256) void set_alpha_kernel(uchar4 *__restrict__ bgra, size_t n, unsigned char val)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) { uchar4 p = bgra[i]; p.w = val; bgra[i] = p; }
}

__global__ __launch_bounds__(256) void cast_f32_u8_kernel(const float *__restrict__ src, unsigned char *__restrict__ dst,
                                                         size_t n, unsigned char max_val)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float v = src[i];
        dst[i] = (max_val != 0 && v >= (float)max_val) ? max_val : (unsigned char)v;    // cast.cu:17-19
    }
}

__global__ __launch_bounds__(256) void downsample2_u8x4_kernel(uchar4 *__restrict__ result, int rw, int rh,
                                                              const uchar4 *__restrict__ source, int sw)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= rw || y >= rh) return;
    result[(size_t)y * rw + x] = source[(size_t)(y * 2) * sw + (x * 2)];
}

__global__ __launch_bounds__(256) void align_points_kernel(const float *__restrict__ sx, const float *__restrict__ sy,
                                                          const float *__restrict__ dx, const float *__restrict__ dy,
                                                          float *__restrict__ csx, float *__restrict__ csy,
                                                          float *__restrict__ cdx, float *__restrict__ cdy,
                                                          const int *__restrict__ matches, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = matches[i];
    if (m != -1) { csx[i] = sx[i]; csy[i] = sy[i]; cdx[i] = dx[m]; cdy[i] = dy[m]; }
    else { csx[i] = -1; csy[i] = -1; cdx[i] = -1; cdy[i] = -1; }
}

inline int stream_blocks(size_t n)
{
    size_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}

}  // namespace

extern "C" {

int nm_grayscale_f32(const unsigned char *bgra, float *output, int width, int height, void *stream)
{
    const size_t n = (size_t)width * height;
    if (!n) return 0;
    hipLaunchKernelGGL(grayscale_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<const uchar4 *>(bgra), output, n);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_extract_channel_f32(const unsigned char *bgra, float *output, int width, int height, int channel, void *stream)
{
    const size_t n = (size_t)width * height;
    if (!n || channel < 0 || channel > 3) return 0;       // other channel numbers: the reference writes nothing
    hipLaunchKernelGGL(extract_channel_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<const uchar4 *>(bgra), output, n, channel);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_put_channel_f32(unsigned char *bgra, const float *input, int width, int height, int channel, void *stream)
{
    const size_t n = (size_t)width * height;
    if (!n || channel < 0 || channel > 3) return 0;
    hipLaunchKernelGGL(put_channel_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<uchar4 *>(bgra), input, n, channel);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_set_alpha_to_const(unsigned char *bgra, int width, int height, unsigned char val, void *stream)
{
    const size_t n = (size_t)width * height;
    if (!n) return 0;
    hipLaunchKernelGGL(set_alpha_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<uchar4 *>(bgra), n, val);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_cast_f32_u8(const float *src, size_t cols, size_t rows, unsigned char *dst, unsigned char max_val, void *stream)
{
    const size_t n = cols * rows;
    if (!n) return 0;
    hipLaunchKernelGGL(cast_f32_u8_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream), src, dst, n, max_val);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_downsample2_u8x4(unsigned char *result, int rw, int rh, const unsigned char *source, int sw, int sh, void *stream)
{
    (void)sh;
    if (rw <= 0 || rh <= 0) return 0;
    dim3 grid(nm_divup(rw, 64), nm_divup(rh, 4));
    hipLaunchKernelGGL(downsample2_u8x4_kernel, grid, dim3(256), 0, nm_stream(stream), reinterpret_cast<uchar4 *>(result),
                       rw, rh, reinterpret_cast<const uchar4 *>(source), sw);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_align_points(const float *src_x, const float *src_y, const float *dst_x, const float *dst_y, float *c_src_x,
                    float *c_src_y, float *c_dst_x, float *c_dst_y, const int *matches, int num_pts, void *stream)
{
    if (num_pts <= 0) return 0;
    hipLaunchKernelGGL(align_points_kernel, dim3(nm_divup(num_pts, 256)), dim3(256), 0, nm_stream(stream), src_x, src_y,
                       dst_x, dst_y, c_src_x, c_src_y, c_dst_x, c_dst_y, matches, num_pts);
    NM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
