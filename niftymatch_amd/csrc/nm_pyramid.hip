// nm_pyramid.hip -- Gaussian pyramid stages for gfx950: separable Gaussian (+ fused DoG), decimation, subtract,
// gradient. Replaces kernels/convolution.cu:16-159, kernels/downsample.cu:6-29, kernels/cudamath.cu:26-80 of the
// reference. All kernels are HBM/L2 streaming stencils; arithmetic order is fixed by the fp spec:
//   conv: taps k = -r..r, sum = fma(x[k], w[r-k], sum) starting from +0, rows first, then columns, zero padding.
#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "../../include/nm_abi.h"

using nmfp::fma32;

// ------------------------------------------------------------------------------------------------------------
// Fused separable Gaussian. A workgroup (256 threads = 4 waves) walks over 64 x TH output tiles:
//   fetch    global -> VGPRs: the (TH+2R) x (64+2RA) input tile of the NEXT tile is requested (float4, all loads in
//            flight together) before the current tile is computed, so HBM latency hides behind the FMA phases
//   phase 1  VGPRs -> LDS (zero outside the image)
//   phase 2  row pass LDS -> LDS: every thread makes 8 consecutive outputs of one row from registers
//   phase 3  column pass LDS -> global: every thread makes TH/4 vertical outputs of one column; a wave writes
//            whole 256-B row segments. DoG = output - input centre comes from the LDS tile for free, and so does
//            the gradient (magnitude, angle) of the INPUT level (kernels/cudamath.cu:38-54), whose 4-neighbourhood is
//            inside the staged halo: levels 1..3 get their gradients from the launch that blurs them into level+1.
// Tiles are dealt so that workgroups sharing an XCD (blockIdx % 8) walk one contiguous band of the image: halo rows
// re-read by vertical neighbours hit that XCD's L2. R is a template parameter: tap loops unroll, windows live in VGPRs.
template <int R, int TH, bool WRITE_BUF, bool WRITE_DOG, bool WRITE_GRAD, bool VEC>
__global__ __launch_bounds__(TH * 8) void conv_sep_kernel(float *__restrict__ result, const float *__restrict__ image,
                                                      float *__restrict__ buffer, float *__restrict__ dog,
                                                      float2 *__restrict__ grad, int width, int height,
                                                      const float *__restrict__ taps, int tiles_x, int ntiles)
{
    constexpr int TW = 64;
    constexpr int RA = (R + 3) & ~3;              // halo rounded up to 4 columns: 16-byte aligned row segments
    constexpr int IN_W = TW + 2 * RA;             // columns staged in LDS (image x = x0 - RA + c)
    constexpr int IN_P = IN_W + 4;                // row pitch: multiple of 4 (b128 reads) + 4 (bank skew)
    constexpr int OFF = RA - R;                   // first column the row pass reads
    constexpr int ROWS = TH + 2 * R;
    constexpr int NT = 2 * R + 1;
    constexpr int V_PER_ROW = IN_W / 4;
    constexpr int NE = VEC ? ROWS * V_PER_ROW : ROWS * IN_W;     // staged elements (float4 or float)
    constexpr int NTH = TH * 8;                   // 256 threads for 64 x 32 tiles, 512 for 64 x 64
    constexpr int PER = (NE + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) float s_in[ROWS * IN_P];
    __shared__ __attribute__((aligned(16))) float s_mid[ROWS * TW];

    const int tid = threadIdx.x;
    float w[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) w[i] = taps[i];

    // tile schedule: XCD-contiguous bands
    const int nxcd = 8;
    const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;   // gridDim.x % 8 == 0
    const int band = (ntiles + nxcd - 1) / nxcd;
    const int t_begin = xcd * band, t_end = min(t_begin + band, ntiles);

    float4 pf4[VEC ? PER : 1];
    float pf1[VEC ? 1 : PER];
    auto fetch = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NTH * i;
            if (VEC) {
                const int row = idx / V_PER_ROW, c4 = idx - row * V_PER_ROW;
                const int gy = y0 - R + row, gx = x0 - RA + 4 * c4;
                pf4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < NE && gy >= 0 && gy < height && gx >= 0 && gx < width)
                    pf4[i] = *reinterpret_cast<const float4 *>(image + (size_t)gy * width + gx);
            } else {
                const int row = idx / IN_W, c = idx - row * IN_W;
                const int gy = y0 - R + row, gx = x0 - RA + c;
                pf1[i] = 0.f;
                if (idx < NE && gy >= 0 && gy < height && gx >= 0 && gx < width) pf1[i] = image[(size_t)gy * width + gx];
            }
        }
    };

    int tile = t_begin + slot;
    if (tile < t_end) fetch(tile);
    for (; tile < t_end; tile += per_xcd) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;

        // phase 1: registers -> LDS
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + NTH * i;
            if (VEC) {
                const int row = idx / V_PER_ROW, c4 = idx - row * V_PER_ROW;
                if (idx < NE) *reinterpret_cast<float4 *>(&s_in[row * IN_P + 4 * c4]) = pf4[i];
            } else {
                const int row = idx / IN_W, c = idx - row * IN_W;
                if (idx < NE) s_in[row * IN_P + c] = pf1[i];
            }
        }
        __syncthreads();
        if (tile + per_xcd < t_end) fetch(tile + per_xcd);        // next tile's loads fly during phases 2 and 3

        // phase 2: rows
        {
            const int xc = tid & 7;
            for (int row = tid >> 3; row < ROWS; row += NTH / 8) {
                float v[8 + 2 * RA];                  // 16-byte aligned window; the taps use v[OFF .. OFF + 8 + 2R)
                const float *p = &s_in[row * IN_P + xc * 8];
#pragma unroll
                for (int j = 0; j < 8 + 2 * RA; ++j) v[j] = p[j];
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
                for (int k = -R; k <= R; ++k) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = fma32(v[OFF + i + R + k], w[R - k], o[i]);
                }
                float *q = &s_mid[row * TW + xc * 8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = o[i];
                if (WRITE_BUF) {
                    const int gy = y0 - R + row;
                    if (row >= R && row < R + TH && gy < height) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int gx = x0 + xc * 8 + i;
                            if (gx < width) buffer[(size_t)gy * width + gx] = o[i];
                        }
                    }
                }
            }
        }
        __syncthreads();

        // phase 3: columns
        {
            constexpr int NY = TH / (NTH / 64);       // 8 vertical outputs per thread
            const int x = tid & 63, yg = tid >> 6;
            float v[NY + 2 * R];
            const float *p = &s_mid[(yg * NY) * TW + x];
#pragma unroll
            for (int j = 0; j < NY + 2 * R; ++j) v[j] = p[j * TW];
            float o[NY];
#pragma unroll
            for (int i = 0; i < NY; ++i) o[i] = 0.f;
#pragma unroll
            for (int k = -R; k <= R; ++k) {
#pragma unroll
                for (int i = 0; i < NY; ++i) o[i] = fma32(v[i + R + k], w[R - k], o[i]);
            }
            const int gx = x0 + x;
            if (gx < width) {
#pragma unroll
                for (int i = 0; i < NY; ++i) {
                    const int yy = yg * NY + i;
                    const int gy = y0 + yy;
                    if (gy < height) {
                        if (result) result[(size_t)gy * width + gx] = o[i];
                        const float *cin = &s_in[(yy + R) * IN_P + x + RA];
                        if (WRITE_DOG) dog[(size_t)gy * width + gx] = o[i] - cin[0];
                        if (WRITE_GRAD) {        // gradient of the INPUT level: its tile (+halo) is already in LDS
                            float g = 0.f, r = 0.f;
                            if (gx >= 1 && gx < width - 1 && gy >= 1 && gy < height - 1) {
                                const float dx = cin[1] - cin[-1], dy = cin[IN_P] - cin[-IN_P];
                                g = (float)(0.5 * (double)__builtin_sqrtf(fma32(dx, dx, dy * dy)));
                                if (g != 0.0f)
                                    r = nmfp::mod_2pi_f((float)((double)nmfp::atan2f_spec(dy, dx) + nmfp::TWO_PI_D));
                            }
                            grad[(size_t)gy * width + gx] = make_float2(g, r);
                        }
                    }
                }
            }
        }
        __syncthreads();                              // s_in / s_mid are rewritten by the next tile
    }
}

// Any-radius fallback (two passes through global memory, one thread per pixel). Same arithmetic order.
__global__ __launch_bounds__(256) void conv_rows_generic(float *__restrict__ out, const float *__restrict__ in,
                                                        int width, int height, const float *__restrict__ taps, int r)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const float *row = in + (size_t)y * width;
    float sum = 0.f;
    for (int k = -r; k <= r; ++k) {
        const int xx = x + k;
        const float v = (xx >= 0 && xx < width) ? row[xx] : 0.f;
        sum = fma32(v, taps[r - k], sum);
    }
    out[(size_t)y * width + x] = sum;
}
__global__ __launch_bounds__(256) void conv_cols_generic(float *__restrict__ out, const float *__restrict__ in,
                                                        float *__restrict__ dog, const float *__restrict__ orig,
                                                        int width, int height, const float *__restrict__ taps, int r)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    float sum = 0.f;
    for (int k = -r; k <= r; ++k) {
        const int yy = y + k;
        const float v = (yy >= 0 && yy < height) ? in[(size_t)yy * width + x] : 0.f;
        sum = fma32(v, taps[r - k], sum);
    }
    out[(size_t)y * width + x] = sum;
    if (dog) dog[(size_t)y * width + x] = sum - orig[(size_t)y * width + x];
}

template <int R, bool VEC, int TH>
static int launch_conv_rvt(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                           int height, const float *taps, hipStream_t stream)
{
    const int tiles_x = nm_divup(width, 64), tiles_y = nm_divup(height, TH);
    const int ntiles = tiles_x * tiles_y;
    // one tile per workgroup up to the chip's residency; grid is a multiple of 8 (XCDs). (Measured on MI355X: more
    // tiles per workgroup with register prefetch is slower than more resident workgroups.)
    const int blocks = ((ntiles + 7) / 8) * 8;
    dim3 grid(blocks);
    float2 *g2 = reinterpret_cast<float2 *>(grad);
#define NM_CONV_LAUNCH(BUF, DOG, GRAD)                                                                              \
    hipLaunchKernelGGL((conv_sep_kernel<R, TH, BUF, DOG, GRAD, VEC>), grid, dim3(TH * 8), 0, stream, result, image, \
                       buffer, dog, g2, width, height, taps, tiles_x, ntiles)
    if (buffer) {
        if (dog || grad || TH != 32) return (int)hipErrorInvalidValue;   // the API path never asks for the fused outputs
        if (TH == 32) NM_CONV_LAUNCH(true, false, false);
    } else if (dog && grad) {
        NM_CONV_LAUNCH(false, true, true);
    } else if (dog) {
        NM_CONV_LAUNCH(false, true, false);
    } else if (grad) {
        return (int)hipErrorInvalidValue;
    } else {
        NM_CONV_LAUNCH(false, false, false);
    }
#undef NM_CONV_LAUNCH
    NM_LAUNCH_CHECK();
    return 0;
}

template <int R, bool VEC>
static int launch_conv_rv(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                          int height, const float *taps, hipStream_t stream)
{
    // 64 x 32 tiles with 256 threads, or 64 x 64 tiles with 512 threads (less halo re-reading and row-pass redundancy)
    // once the image has enough tiles to fill the chip that way.
    if (VEC && !buffer && (long)width * height >= 256L * 64 * 64)
        return launch_conv_rvt<R, VEC, VEC ? 64 : 32>(result, image, buffer, dog, grad, width, height, taps, stream);
    return launch_conv_rvt<R, VEC, 32>(result, image, buffer, dog, grad, width, height, taps, stream);
}

template <int R>
static int launch_conv_r(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                         int height, const float *taps, hipStream_t stream)
{
    const bool vec = (width % 4 == 0) && ((reinterpret_cast<uintptr_t>(image) & 15) == 0);
    return vec ? launch_conv_rv<R, true>(result, image, buffer, dog, grad, width, height, taps, stream)
               : launch_conv_rv<R, false>(result, image, buffer, dog, grad, width, height, taps, stream);
}

int nm_launch_convolve(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                       int height, const float *taps, int radius, hipStream_t stream)
{
    if (width <= 0 || height <= 0) return 0;
    if (radius < 0) return (int)hipErrorInvalidValue;
    switch (radius) {
        case 5: return launch_conv_r<5>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 7: return launch_conv_r<7>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 8: return launch_conv_r<8>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 10: return launch_conv_r<10>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 12: return launch_conv_r<12>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 13: return launch_conv_r<13>(result, image, buffer, dog, grad, width, height, taps, stream);
        case 16: return launch_conv_r<16>(result, image, buffer, dog, grad, width, height, taps, stream);
        default: break;
    }
    // generic radius: the row pass needs a real intermediate. Without a caller buffer there is none to use.
    if (!buffer) return (int)hipErrorInvalidValue;
    dim3 grid(nm_divup(width, 64), nm_divup(height, 4));
    hipLaunchKernelGGL(conv_rows_generic, grid, dim3(256), 0, stream, buffer, image, width, height, taps, radius);
    NM_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_cols_generic, grid, dim3(256), 0, stream, result, buffer, dog, image, width, height, taps, radius);
    NM_LAUNCH_CHECK();
    if (grad) {
        NmGradBatch b{};
        b.src[0] = image; b.dst[0] = grad; b.n = 1;
        return nm_launch_gradient_batch(b, width, height, stream);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void downsample2_kernel(float *__restrict__ result, int rw, int rh,
                                                         const float *__restrict__ source, int sw)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= rw || y >= rh) return;
    result[(size_t)y * rw + x] = source[(size_t)(y * 2) * sw + (x * 2)];
}

__global__ __launch_bounds__(256) void subtract_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                      float *__restrict__ C, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) C[i] = A[i] - B[i];
}

// gradient: g = 0.5*sqrt(dx^2+dy^2), theta = mod_2pi(atan2(dy,dx) + 2pi) in (0, 2pi], 0 when g == 0; border = (0,0).
__global__ __launch_bounds__(256) void gradient_kernel(NmGradBatch b, int width, int height)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const float *__restrict__ src = b.src[blockIdx.z];
    float2 *__restrict__ dst = reinterpret_cast<float2 *>(b.dst[blockIdx.z]);
    float g = 0.f, r = 0.f;
    if (x >= 1 && x < width - 1 && y >= 1 && y < height - 1) {
        const size_t c = (size_t)y * width + x;
        const float dx = src[c + 1] - src[c - 1];
        const float dy = src[c + width] - src[c - width];
        g = (float)(0.5 * (double)__builtin_sqrtf(fma32(dx, dx, dy * dy)));
        if (g != 0.0f) r = nmfp::mod_2pi_f((float)((double)nmfp::atan2f_spec(dy, dx) + nmfp::TWO_PI_D));
    }
    dst[(size_t)y * width + x] = make_float2(g, r);
}

int nm_launch_gradient_batch(const NmGradBatch &b, int width, int height, hipStream_t stream)
{
    if (width <= 0 || height <= 0 || b.n <= 0) return 0;
    dim3 grid(nm_divup(width, 64), nm_divup(height, 4), b.n);
    hipLaunchKernelGGL(gradient_kernel, grid, dim3(256), 0, stream, b, width, height);
    NM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
extern "C" {

int DivUp(int a, int b) { return ((a % b) != 0) ? (a / b + 1) : (a / b); }
int DivDown(int a, int b) { return a / b; }
int AlignUp(int a, int b) { return ((a % b) != 0) ? (a - a % b + b) : a; }
int AlignDown(int a, int b) { return a - a % b; }

int nm_convolve_f32(float *result, const float *image, float *buffer, int width, int height, const float *kernel,
                    int kernel_radius, void *stream)
{
    if (!result || !image || !buffer || !kernel) return (int)hipErrorInvalidValue;
    return nm_launch_convolve(result, image, buffer, nullptr, nullptr, width, height, kernel, kernel_radius, nm_stream(stream));
}

int nm_downsample2_f32(float *result, int rw, int rh, const float *source, int sw, int sh, void *stream)
{
    (void)sh;
    if (rw <= 0 || rh <= 0) return 0;
    dim3 grid(nm_divup(rw, 64), nm_divup(rh, 4));
    hipLaunchKernelGGL(downsample2_kernel, grid, dim3(256), 0, nm_stream(stream), result, rw, rh, source, sw);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_subtract_f32(const float *A, const float *B, float *C, int width, int height, void *stream)
{
    const size_t n = (size_t)width * height;
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(subtract_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream), A, B, C, n);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_gradient_f32(const float *source, float *result, int width, int height, void *stream)
{
    NmGradBatch b{};
    b.src[0] = source; b.dst[0] = result; b.n = 1;
    return nm_launch_gradient_batch(b, width, height, nm_stream(stream));
}

}  // extern "C"
