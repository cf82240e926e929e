// nm_warp.hip -- SURVEY.md 8(f) rows N3 (undistortion map + resample) and N4 (perspective warp, mosaicking blend):
// kernels/undistort.cu:7-64, kernels/resample.cu:7-248. The reference samples through CUDA texture objects created by
// utils/cudatex2D.cu:13-19 (border addressing, linear filter, unnormalised coordinates, normalised-float reads). CDNA4
// code objects launched through HIP have no such object here: the sampler is written out (border test, weights
// quantised to 1/256 like the texture unit's 9-bit fixed point, fixed left-to-right sum). Same operation sequence as
// oracle/nmo_warp.h. All kernels are one pixel per lane on 64 x 4 tiles: coordinate streams are read coalesced, the 4
// texel taps are gathers that hit L2 (a warp of neighbouring pixels touches neighbouring texels).
#include "nm_common.hpp"
#include "../../include/nm_abi.h"

namespace {

struct Tex { const void *data; int w, h, fmt; };

__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <int FMT>
__device__ __forceinline__ float texel(const Tex &t, int i, int j, int ch)
{
    if (i < 0 || i >= t.w || j < 0 || j >= t.h) return 0.f;
    const size_t p = (size_t)j * t.w + i;
    if (FMT == NM_TEX_F32) return ((const float *)t.data)[p];
    if (FMT == NM_TEX_U8N) return (float)((const unsigned char *)t.data)[p] / 255.0f;
    return (float)((const unsigned char *)t.data)[4 * p + ch] / 255.0f;
}

__device__ __forceinline__ bool tex_setup(const Tex &t, float x, float y, int &i, int &j, float w[4])
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    if (!(xb >= -1.0f && xb < (float)t.w && yb >= -1.0f && yb < (float)t.h)) return false;
    const float fi = __builtin_floorf(xb), fj = __builtin_floorf(yb);
    const float a = __builtin_floorf((xb - fi) * 256.0f + 0.5f) * 0.00390625f;
    const float b = __builtin_floorf((yb - fj) * 256.0f + 0.5f) * 0.00390625f;
    i = (int)fi; j = (int)fj;
    w[0] = (1.0f - a) * (1.0f - b); w[1] = a * (1.0f - b); w[2] = (1.0f - a) * b; w[3] = a * b;
    return true;
}

template <int FMT>
__device__ __forceinline__ float tex2d(const Tex &t, float x, float y)
{
    int i, j; float w[4];
    if (!tex_setup(t, x, y, i, j, w)) return 0.f;
    return ((w[0] * texel<FMT>(t, i, j, 0) + w[1] * texel<FMT>(t, i + 1, j, 0)) + w[2] * texel<FMT>(t, i, j + 1, 0)) +
           w[3] * texel<FMT>(t, i + 1, j + 1, 0);
}

__device__ __forceinline__ float tex2d_any(const Tex &t, float x, float y)
{
    return t.fmt == NM_TEX_F32 ? tex2d<NM_TEX_F32>(t, x, y) : tex2d<NM_TEX_U8N>(t, x, y);
}

// uchar4 texture: one 4-byte load per tap, the four channels share the weights
__device__ __forceinline__ void tex2d_u8x4(const Tex &t, float x, float y, float out[4])
{
    int i, j; float w[4];
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (!tex_setup(t, x, y, i, j, w)) return;
    float tap[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ii = i + (k & 1), jj = j + (k >> 1);
        uchar4 p = make_uchar4(0, 0, 0, 0);
        if (ii >= 0 && ii < t.w && jj >= 0 && jj < t.h) p = ((const uchar4 *)t.data)[(size_t)jj * t.w + ii];
        tap[k][0] = (float)p.x / 255.0f; tap[k][1] = (float)p.y / 255.0f;
        tap[k][2] = (float)p.z / 255.0f; tap[k][3] = (float)p.w / 255.0f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = ((w[0] * tap[0][c] + w[1] * tap[1][c]) + w[2] * tap[2][c]) + w[3] * tap[3][c];
}

__device__ __forceinline__ void project(const float *m, float x, float y, float &xp, float &yp)
{
    const float a = fmaf_(m[0], x, m[1] * y) + m[2];
    const float b = fmaf_(m[3], x, m[4] * y) + m[5];
    const float s = fmaf_(m[6], x, m[7] * y) + m[8];
    xp = a / s; yp = b / s;
}

__device__ __forceinline__ void invert3x3(const float *t, float *inv)
{
    const float c0 = fmaf_(t[4], t[8], -(t[7] * t[5]));
    const float c1 = fmaf_(t[3], t[8], -(t[5] * t[6]));
    const float c2 = fmaf_(t[3], t[7], -(t[4] * t[6]));
    const float det = fmaf_(t[2], c2, fmaf_(t[0], c0, -(t[1] * c1)));
    const float invdet = 1.0f / det;
    inv[0] = c0 * invdet;
    inv[1] = fmaf_(t[2], t[7], -(t[1] * t[8])) * invdet;
    inv[2] = fmaf_(t[1], t[5], -(t[2] * t[4])) * invdet;
    inv[3] = fmaf_(t[5], t[6], -(t[3] * t[8])) * invdet;
    inv[4] = fmaf_(t[0], t[8], -(t[2] * t[6])) * invdet;
    inv[5] = fmaf_(t[3], t[2], -(t[0] * t[5])) * invdet;
    inv[6] = fmaf_(t[3], t[7], -(t[6] * t[4])) * invdet;
    inv[7] = fmaf_(t[6], t[1], -(t[0] * t[7])) * invdet;
    inv[8] = fmaf_(t[0], t[4], -(t[3] * t[1])) * invdet;
}

__global__ __launch_bounds__(256) void undistort_map_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           size_t n, const float *__restrict__ cam,
                                                           const float *__restrict__ dist, float *__restrict__ u,
                                                           float *__restrict__ v)
{
    const float k1 = dist[0], k2 = dist[1], k3 = dist[2];
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; p < n; p += stride) {
        float a = (x[p] - cx) / fx;
        float b = (y[p] - cy) / fy;
        const float r2 = fmaf_(a, a, b * b);
        const float r4 = r2 * r2, r6 = r4 * r2;
        const float poly = fmaf_(k3, r6, fmaf_(k2, r4, fmaf_(k1, r2, 1.0f)));
        u[p] = fmaf_(a * poly, fx, cx);
        v[p] = fmaf_(b * poly, fy, cy);
    }
}

__global__ __launch_bounds__(256) void resample_f32_kernel(Tex t, const float *__restrict__ x, const float *__restrict__ y,
                                                          size_t n, float *__restrict__ out)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; p < n; p += stride) out[p] = tex2d_any(t, x[p] + 0.5f, y[p] + 0.5f) * 255.9999f;
}

__global__ __launch_bounds__(256) void resample_mask_kernel(unsigned char *__restrict__ result, Tex t,
                                                           const float *__restrict__ x, const float *__restrict__ y,
                                                           size_t n, float lower_limit)
{
    size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; p < n; p += stride) {
        const float r = tex2d_any(t, x[p] + 0.5f, y[p] + 0.5f);
        result[p] = (r <= lower_limit) ? (unsigned char)0 : (unsigned char)(r * 255.999f);
    }
}

// apply_perspective[_inverse] + resample_2D<uchar4> in one launch: the coordinates are written AND used from registers
__global__ __launch_bounds__(256) void perspective_resample_kernel(uchar4 *__restrict__ result, Tex t, int cols, int rows,
                                                                  float *__restrict__ x_pos, float *__restrict__ y_pos,
                                                                  const float *__restrict__ mat3x3, int inverse)
{
    __shared__ float m[9];
    if (threadIdx.x == 0) {
        float src[9];
        for (int k = 0; k < 9; ++k) src[k] = mat3x3[k];
        if (inverse) { float inv[9]; invert3x3(src, inv); for (int k = 0; k < 9; ++k) m[k] = inv[k]; }
        else for (int k = 0; k < 9; ++k) m[k] = src[k];
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const size_t p = (size_t)y * cols + x;
    float xp, yp;
    project(m, (float)x, (float)y, xp, yp);
    x_pos[p] = xp; y_pos[p] = yp;
    float r[4];
    tex2d_u8x4(t, xp + 0.5f, yp + 0.5f, r);
    result[p] = make_uchar4((unsigned char)(r[0] * 255.9999f), (unsigned char)(r[1] * 255.9999f),
                            (unsigned char)(r[2] * 255.9999f), (unsigned char)(r[3] * 255.9999f));
}

__global__ __launch_bounds__(256) void transform_blend_kernel(uchar4 *__restrict__ canvas, int cw, int ch, Tex frame,
                                                             int nw, int nh, const float *__restrict__ mat3x3, int tx,
                                                             int ty, Tex mask, float *__restrict__ canvas_wts, Tex wts)
{
    __shared__ float m[9];
    if (threadIdx.x < 9) m[threadIdx.x] = mat3x3[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int px = x + tx, py = y + ty;
    if (x >= nw || y >= nh || px < 0 || px >= cw || py < 0 || py >= ch) return;
    float xp, yp;
    project(m, (float)x, (float)y, xp, yp);
    if (xp >= (float)frame.w || yp >= (float)frame.h) return;
    const float u = xp + 0.5f, v = yp + 0.5f;
    if (tex2d_any(mask, u, v) <= 0.5f) return;
    const float nwt = tex2d_any(wts, u, v);
    float r[4];
    tex2d_u8x4(frame, u, v, r);
    const size_t idx = (size_t)py * cw + px;
    const float cwt = canvas_wts[idx];
    uchar4 c;
    if (cwt == 0) {
        c = make_uchar4((unsigned char)(r[0] * 255.9999f), (unsigned char)(r[1] * 255.9999f),
                        (unsigned char)(r[2] * 255.9999f), 255);
        canvas_wts[idx] = nwt;
    } else {
        const uchar4 cur = canvas[idx];
        const float sum = cwt + nwt;
        c.x = (unsigned char)(fmaf_(r[0] * nwt, 255.9999f, (float)cur.x * cwt) / sum);
        c.y = (unsigned char)(fmaf_(r[1] * nwt, 255.9999f, (float)cur.y * cwt) / sum);
        c.z = (unsigned char)(fmaf_(r[2] * nwt, 255.9999f, (float)cur.z * cwt) / sum);
        c.w = 255;
        canvas_wts[idx] = sum;
    }
    canvas[idx] = c;
}

inline int stream_blocks(size_t n)
{
    size_t b = (n + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b ? b : 1));
}
inline bool scalar_fmt(int f) { return f == NM_TEX_U8N || f == NM_TEX_F32; }

}  // namespace

extern "C" {

int nm_undistort_map_f32(const float *x, const float *y, size_t cols, size_t rows, const float *camera_matrix,
                         const float *distortion_coeffs, float *u, float *v, void *stream)
{
    const size_t n = cols * rows;
    if (!n) return 0;
    hipLaunchKernelGGL(undistort_map_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream), x, y, n,
                       camera_matrix, distortion_coeffs, u, v);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_resample_undistort_f32(const void *tex, int tex_width, int tex_height, int tex_format, const float *x,
                              const float *y, size_t cols, size_t rows, float *undistorted, void *stream)
{
    const size_t n = cols * rows;
    if (!n) return 0;
    if (!scalar_fmt(tex_format) || tex_width <= 0 || tex_height <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(resample_f32_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream),
                       Tex{tex, tex_width, tex_height, tex_format}, x, y, n, undistorted);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_resample_mask_u8(unsigned char *result, const void *tex, int tex_width, int tex_height, int tex_format, int cols,
                        int rows, const float *x_pos, const float *y_pos, float threshold, void *stream)
{
    if (cols <= 0 || rows <= 0) return 0;
    if (!scalar_fmt(tex_format) || tex_width <= 0 || tex_height <= 0) return (int)hipErrorInvalidValue;
    const size_t n = (size_t)cols * rows;
    hipLaunchKernelGGL(resample_mask_kernel, dim3(stream_blocks(n)), dim3(256), 0, nm_stream(stream), result,
                       Tex{tex, tex_width, tex_height, tex_format}, x_pos, y_pos, n, threshold);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_resample_perspective_u8x4(unsigned char *result, const unsigned char *tex, int tex_width, int tex_height, int cols,
                                 int rows, float *x_pos, float *y_pos, const float *mat3x3, int inverse, void *stream)
{
    if (cols <= 0 || rows <= 0) return 0;
    if (tex_width <= 0 || tex_height <= 0) return (int)hipErrorInvalidValue;
    dim3 grid(nm_divup(cols, 64), nm_divup(rows, 4));
    hipLaunchKernelGGL(perspective_resample_kernel, grid, dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<uchar4 *>(result), Tex{tex, tex_width, tex_height, NM_TEX_U8X4N}, cols, rows, x_pos,
                       y_pos, mat3x3, inverse);
    NM_LAUNCH_CHECK();
    return 0;
}

int nm_transform_blend(unsigned char *canvas, int cw, int ch, const unsigned char *frame, int fw, int fh, int nw, int nh,
                       const float *mat3x3, int tx, int ty, const void *frame_mask, int mask_format, float *canvas_wts,
                       const void *frame_wts, int wts_format, void *stream)
{
    if (nw <= 0 || nh <= 0) return 0;
    if (!scalar_fmt(mask_format) || !scalar_fmt(wts_format) || fw <= 0 || fh <= 0) return (int)hipErrorInvalidValue;
    dim3 grid(nm_divup(nw, 64), nm_divup(nh, 4));
    hipLaunchKernelGGL(transform_blend_kernel, grid, dim3(256), 0, nm_stream(stream), reinterpret_cast<uchar4 *>(canvas),
                       cw, ch, Tex{frame, fw, fh, NM_TEX_U8X4N}, nw, nh, mat3x3, tx, ty,
                       Tex{frame_mask, fw, fh, mask_format}, canvas_wts, Tex{frame_wts, fw, fh, wts_format});
    NM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
