// nm_pyramid.hip -- Gaussian pyramid stages for gfx950: separable Gaussian (+ fused DoG), decimation, subtract,
// gradient. Replaces kernels/convolution.cu:16-159, kernels/downsample.cu:6-29, kernels/cudamath.cu:26-80 of the
// reference. All kernels are HBM/L2 streaming stencils; arithmetic order is fixed by the fp spec:
//   conv: taps k = -r..r, sum = fma(x[k], w[r-k], sum) starting from +0, rows first, then columns, zero padding.
#include <cstdlib>

#include "nm_common.hpp"
#include "nm_fpspec.hpp"
#include "nm_grad_dev.hpp"
#include "nm_pk_dev.hpp"
#include "../../include/nm_abi.h"

using nmfp::fma32;
using namespace nmgrad;
using namespace nmpk;

int nm_cu_count()
{
    static int cache[64];                       // 0 = not read yet; benign race: every thread computes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cache[dev] = n;
    }
    return cache[dev];
}

int nm_xcd_count()
{
    const int x = nm_cu_count() / 32;
    return x < 1 ? 1 : (x > 8 ? 8 : x);
}

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
                                                      const float *__restrict__ taps, int tiles_x, int ntiles, int nxcd)
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
    const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, per_xcd = gridDim.x / nxcd;   // gridDim.x % nxcd == 0
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

// ------------------------------------------------------------------------------------------------------------
// Packed-math variant of the fused separable Gaussian (the frame driver's path: no row-pass output, width % 4 == 0).
// The launches that also produce the gradient are VALU-bound (~190 VALU instructions per output pixel in the kernel
// above), so this version is organised around v_pk_fma_f32 (two fp32 FMAs per lane per issue) without changing a single
// rounding: every output is still  sum = fma(x[k], w[r-k], sum)  from +0 in k order.
//   * the input tile is staged ROW-PAIR INTERLEAVED: LDS holds (row 2p, row 2p+1) of one column side by side, so a
//     b128 read yields two aligned (rowA, rowB) register pairs and the row pass of TWO rows runs as packed FMAs at
//     every tap shift (adjacent-column packing would need unaligned register pairs at odd shifts);
//   * the column pass packs two adjacent columns: a b64 read of the row-pass result is the aligned pair;
//   * lanes of a quarter wave read LDS at strides chosen bank-conflict-free (pitch = IN_W + 2 column pairs);
//   * the gradient uses a correctly rounded sqrt built from v_rsq_f32 + two exact-residual corrections (validated
//     against the IEEE expansion for every float of its domain by nm_selftest_sqrt), the single-subtraction form of
//     mod_2pi, and 0.5f*x for (float)(0.5*(double)x) (exact scaling).

namespace {

__device__ __forceinline__ v2f pk_fma(v2f a, float w, v2f c)
{
    const v2f ww = {w, w};
    return __builtin_elementwise_fma(a, ww, c);
}

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float c) { return (v2f){c, c}; }

// num / den by the instruction sequence of the IEEE expansion (rcp, one refinement, two quotient corrections) without its
// range scaling and special-case fix-up: identical result whenever neither is needed -- den in [2^-49, 2^49] and num = 0
// or |num| >= 2^-100 (below that the exact residuals leave the normal range). Smaller numerators may give a different
// quotient, which grad_pair tolerates (see there). Checked against `/` by nm_selftest_sqrt.
__device__ __forceinline__ v2f div_pair(v2f num, v2f den)
{
    v2f rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    rc = fma2(fma2(-den, rc, splat(1.0f)), rc, rc);
    v2f q = num * rc;
    q = fma2(fma2(-den, q, num), rc, q);
    q = fma2(fma2(-den, q, num), rc, q);
    return q;
}

// Branch-free, packed evaluation of grad_of for two pixels. `ok` is false when an input leaves the domain on which the
// shortcuts are proven equal to the generic sequence (dx^2+dy^2 = 0 or in [2^-96, 2^96)); the caller then redoes the
// pixel with grad_of. Inside the domain den >= 2^-48.5, so a numerator below 2^-100 means a quotient below 2^-51; such a
// quotient only ever arises where it is added to pi/2 (c1) or, in binary64, to 2 pi (third range), and is absorbed by
// the rounding there whatever its low bits are (the cancelling numerator ay - ax of the middle range is 0 or >= 2^-72).
__device__ __forceinline__ bool grad_pair(v2f xm, v2f xp, v2f ym, v2f yp, v2f &g, v2f &r)
{
    const v2f dx = xp - xm, dy = yp - ym;
    const v2f s = fma2(dx, dx, dy * dy);
    const bool ok = (s.x < 0x1p96f) && (s.x >= 0x1p-96f || s.x == 0.0f) && (s.y < 0x1p96f) && (s.y >= 0x1p-96f || s.y == 0.0f);
    // magnitude: sqrt_rn_fast_core on both halves
    const v2f e = {__builtin_amdgcn_rsqf(__builtin_fmaxf(s.x, 0x1p-100f)), __builtin_amdgcn_rsqf(__builtin_fmaxf(s.y, 0x1p-100f))};
    const v2f h = e * splat(0.5f);
    v2f y = s * e;
    y = fma2(fma2(-y, y, s), h, y);
    y = fma2(fma2(-y, y, s), h, y);
    g = y * splat(0.5f);
    // angle: nmfp::atan2f_spec with its three argument ranges selected per half
    const v2f ax = {__builtin_fabsf(dx.x), __builtin_fabsf(dx.y)}, ay = {__builtin_fabsf(dy.x), __builtin_fabsf(dy.y)};
    const v2f t1 = ax * splat(2.414213562373095f), t2 = ax * splat(0.4142135623730950f);
    const v2f dif = ay - ax, sum = ay + ax;
    v2f num, den, hi, lo;
#define NM_SEL(F)                                                                                              \
    {                                                                                                          \
        const bool c1 = ay.F > t1.F, c2 = ay.F > t2.F;                                                         \
        num.F = c1 ? -ax.F : (c2 ? dif.F : ay.F);                                                              \
        den.F = c1 ? ay.F : (c2 ? sum.F : ax.F);                                                               \
        hi.F = c1 ? 1.57079637050628662109375f : (c2 ? 0.785398185253143310546875f : 0.0f);                    \
        lo.F = c1 ? -4.37113900018624283e-8f : (c2 ? -2.18556950009312142e-8f : 0.0f);                         \
    }
    NM_SEL(x)
    NM_SEL(y)
#undef NM_SEL
    const v2f z = div_pair(num, den);
    const v2f zz = z * z;
    v2f p = fma2(splat(-0.06459416449069977f), zz, splat(0.10746313631534576f));
    p = fma2(p, zz, splat(-0.14264234900474548f));
    p = fma2(p, zz, splat(0.1999955028295517f));
    p = fma2(p, zz, splat(-0.3333333134651184f));
    p = p * zz;
    p = fma2(p, z, z);
    v2f a = hi + (p + lo);
    const v2f alt = (splat(3.1415927410125732421875f) - a) + splat(-8.74227800037248566e-8f);
    a.x = dx.x < 0.0f ? alt.x : a.x;
    a.y = dx.y < 0.0f ? alt.y : a.y;
    a.x = dy.x < 0.0f ? -a.x : a.x;
    a.y = dy.y < 0.0f ? -a.y : a.y;
    v2f t = {(float)((double)a.x + nmfp::TWO_PI_D), (float)((double)a.y + nmfp::TWO_PI_D)};
    const v2f tw = t - splat(nmfp::TWO_PI_F);
    t.x = t.x > nmfp::TWO_PI_F ? tw.x : t.x;
    t.y = t.y > nmfp::TWO_PI_F ? tw.y : t.y;
    r.x = g.x != 0.0f ? t.x : 0.0f;
    r.y = g.y != 0.0f ? t.y : 0.0f;
    return ok;
}

}  // namespace

// Geometry shared by the packed kernels (64-column tiles / strips, 32 output rows per step).
template <int R>
struct PkGeom {
    static constexpr int TW = 64, TH = 32;
    static constexpr int RA = (R + 3) & ~3;
    static constexpr int IN_W = TW + 2 * RA;             // staged columns (image x = x0 - RA + c)
    static constexpr int IN_P2 = IN_W + 2;               // pitch in column PAIRS-of-rows; (2*IN_P2) % 32 == 4*odd: see above
    static constexpr int ROWS = TH + 2 * R;              // even
    static constexpr int RP = ROWS / 2;                  // row pairs; <= 32
    static constexpr int MID_P = TW + 2;                 // row-pass result pitch: 8-byte aligned rows (b64 accesses). With + 4 and
                                                         // b128 stores the R = 10 tile was 96 bytes over 160 KB / 5 and R = 7 over / 6
    static constexpr int OFF = RA - R;                   // first staged column the taps touch (for output column 0)
    static constexpr int W0 = OFF & ~1;                  // even start of the b128 window
    static constexpr int D = OFF - W0;
    static constexpr int NC = (8 + 2 * R + D + 1) & ~1;  // window columns per task (even)
    static constexpr int V = IN_W / 4;
    static_assert(RP <= 32 && (IN_P2 % 2) == 0, "tile geometry");
};

// global -> registers: row pairs (gy_first + 2 p, gy_first + 2 p + 1), p < npairs, of the IN_W staged columns; zero outside the image
template <int R, int NL>
__device__ __forceinline__ void pk_load_pairs(float4 (&a)[NL], float4 (&b)[NL], const float *__restrict__ image, int npairs,
                                              int gy_first, int x0, int width, int height, int tid)
{
    using G = PkGeom<R>;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = tid + 256 * i;
        const int p = t / G::V, c4 = t - p * G::V;
        const int gy = gy_first + 2 * p, gx = x0 - G::RA + 4 * c4;
        a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        b[i] = a[i];
        if (t < npairs * G::V && gx >= 0 && gx < width) {
            if (gy >= 0 && gy < height) a[i] = *reinterpret_cast<const float4 *>(image + (size_t)gy * width + gx);
            if (gy + 1 >= 0 && gy + 1 < height) b[i] = *reinterpret_cast<const float4 *>(image + (size_t)(gy + 1) * width + gx);
        }
    }
}

// registers -> LDS, row-pair interleaved, into the pairs [pair_first, pair_first + npairs) of s_in
template <int R, int NL>
__device__ __forceinline__ void pk_store_pairs(const float4 (&a)[NL], const float4 (&b)[NL], float *s_in, int npairs,
                                               int pair_first, int tid)
{
    using G = PkGeom<R>;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = tid + 256 * i;
        const int p = t / G::V, c4 = t - p * G::V;
        if (t < npairs * G::V) {
            float4 *q = reinterpret_cast<float4 *>(&s_in[((pair_first + p) * G::IN_P2 + 4 * c4) * 2]);
            q[0] = make_float4(a[i].x, b[i].x, a[i].y, b[i].y);
            q[1] = make_float4(a[i].z, b[i].z, a[i].w, b[i].w);
        }
    }
}

// rows: task = (row pair p, 8 output columns cg*8..): 16 outputs as 8 packed (rowA, rowB) accumulators -> s_mid rows 2p, 2p+1
template <int R, bool WRITE_BUF>
__device__ __forceinline__ void pk_row_task(const float *s_in, float *s_mid, const unsigned long long (&w)[2 * R + 1], int p,
                                            int cg, float *__restrict__ buffer, int x0, int y0, int width, int height)
{
    using G = PkGeom<R>;
    constexpr int NC = G::NC, D = G::D;
    const v2f *src = reinterpret_cast<const v2f *>(&s_in[(p * G::IN_P2 + cg * 8 + G::W0) * 2]);
    v2f win[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) win[j] = src[j];
    v2f o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = pk_fma_s0(win[D + i], w[2 * R]);
#pragma unroll
    for (int t = 1; t <= 2 * R; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = pk_fma_s(win[D + i + t], w[2 * R - t], o[i]);
    }
    float2 *qa = reinterpret_cast<float2 *>(&s_mid[(2 * p) * G::MID_P + cg * 8]);
    float2 *qb = reinterpret_cast<float2 *>(&s_mid[(2 * p + 1) * G::MID_P + cg * 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa[i] = make_float2(o[2 * i].x, o[2 * i + 1].x);
        qb[i] = make_float2(o[2 * i].y, o[2 * i + 1].y);
    }
    if (WRITE_BUF) {     // API path: the row pass of the tile's own rows is the caller's `buffer` (convolution.cu:141-159)
        const int gx = x0 + cg * 8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int rr = 2 * p + half, gy = y0 - R + rr;
            if (rr >= R && rr < R + G::TH && gy < height && gx < width) {
                float4 *q = reinterpret_cast<float4 *>(buffer + (size_t)gy * width + gx);
                q[0] = half ? make_float4(o[0].y, o[1].y, o[2].y, o[3].y) : make_float4(o[0].x, o[1].x, o[2].x, o[3].x);
                if (gx + 4 < width)
                    q[1] = half ? make_float4(o[4].y, o[5].y, o[6].y, o[7].y) : make_float4(o[4].x, o[5].x, o[6].x, o[7].x);
            }
        }
    }
}

// columns + epilogue: task = (column pair xp, 4 output rows): 8 outputs as 4 packed (x, x+1) accumulators; DoG, gradient of
// the input level and the decimated next-octave plane come from the LDS tiles
template <int R, bool WRITE_DOG, bool WRITE_GRAD, bool WRITE_BUF>
__device__ __forceinline__ void pk_cols_epilogue(const float *s_in, const float *s_mid, const unsigned long long (&w)[2 * R + 1],
                                                 int tid, int x0, int y0, int width, int height, float *__restrict__ result,
                                                 float *__restrict__ dog, float2 *__restrict__ grad, float *__restrict__ down)
{
    using G = PkGeom<R>;
    constexpr int RA = G::RA, IN_P2 = G::IN_P2, MID_P = G::MID_P;
    constexpr int NY = 4;
    const int xp = tid & 31, yg = tid >> 5;
    v2f o[NY];
    {
        const float *src = &s_mid[(yg * NY) * MID_P + 2 * xp];
        v2f win[NY + 2 * R];
#pragma unroll
        for (int j = 0; j < NY + 2 * R; ++j) win[j] = *reinterpret_cast<const v2f *>(src + j * MID_P);
#pragma unroll
        for (int i = 0; i < NY; ++i) o[i] = pk_fma_s0(win[i], w[2 * R]);
#pragma unroll
        for (int t = 1; t <= 2 * R; ++t) {
#pragma unroll
            for (int i = 0; i < NY; ++i) o[i] = pk_fma_s(win[i + t], w[2 * R - t], o[i]);
        }
    }
    const int gx = x0 + 2 * xp;
    if (gx < width) {
        // input-level values around the 2 x 4 patch, from the interleaved tile: rows yy-1 .. yy+4, columns x-1 .. x+2
        const int cc = RA + 2 * xp;
        // One b128 read of the interleaved tile at an even column c returns (row 2p, c), (row 2p+1, c), (row 2p, c+1),
        // (row 2p+1, c+1): the 6 x 4 patch comes from 3-4 row pairs x 3 aligned reads, conflict-free (a lane's address
        // advances by 4 floats with xp), instead of 28 scalar reads that hit 8 banks 4 ways each.
        constexpr int PAR = (R - 1) & 1;                  // parity of the first needed row (yg * NY is even)
        constexpr int J0 = WRITE_GRAD ? 0 : 1, J1 = WRITE_GRAD ? NY + 1 : NY;      // rows j needed
        constexpr int P0 = (PAR + J0) >> 1, P1 = (PAR + J1) >> 1;                  // row pairs relative to rp0
        const int rp0 = (yg * NY + R - 1) >> 1;
        float patch[2 * (P1 + 1)][4];                       // [row relative to 2 rp0][column x-1 .. x+2]
#pragma unroll
        for (int pp = P0; pp <= P1; ++pp) {
            const float *base = &s_in[((rp0 + pp) * IN_P2 + cc) * 2];
            const float4 m = *reinterpret_cast<const float4 *>(base);
            patch[2 * pp][1] = m.x; patch[2 * pp + 1][1] = m.y; patch[2 * pp][2] = m.z; patch[2 * pp + 1][2] = m.w;
            if (WRITE_GRAD) {
                const float4 l = *reinterpret_cast<const float4 *>(base - 4), rr4 = *reinterpret_cast<const float4 *>(base + 4);
                patch[2 * pp][0] = l.z; patch[2 * pp + 1][0] = l.w; patch[2 * pp][3] = rr4.x; patch[2 * pp + 1][3] = rr4.y;
            }
        }
        v2f mid[NY + 2], lft[NY + 2], rgt[NY + 2];      // columns (x, x+1), (x-1, x), (x+1, x+2) of each row
#pragma unroll
        for (int j = 0; j < NY + 2; ++j) {
            if (WRITE_GRAD || (j >= 1 && j <= NY)) mid[j] = (v2f){patch[PAR + j][1], patch[PAR + j][2]};
            if (WRITE_GRAD && j >= 1 && j <= NY) {
                lft[j] = (v2f){patch[PAR + j][0], patch[PAR + j][1]};
                rgt[j] = (v2f){patch[PAR + j][2], patch[PAR + j][3]};
            }
        }
        const unsigned row_bytes = (unsigned)width * 4u;
        const bool in0 = gx >= 1, in1 = gx + 1 < width - 1;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int gy = y0 + yg * NY + i;
            if (gy < height) {
                const unsigned off = (unsigned)gy * row_bytes + (unsigned)gx * 4u;       // < 4 GiB: checked by the host
                if (result) *reinterpret_cast<float2 *>(reinterpret_cast<char *>(result) + off) = make_float2(o[i].x, o[i].y);
                if (!WRITE_BUF && down && !(gy & 1)) {         // next octave's level 0 = every other pixel of every other row (gx is even)
                    const int dw = width >> 1, dx = gx >> 1, dy = gy >> 1;
                    if (dx < dw && dy < (height >> 1)) down[(size_t)dy * dw + dx] = o[i].x;
                }
                if (WRITE_DOG) {
                    const v2f d = o[i] - mid[i + 1];
                    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(dog) + off) = make_float2(d.x, d.y);
                }
                if (WRITE_GRAD) {
                    v2f g, r;
                    const bool ok = grad_pair(lft[i + 1], rgt[i + 1], mid[i], mid[i + 2], g, r);
                    if (__builtin_expect(!ok, 0)) {      // out-of-domain input (denormal-range or huge differences)
                        const float2 a = grad_of(lft[i + 1].x, rgt[i + 1].x, mid[i].x, mid[i + 2].x);
                        const float2 b = grad_of(lft[i + 1].y, rgt[i + 1].y, mid[i].y, mid[i + 2].y);
                        g = (v2f){a.x, b.x};
                        r = (v2f){a.y, b.y};
                    }
                    const bool rowin = gy >= 1 && gy < height - 1;
                    const bool k0 = rowin && in0, k1 = rowin && in1;
                    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(grad) + 2 * (size_t)off) =
                        make_float4(k0 ? g.x : 0.f, k0 ? r.x : 0.f, k1 ? g.y : 0.f, k1 ? r.y : 0.f);
                }
            }
        }
    }
}

#ifdef NM_CONV_STAMPS        // diagnostic builds only (tools/kconv_stamps.py): where a tile's waves spend their cycles
__device__ unsigned long long nm_conv_stamps[4096 * 8];
__device__ unsigned nm_conv_stamp_n;
#endif
template <int R, bool WRITE_DOG, bool WRITE_GRAD, bool WRITE_BUF = false>
__global__ __launch_bounds__(256) void conv_pk_kernel(NmConvBatch batch, int width, int height,
                                                     const float *__restrict__ taps, int tiles_x, int ntiles,
                                                     int blocks_per_frame, int nxcd)
{
    using G = PkGeom<R>;
    constexpr int TW = G::TW, TH = G::TH, RP = G::RP;
    constexpr int NLOAD = (RP * G::V + 255) / 256;
    __shared__ __attribute__((aligned(16))) float s_in[RP * G::IN_P2 * 2];
    __shared__ __attribute__((aligned(16))) float s_mid[G::ROWS * G::MID_P];

    const int tid = threadIdx.x;
    const int frame = blockIdx.x / blocks_per_frame;              // blocks_per_frame % nxcd == 0: blockIdx % nxcd is the XCD
    const int blk = blockIdx.x - frame * blocks_per_frame;
    const int xcd = blk % nxcd, slot = blk / nxcd;
    const int band = (ntiles + nxcd - 1) / nxcd;
    const int tile = xcd * band + slot;
    if (slot >= band || tile >= ntiles) return;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    float *__restrict__ result = batch.result[frame];           // uniform index into the kernarg segment: scalar loads
    const float *__restrict__ image = batch.image[frame];
    float *__restrict__ dog = batch.dog[frame];
    float2 *__restrict__ grad = reinterpret_cast<float2 *>(batch.grad[frame]);
    float *__restrict__ down = batch.down[frame];               // WRITE_BUF: the API launcher passes `buffer` in this slot

    unsigned long long w[2 * R + 1];             // taps as scalar operands of v_pk_fma_f32 (low dword = the float)
#pragma unroll
    for (int i = 0; i <= 2 * R; ++i) w[i] = (unsigned long long)__float_as_uint(taps[i]);

#ifdef NM_CONV_STAMPS
    unsigned long long st_[6];
#define NM_CSTAMP(K) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_[K]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NM_CSTAMP(K) do { } while (0)
#endif
    NM_CSTAMP(0);
    // phase 1: global -> LDS, two rows per thread, interleaved
    {
        float4 a[NLOAD], b[NLOAD];
        pk_load_pairs<R, NLOAD>(a, b, image, RP, y0 - R, x0, width, height, tid);
        pk_store_pairs<R, NLOAD>(a, b, s_in, RP, 0, tid);
    }
    NM_CSTAMP(1);
    __syncthreads();
    NM_CSTAMP(2);
    // phase 2: rows
    {
        const int p = tid & 31, cg = tid >> 5;
        if (p < RP) pk_row_task<R, WRITE_BUF>(s_in, s_mid, w, p, cg, down, x0, y0, width, height);
    }
    NM_CSTAMP(3);
    __syncthreads();
    NM_CSTAMP(4);
    // phase 3: columns + epilogue
    pk_cols_epilogue<R, WRITE_DOG, WRITE_GRAD, WRITE_BUF>(s_in, s_mid, w, tid, x0, y0, width, height, result, dog, grad, down);
#ifdef NM_CONV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the epilogue's stores acknowledged
    NM_CSTAMP(5);
    // one in 61 workgroups of frame 0 records (R, WRITE_GRAD) and its five segment lengths, wave by wave
    if (frame == 0 && (blk % 61) == 0 && (tid & 63) == 0) {
        const unsigned slot = atomicAdd(&nm_conv_stamp_n, 1u);
        if (slot < 4096) {
            unsigned long long *d = nm_conv_stamps + slot * 8;
            d[0] = (unsigned long long)(R * 2 + (WRITE_GRAD ? 1 : 0)) | ((unsigned long long)width << 32);
            for (int k = 0; k < 5; ++k) d[1 + k] = st_[k + 1] - st_[k];
        }
    }
#endif
}

// (Round 6, measured and removed -- the version is in the repository's history, commit "Experiments (measured, negative)":
// conv_pk_ring_kernel, ONE persistent workgroup per LDS slot walking the same XCD-banded tile list with a 2-deep LDS-DMA input
// ring: the next tile's rows requested by buffer_load_dword ... lds (inline asm, so that hipcc places no vmcnt(0) in front of
// the ds_reads) BEFORE the current tile's epilogue stores, `s_waitcnt vmcnt(#stores)` at the top of the next iteration so the
// rows are waited for but not the stores behind them, raw s_barrier + lgkmcnt(0). Bit-exact (44 GPU tests), and 25-40 % SLOWER:
// 65.2 / 66.0 / 73.6 us per frame at 8 / 3 / 2 workgroups per CU against 52.3 for the launches above (64-frame chain, same box,
// alternating). The row-pair interleaved tile forces 4-BYTE DMA -- lane l lands row (l & 1), column l >> 1: 24 wave-instructions
// per wave and tile where the register path issues 6 16-byte loads per thread -- and two input images leave 2-4 workgroups per CU
// where 4-6 one-tile workgroups already overlap one another's loads (the guide's regime rule: LDS-DMA spans pay at ~1 block per CU,
// not at high occupancy). profiles/r06_e_conv_ring_experiment.txt. Also measured, no effect (+-1 %, 16- and 64-frame chains):
// the gradient planes stored non-temporally, so that they would not displace the level the next launch reads. And the frame
// driver's chain in GROUPS of frames (the five launches of octave 0 / 1 for 8, 16 or 32 frames back to back, so that a level is
// still in the 256 MB Infinity Cache when the next launch reads it): 56.1 / 53.9 / 52.6 against 51.9 us per frame.)
// (Round 6, measured and removed: the frames of every other launch walked in REVERSE order, so that a launch starts on the planes
// the previous one wrote last (256 MB of Infinity Cache = the last ~10 frames' outputs): 51.1-51.5 against 51.4-51.7 us per frame,
// three alternations -- nothing. profiles/r06_z_conv_pingpong.txt)
// (Round 5, measured and removed: issue priorities (s_setprio 3 while the tile's loads are issued / 2 or 1 in the epilogue / both):
// 54.2-54.9 us per frame against 54.4-54.5, 64-frame chain, same box -- nothing beyond the run-to-run spread.)
// (Round 5, measured and removed: 64-ROW tiles with 512 threads -- the share of halo rows the row pass filters and the loads
// fetch drops from 1.81 to 1.41 at R = 13 and from 1.44 to 1.22 at R = 7, the row tasks mapped densely (row pair fastest) so
// that whole waves skip the pass; bit-identical (60 GPU tests). 64-frame chain: 57.83 -> 57.91 us per frame with the DoG
// planes, 51.9 -> 53.3 without, 42.1 -> 42.6 without the gradients. Per launch (octave 0, 64 frames): R = 10 with gradient
// 589 -> 546 us (3 workgroups of 8 waves fit a CU where 5 of 4 did), R = 13 438 -> 467, R = 5 222 -> 240, the others +-1 %:
// the launches follow the waves in flight and the workgroup granularity, not the FMA or load count -- like round 4's 48-row
// tiles. profiles/r05_x_kpyr_tall.txt, r05_y_kpyr_tall{0,1}.txt)
// (Round 3, measured and removed: a STREAMING form of this kernel -- a workgroup owns a 64-column strip segment and marches
// down it in 32-row bands, fetching and row-filtering only the 32 new input rows per band, moving the last 2R rows of both LDS
// tiles to their head, with the next band's rows prefetched into registers -- is bit-identical and does 38 % less staging
// and row-pass work, but ran the 16-frame chain in 78-83 us per frame instead of 66-68 (segments of 4 / 6 / 9 / 17 bands:
// 84.5 / 85 / 90 / 97). With its stores off 51, with the prefetch loads off 63, with both off 38 us: inside one workgroup the
// phases of a band are strictly sequential, and on gfx950 stores count in vmcnt, so waiting for the prefetched rows also
// waits for the previous band's stores. Many short workgroups overlap each other's memory phases better than few long
// ones overlap their own. profiles/r03_c_conv_strip_experiment.txt)

// Exhaustive self-test of sqrt_rn's fast path against the IEEE expansion (tests/test_gpu_stages.py).
__global__ __launch_bounds__(256) void selftest_sqrt_kernel(unsigned long long *mismatches)
{
    const unsigned lo = 0x0F800000u /* 2^-96 */, hi = 0x6F800000u /* 2^96 */;
    unsigned long long bad = 0;
    for (unsigned long long u = lo + (unsigned long long)blockIdx.x * 256 + threadIdx.x; u < hi; u += (unsigned long long)gridDim.x * 256) {
        const float s = __uint_as_float((unsigned)u);
        if (__float_as_uint(sqrt_rn_fast_core(s)) != __float_as_uint(__builtin_sqrtf(s))) ++bad;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && __float_as_uint(sqrt_rn_fast_core(0.0f)) != 0u) ++bad;
    // div_pair against `/`: 2^31 pseudo-random (num, den) with den in [2^-49, 2^49), |num| >= 2^-100, |num/den| < 4
    unsigned long long st = 0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * 256 + threadIdx.x + 1);
    for (int it = 0; it < 1024; ++it) {
        float nd[4];
        for (int k = 0; k < 2; ++k) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned a = (unsigned)(st >> 32), b = (unsigned)st;
            const int ed = (int)((a >> 23) % 98u) - 49, eq = -(int)((b >> 23) % 102u) + 1;
            const float den = __uint_as_float(((unsigned)(ed + 127) << 23) | (a & 0x7FFFFFu));
            int en = ed + eq;
            en = en < -100 ? -100 : en;
            const float num = __uint_as_float(((unsigned)(en + 127) << 23) | (b & 0x7FFFFFu) | ((a >> 31) << 31));
            nd[2 * k] = num; nd[2 * k + 1] = den;
        }
        const v2f q = div_pair((v2f){nd[0], nd[2]}, (v2f){nd[1], nd[3]});
        const float q0 = nd[0] / nd[1], q1 = nd[2] / nd[3];
        if (__float_as_uint(q.x) != __float_as_uint(q0)) ++bad;
        if (__float_as_uint(q.y) != __float_as_uint(q1)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
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
    const int nxcd = nm_xcd_count();
    const int blocks = nm_divup(ntiles, nxcd) * nxcd;
    dim3 grid(blocks);
    float2 *g2 = reinterpret_cast<float2 *>(grad);
#define NM_CONV_LAUNCH(BUF, DOG, GRAD)                                                                              \
    hipLaunchKernelGGL((conv_sep_kernel<R, TH, BUF, DOG, GRAD, VEC>), grid, dim3(TH * 8), 0, stream, result, image, \
                       buffer, dog, g2, width, height, taps, tiles_x, ntiles, nxcd)
    if (buffer) {
        if (dog || grad || TH != 32) return (int)hipErrorInvalidValue;   // the API path never asks for the fused outputs
        if (TH == 32) NM_CONV_LAUNCH(true, false, false);
    } else if (dog && grad) {
        NM_CONV_LAUNCH(false, true, true);
    } else if (dog) {
        NM_CONV_LAUNCH(false, true, false);
    } else if (grad) {
        NM_CONV_LAUNCH(false, false, true);
    } else {
        NM_CONV_LAUNCH(false, false, false);
    }
#undef NM_CONV_LAUNCH
    NM_LAUNCH_CHECK();
    return 0;
}

template <int R>
static int launch_conv_pk(const NmConvBatch &b, int width, int height, const float *taps, hipStream_t stream)
{
    const int tiles_x = nm_divup(width, 64), tiles_y = nm_divup(height, 32);
    const int ntiles = tiles_x * tiles_y;
    const int nxcd = nm_xcd_count();
    const bool dog = b.dog[0] != nullptr, grad = b.grad[0] != nullptr;
    const int bpf = nm_divup(ntiles, nxcd) * nxcd;
    dim3 grid(bpf * b.n);
#define NM_PK_LAUNCH(DOG, GRAD)                                                                                     \
    hipLaunchKernelGGL((conv_pk_kernel<R, DOG, GRAD>), grid, dim3(256), 0, stream, b, width, height, taps, tiles_x, \
                       ntiles, bpf, nxcd)
    if (dog && grad) NM_PK_LAUNCH(true, true);
    else if (dog) NM_PK_LAUNCH(true, false);
    else if (grad) NM_PK_LAUNCH(false, true);       // frame driver: the DoG planes are not materialised (detection subtracts the levels)
    else NM_PK_LAUNCH(false, false);
#undef NM_PK_LAUNCH
    NM_LAUNCH_CHECK();
    return 0;
}

template <int R>
static int launch_conv_pk_buf(float *result, const float *image, float *buffer, int width, int height, const float *taps,
                              hipStream_t stream)
{
    NmConvBatch b{};
    b.result[0] = result; b.image[0] = image; b.down[0] = buffer; b.n = 1;
    const int tiles_x = nm_divup(width, 64), tiles_y = nm_divup(height, 32);
    const int ntiles = tiles_x * tiles_y;
    const int nxcd = nm_xcd_count();
    const int bpf = nm_divup(ntiles, nxcd) * nxcd;
    hipLaunchKernelGGL((conv_pk_kernel<R, false, false, true>), dim3(bpf), dim3(256), 0, stream, b, width, height, taps,
                       tiles_x, ntiles, bpf, nxcd);
    NM_LAUNCH_CHECK();
    return 0;
}

template <int R, bool VEC>
static int launch_conv_rv(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                          int height, const float *taps, hipStream_t stream)
{
    if (VEC && buffer && !dog && !grad && result && (size_t)width * height * 4 < (1ull << 32) &&
        ((reinterpret_cast<uintptr_t>(buffer) | reinterpret_cast<uintptr_t>(result)) & 15) == 0)
        return launch_conv_pk_buf<R>(result, image, buffer, width, height, taps, stream);
    if (VEC && !buffer && (size_t)width * height * 4 < (1ull << 32)) {
        NmConvBatch b{};
        b.result[0] = result; b.image[0] = image; b.dog[0] = dog; b.grad[0] = grad; b.n = 1;
        return launch_conv_pk<R>(b, width, height, taps, stream);
    }
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

// Batched form used by the frame driver: one launch for all frames when the packed kernel applies (same geometry, all
// planes 16-byte aligned, same set of outputs), otherwise frame by frame.
int nm_launch_convolve_batch(const NmConvBatch &b, int width, int height, const float *taps, int radius, hipStream_t stream)
{
    if (b.n <= 0 || width <= 0 || height <= 0) return 0;
    if (b.n > NM_MAX_BATCH) return (int)hipErrorInvalidValue;
    bool pk = (width % 4 == 0) && (size_t)width * height * 4 < (1ull << 32);
    for (int f = 0; f < b.n; ++f) {
        pk = pk && ((reinterpret_cast<uintptr_t>(b.image[f]) & 15) == 0);
        pk = pk && ((b.result[f] != nullptr) == (b.result[0] != nullptr)) && ((b.dog[f] != nullptr) == (b.dog[0] != nullptr)) &&
             ((b.grad[f] != nullptr) == (b.grad[0] != nullptr)) && ((b.down[f] != nullptr) == (b.down[0] != nullptr));
    }
    if (pk) {
        switch (radius) {
            case 5: return launch_conv_pk<5>(b, width, height, taps, stream);
            case 7: return launch_conv_pk<7>(b, width, height, taps, stream);
            case 8: return launch_conv_pk<8>(b, width, height, taps, stream);
            case 10: return launch_conv_pk<10>(b, width, height, taps, stream);
            case 12: return launch_conv_pk<12>(b, width, height, taps, stream);
            case 13: return launch_conv_pk<13>(b, width, height, taps, stream);
            case 16: return launch_conv_pk<16>(b, width, height, taps, stream);
            default: break;
        }
    }
    for (int f = 0; f < b.n; ++f) {
        const int rc = nm_launch_convolve(b.result[f], b.image[f], nullptr, b.dog[f], b.grad[f], width, height, taps, radius, stream);
        if (rc) return rc;
    }
    if (b.down[0]) {                  // the packed kernel decimates in its epilogue; here it is a launch of its own
        if (!b.result[0]) return (int)hipErrorInvalidValue;
        NmPlaneBatch d{};
        d.n = b.n;
        for (int f = 0; f < b.n; ++f) { d.dst[f] = b.down[f]; d.src[f] = b.result[f]; }
        return nm_launch_downsample2_batch(d, width >> 1, height >> 1, width, stream);
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

__global__ __launch_bounds__(256) void downsample2_batch_kernel(NmPlaneBatch b, int rw, int rh, int sw)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= rw || y >= rh) return;
    float *__restrict__ dst = b.dst[blockIdx.z];
    const float *__restrict__ src = b.src[blockIdx.z];
    dst[(size_t)y * rw + x] = src[(size_t)(y * 2) * sw + (x * 2)];
}

int nm_launch_downsample2_batch(const NmPlaneBatch &b, int rw, int rh, int sw, hipStream_t stream)
{
    if (b.n <= 0 || rw <= 0 || rh <= 0) return 0;
    if (b.n > NM_MAX_BATCH) return (int)hipErrorInvalidValue;
    dim3 grid(nm_divup(rw, 64), nm_divup(rh, 4), b.n);
    hipLaunchKernelGGL(downsample2_batch_kernel, grid, dim3(256), 0, stream, b, rw, rh, sw);
    NM_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void subtract_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                      float *__restrict__ C, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) C[i] = A[i] - B[i];
}

struct NmSubBatch { const float *A[8]; const float *B[8]; float *C[8]; };
__global__ __launch_bounds__(256) void subtract_batch_kernel(NmSubBatch b, size_t n)
{
    const float *__restrict__ A = b.A[blockIdx.y], *__restrict__ B = b.B[blockIdx.y];
    float *__restrict__ C = b.C[blockIdx.y];
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

int nm_selftest_sqrt(unsigned long long *d_mismatches, void *stream)
{
    if (!d_mismatches) return (int)hipErrorInvalidValue;
    NM_RETURN_IF(hipMemsetAsync(d_mismatches, 0, sizeof(unsigned long long), nm_stream(stream)));
    hipLaunchKernelGGL(selftest_sqrt_kernel, dim3(4096), dim3(256), 0, nm_stream(stream), d_mismatches);
    NM_LAUNCH_CHECK();
    return 0;
}

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

// compute_dog's loop (sift/siftfunctions.cu:42-51) as one launch: C[k] = A[k] - B[k] for n <= 8 planes.
int nm_subtract_batch_f32(int n, const float *const *A, const float *const *B, float *const *C, int width, int height,
                          void *stream)
{
    const size_t npx = (size_t)width * height;
    if (n <= 0 || npx == 0) return 0;
    if (n > 8 || !A || !B || !C) return (int)hipErrorInvalidValue;
    NmSubBatch b{};
    for (int k = 0; k < n; ++k) { b.A[k] = A[k]; b.B[k] = B[k]; b.C[k] = C[k]; }
    int blocks = (int)((npx + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(subtract_batch_kernel, dim3(blocks, n), dim3(256), 0, nm_stream(stream), b, npx);
    NM_LAUNCH_CHECK();
    return 0;
}

// compute_gradients' loop (sift/siftfunctions.cu:53-63) as one launch: n <= 3 planes.
int nm_gradient_batch_f32(int n, const float *const *source, float *const *result, int width, int height, void *stream)
{
    if (n <= 0) return 0;
    if (n > 3 || !source || !result) return (int)hipErrorInvalidValue;
    NmGradBatch b{};
    b.n = n;
    for (int k = 0; k < n; ++k) { b.src[k] = source[k]; b.dst[k] = result[k]; }
    return nm_launch_gradient_batch(b, width, height, nm_stream(stream));
}

int nm_gradient_f32(const float *source, float *result, int width, int height, void *stream)
{
    NmGradBatch b{};
    b.src[0] = source; b.dst[0] = result; b.n = 1;
    return nm_launch_gradient_batch(b, width, height, nm_stream(stream));
}

}  // extern "C"

#ifdef NM_CONV_STAMPS
extern "C" __attribute__((visibility("default"))) int nm_debug_conv_stamps(unsigned long long *host_dst, unsigned *n, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(nm_conv_stamp_n), sizeof(unsigned));
    if (!rc) rc = (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nm_conv_stamps), sizeof(nm_conv_stamps));
    if (!rc && reset) { const unsigned z = 0; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(nm_conv_stamp_n), &z, sizeof(z)); }
    return rc;
}
#endif
