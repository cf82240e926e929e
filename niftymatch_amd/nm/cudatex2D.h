// cudatex2D.h -- drop-in for NiftyMatch src/gpu/utils/cudatex2D.h:12-52.
// The reference wraps a cudaArray in a cudaTextureObject_t (border addressing, linear filter, unnormalised coordinates).
// Here a texture is a VIEW of a linear device plane; the library filters in software with the same addressing rules, so
// no array copy is needed. NmTexture takes the place of cudaTextureObject_t in resample.h.
#ifndef __CUDA_TEX2D_H__
#define __CUDA_TEX2D_H__

#include <hip/hip_vector_types.h>
#include "macros.h"

enum NmTexelFormat { NM_TEXEL_U8_NORM = 0, NM_TEXEL_U8X4_NORM = 1, NM_TEXEL_F32 = 2 };

struct NmTexture {
    const void *data = nullptr;     // device pointer, row-major, pitch = width texels
    int width = 0, height = 0;
    int format = NM_TEXEL_F32;
    explicit operator bool() const { return data != nullptr; }
};

class CudaTex2D {
public:
    CudaTex2D() {}
    CudaTex2D(const float *plane, int width, int height) { set(plane, width, height); }
    CudaTex2D(const unsigned char *plane, int width, int height) { set(plane, width, height); }
    CudaTex2D(const uchar4 *plane, int width, int height) { set(plane, width, height); }
    ~CudaTex2D() {}

    void set(const float *plane, int width, int height) { bind(plane, width, height, NM_TEXEL_F32); }
    void set(const unsigned char *plane, int width, int height) { bind(plane, width, height, NM_TEXEL_U8_NORM); }
    void set(const uchar4 *plane, int width, int height) { bind(plane, width, height, NM_TEXEL_U8X4_NORM); }
    void release() { _tex = NmTexture(); }

    operator NmTexture() const { return _tex; }

private:
    void bind(const void *p, int w, int h, int f) { _tex.data = p; _tex.width = w; _tex.height = h; _tex.format = f; }
    NmTexture _tex;
    DISALLOW_COPY_AND_ASSIGNMENT(CudaTex2D);
};

#endif
