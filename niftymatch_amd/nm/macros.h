// macros.h -- anchor header of the install layout: NiftyMatchConfig.cmake locates the include directory by searching
// for this file name (reference: src/cmake/NiftyMatchConfig.cmake:16-19), so it has to exist in <prefix>/include/nm.
// It carries the one utility macro the reference ships here, plus the identification macros of the MI355X build.
#ifndef __MACROS_H__
#define __MACROS_H__

// ---- identification of this implementation (absent from the reference) ----
#define NM_AMD_BACKEND 1              // hand-written HIP kernels for gfx950, no CUDA path
#define NM_AMD_TARGET_ARCH "gfx950"
#define NM_AMD_VERSION_MAJOR 0
#define NM_AMD_VERSION_MINOR 1
#define NM_AMD_WAVEFRONT 64

// ---- utility macro of the reference's macros.h ----
// Deletes the copy constructor and copy assignment of TypeName; put it in the private section of a class that owns
// device memory or HIP handles (CudaTimer's events, for instance).
#define DISALLOW_COPY_AND_ASSIGNMENT(TypeName) \
    TypeName(const TypeName &) = delete;       \
    void operator=(const TypeName &) = delete

#endif
