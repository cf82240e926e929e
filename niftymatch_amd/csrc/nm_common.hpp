// nm_common.hpp -- shared host/device helpers for libnm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define NM_WAVE 64

#define NM_RETURN_IF(expr)                                  \
    do {                                                    \
        hipError_t nm_e_ = (expr);                          \
        if (nm_e_ != hipSuccess) return (int)nm_e_;         \
    } while (0)

#define NM_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t nm_e_ = hipGetLastError();               \
        if (nm_e_ != hipSuccess) return (int)nm_e_;         \
    } while (0)

static inline int nm_divup(int a, int b) { return (a + b - 1) / b; }

// Compute units / XCDs of the CURRENT device, read once per device from hipDeviceProp_t (a partitioned mode such as CPX
// shows 32 CUs and one XCD per device). Without a device (host-only planning calls on a CPU box) the MI355X SPX values
// 256 / 8 are assumed. XCDs are not a device property: 32 CUs per XCD on gfx950.
int nm_cu_count();
int nm_xcd_count();
static inline hipStream_t nm_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- profiling hook (nm_profile_events / nm_profile_event_pairs) ----
// Either one (start, stop) pair, re-recorded by every launch of the site, or a caller-owned list of pairs consumed in
// launch order (a batched call launches the site several times).
struct NmProfSite { hipEvent_t start, stop; void *const *list; int n, next; };
// Detection units (nm_detect_dev.hpp): a unit is one image row of a segment of NM_DET_SEG_W pixels (four waves of NM_DET_WAVE_W tested
// columns + two halo lanes each); units per row = nm_divup(width, NM_DET_SEG_W); a unit's staging slots stay 256.
constexpr int NM_DET_WAVE_W = 62, NM_DET_SEG_W = 4 * NM_DET_WAVE_W;
extern thread_local NmProfSite nm_prof_sites[6];   // NM_PROF_SITES (include/nm_abi.h)
static inline void nm_prof_begin(int site, hipStream_t st)
{
    NmProfSite &p = nm_prof_sites[site];
    if (p.list) { if (p.next < p.n) (void)hipEventRecord(static_cast<hipEvent_t>(p.list[2 * p.next]), st); }
    else if (p.start) (void)hipEventRecord(p.start, st);
}
static inline void nm_prof_end(int site, hipStream_t st)
{
    NmProfSite &p = nm_prof_sites[site];
    if (p.list) { if (p.next < p.n) { (void)hipEventRecord(static_cast<hipEvent_t>(p.list[2 * p.next + 1]), st); ++p.next; } }
    else if (p.stop) (void)hipEventRecord(p.stop, st);
}

// ---- internal launchers shared between translation units (not part of the C ABI) ----
struct NmGradBatch {            // up to 3 planes per launch (gradient levels 0..2 of one octave)
    const float *src[3];
    float *dst[3];
    int n;
};
int nm_launch_gradient_batch(const NmGradBatch &b, int width, int height, hipStream_t stream);
// Gaussian level + fused DoG (dog = result - image) + fused gradient of `image` (float2 plane); buffer, dog and grad
// may be NULL. buffer (the materialised row pass of the API path) excludes dog/grad.
int nm_launch_convolve(float *result, const float *image, float *buffer, float *dog, float *grad, int width,
                       int height, const float *taps_dev, int radius, hipStream_t stream);

// The same Gaussian launch over up to NM_MAX_BATCH equally sized frames (one grid; the frame index is the slow part of
// blockIdx.x): 1080p octave 0 is only ~4 workgroups per CU, a frame pair or quad fills the chip and amortises the
// launch / drain phases that bound a single frame.
#define NM_MAX_BATCH 64
struct NmConvBatch {
    float *result[NM_MAX_BATCH];
    const float *image[NM_MAX_BATCH];
    float *dog[NM_MAX_BATCH];
    float *grad[NM_MAX_BATCH];
    float *down[NM_MAX_BATCH];      // optional: `result` decimated by 2 ((width/2) x (height/2), kernels/downsample.cu:6-17)
    int n;
};
int nm_launch_convolve_batch(const NmConvBatch &b, int width, int height, const float *taps_dev, int radius,
                             hipStream_t stream);
struct NmPlaneBatch {
    float *dst[NM_MAX_BATCH];
    const float *src[NM_MAX_BATCH];
    int n;
};
int nm_launch_downsample2_batch(const NmPlaneBatch &b, int rw, int rh, int sw, hipStream_t stream);

// Device-side record the frame driver shares between its kernels.
struct NmFrameBook {
    int num_items;        // descriptors written so far (<= capacity)
    int oct_base[21];     // first output index of octave o; oct_base[num_octaves] = num_items at the end
    int lvl_count[20][3]; // accepted keypoints per (octave, level) after the empty-level rule
    int lvl_base[20][3];  // output index of the first keypoint of (octave, level)
    int lvl_n[20][3];     // keypoints of (octave, level) that fit under capacity
};
