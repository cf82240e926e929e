// nm_frame.hip -- library-level entry points: device helpers, host-side tap generation, and the per-frame driver
// that runs the reference's implied client loop (orchestration order of sift/siftfunctions.cu:42-181)
// as one allocation-free, sync-free launch sequence on a stream.
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "../../include/nm_abi.h"
#include "../nm/siftparams.h"
#include "nm_common.hpp"
#include "nm_describe.hpp"
#include "nm_keypoint.hpp"
#include "nm_tail.hpp"

namespace {

__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned int *__restrict__ p, size_t n, unsigned int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

}  // namespace

thread_local NmProfSite nm_prof_sites[NM_PROF_SITES] = {};

struct nm_sift_arena {
    int width, height, capacity;
    int device;                // the HIP device every buffer, the side stream and the events belong to
    SiftParams params;
    size_t npix;
    size_t bytes;
    std::vector<void *> allocs;
    float *taps_base; int base_radius;
    float *taps[8]; int radii[8];
    float *level[6];           // Gaussian levels of octave 0 (and of every octave in the single-octave API call)
    float *lev[20][6];         // Gaussian levels PER OCTAVE (lev[0] = level): the frame driver's detection reads them (DoG =
                               // difference of consecutive levels, formed in the detection kernel) while the next octave's
                               // pyramid is being computed, so the octaves cannot share planes
    float *dog[20][5];         // DoG planes PER OCTAVE: detection of octave o overlaps the pyramid of octave o+1
    hipStream_t side;          // detection / compaction stream forked off the caller's stream
    hipStream_t desc;          // orientation + descriptors of the large octaves, beside the small octaves' pyramids / detection
    hipEvent_t ev_pyr[20], ev_join, ev_det, ev_desc;
    float *grad[20];           // per octave: 3 float2 planes
    size_t grad_off[20];       // grad[o] = grad[0] + grad_off[o]: the gradient planes of all octaves are one block
    size_t plane_stride[20];   // floats between consecutive levels / DoG planes of an octave (one block per octave)
    float *staging; size_t stage_stride;
    int *counts, *offsets; int max_blocks;
    NmFrameBook *book;
    float *kpts, *orients;     // internal lists used when the caller passes NULL
    const float *mask;         // nm_sift_arena_set_mask: caller-owned full-resolution plane (width x height) or NULL
    // octave tail (nm_tail.hip): the octaves >= tail.T of a call run as ONE persistent launch. Their detection stages into
    // per-octave lists (an octave's gather may run after the next octave's detection), the launch finds a frame's planes in a
    // device-resident table, and the FIRST arena of a call lends its state words (zero between launches).
    float *stg[20]; size_t stg_stride[20]; int *cnt[20];
    NmTailFrame tail_frame;      // this arena's planes of the tail octaves (copied into the launch's arguments)
    int *tail_state;
    NmTailArgs tail;           // the plan for this geometry (per-call fields are filled by the driver)
    bool tail_ok;

    template <typename T>
    int alloc(T **p, size_t n)
    {
        void *q = nullptr;
        const size_t b = n * sizeof(T);
        hipError_t e = hipMalloc(&q, b ? b : 4);
        if (e != hipSuccess) return (int)e;
        allocs.push_back(q);
        bytes += b;
        *p = static_cast<T *>(q);
        return 0;
    }
};

// write_dog = false (frame driver): the DoG planes are not materialised -- detection forms them from the levels -- which
// takes 20 of the chain's 64 written bytes per pixel away (level 5 has to be stored instead: + 4). per_octave: the
// levels live in the octave's own planes (lev[o]); otherwise in level[] (single-octave API call).
// NM_FRAME_DOG=1: the frame driver materialises the DoG planes as in round 1 (detection then reads them).
static bool frame_driver_writes_dog()
{
    static const bool v = [] { const char *e = getenv("NM_FRAME_DOG"); return e && e[0] == '1'; }();
    return v;
}

extern "C" {

const char *nm_version(void) { return "niftymatch_amd 0.1.0 gfx950"; }
int nm_device_count(int *count) { return (int)hipGetDeviceCount(count); }
int nm_set_device(int device) { return (int)hipSetDevice(device); }
const char *nm_error_string(int status) { return hipGetErrorString((hipError_t)status); }

int nm_profile_events(int site, void *start_event, void *stop_event)
{
    if (site < 0 || site >= NM_PROF_SITES) return (int)hipErrorInvalidValue;
    nm_prof_sites[site].start = static_cast<hipEvent_t>(start_event);
    nm_prof_sites[site].stop = static_cast<hipEvent_t>(stop_event);
    nm_prof_sites[site].list = nullptr; nm_prof_sites[site].n = nm_prof_sites[site].next = 0;
    return 0;
}

int nm_profile_event_pairs(int site, void *const *events, int npairs)
{
    if (site < 0 || site >= NM_PROF_SITES || npairs < 0) return (int)hipErrorInvalidValue;
    nm_prof_sites[site].start = nm_prof_sites[site].stop = nullptr;
    nm_prof_sites[site].list = npairs ? events : nullptr;
    nm_prof_sites[site].n = npairs; nm_prof_sites[site].next = 0;
    return 0;
}

int nm_fill_u32(void *dst, size_t count, unsigned int pattern, void *stream)
{
    if (count == 0) return 0;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, nm_stream(stream),
                       static_cast<unsigned int *>(dst), count, pattern);
    NM_LAUNCH_CHECK();
    return 0;
}

// sift/pyramidata.cu:105-123: radius = ceil(4 sigma); w_j = (float)exp(-0.5 ((j-r)/sigma)^2) with a double exp on the
// host; normalised by the float running sum.
int nm_create_kernel_for_sigma(float sigma, float *taps)
{
    const int radius = (int)(std::ceil(sigma * 4));
    const int length = 2 * radius + 1;
    float sum = 0.f;
    for (int j = 0; j < length; ++j) {
        float u = ((float)j - radius) / sigma;
        u = (float)std::exp(-0.5 * (u * u));
        if (taps) taps[j] = u;
        sum += u;
    }
    if (taps)
        for (int j = 0; j < length; ++j) taps[j] = taps[j] / sum;
    return radius;
}

int nm_sift_arena_create(int width, int height, int capacity, nm_sift_arena **out)
{
    if (!out || width <= 0 || height <= 0 || capacity <= 0) return (int)hipErrorInvalidValue;
    nm_sift_arena *a = new (std::nothrow) nm_sift_arena();
    if (!a) return (int)hipErrorOutOfMemory;
    a->width = width; a->height = height; a->capacity = capacity;
    a->side = nullptr; a->ev_join = nullptr;
    a->desc = nullptr; a->ev_det = nullptr; a->ev_desc = nullptr;
    a->mask = nullptr;
    a->device = -1;
    (void)hipGetDevice(&a->device);
    for (int o = 0; o < 20; ++o) a->ev_pyr[o] = nullptr;
    a->params = SiftParams(width, height);
    a->npix = (size_t)width * height;
    a->bytes = 0;
    const SiftParams &P = a->params;
    if (P._num_octaves > 20 || (int)P._sigmas.size() > 8) { delete a; return (int)hipErrorInvalidValue; }
    int rc = 0;
    auto upload = [&](float sigma, float **dev, int *radius) -> int {
        *radius = nm_create_kernel_for_sigma(sigma, nullptr);
        std::vector<float> h(2 * *radius + 1);
        nm_create_kernel_for_sigma(sigma, h.data());
        int e = a->alloc(dev, h.size());
        if (e) return e;
        return (int)hipMemcpy(*dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    };
    rc = upload(P._base_smooth, &a->taps_base, &a->base_radius);
    for (size_t i = 0; !rc && i < P._sigmas.size(); ++i) rc = upload(P._sigmas[i], &a->taps[i], &a->radii[i]);
    // an octave's six levels (and its five DoG planes) are ONE block, plane p at p * plane_stride[o]: the detection launches then
    // take one pointer per frame (NmDetectArgs). The stride is the plane rounded up to 4 floats: every plane 16-byte aligned.
    for (int o = 0; !rc && o < P._num_octaves; ++o) {
        const size_t plane = (size_t)(width >> o) * (height >> o);
        a->plane_stride[o] = (plane + 3) & ~(size_t)3;
        float *blk = nullptr;
        rc = a->alloc(&blk, 6 * a->plane_stride[o]);
        for (int i = 0; i < 6; ++i) a->lev[o][i] = blk + i * a->plane_stride[o];
        if (!rc) rc = a->alloc(&blk, 5 * a->plane_stride[o]);
        for (int i = 0; i < 5; ++i) a->dog[o][i] = blk + i * a->plane_stride[o];
    }
    for (int i = 0; i < 6; ++i) a->level[i] = a->lev[0][i];
    {   // the gradient planes of all octaves: one block (NmDescribeArgs takes one pointer per frame and the offsets)
        size_t total = 0;
        for (int o = 0; o < 20; ++o) a->grad_off[o] = 0;
        for (int o = 0; o < P._num_octaves; ++o) {
            a->grad_off[o] = total;
            total += (6 * (size_t)(width >> o) * (height >> o) + 3) & ~(size_t)3;
        }
        float *blk = nullptr;
        if (!rc) rc = a->alloc(&blk, total);
        for (int o = 0; o < P._num_octaves; ++o) a->grad[o] = blk + a->grad_off[o];
    }
    for (int o = 0; !rc && o < P._num_octaves; ++o) rc = (int)hipEventCreateWithFlags(&a->ev_pyr[o], hipEventDisableTiming);
    if (!rc) rc = (int)hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming);
    if (!rc) rc = (int)hipEventCreateWithFlags(&a->ev_det, hipEventDisableTiming);
    if (!rc) rc = (int)hipEventCreateWithFlags(&a->ev_desc, hipEventDisableTiming);
    // (Round 5, measured: HIP stream priorities for these two streams -- the device offers 0 and -1 -- change nothing when one of
    // them is raised (headline 2 934-3 058 against 2 953-3 047) and cost 17 % when both are (2 479-2 498).)
    if (!rc) rc = (int)hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking);
    if (!rc) rc = (int)hipStreamCreateWithFlags(&a->desc, hipStreamNonBlocking);
    a->max_blocks = height * nm_divup(width, NM_DET_SEG_W);
    a->stage_stride = (size_t)a->max_blocks * 256;
    if (!rc) rc = a->alloc(&a->staging, 3 * a->stage_stride * 4);
    if (!rc) rc = a->alloc(&a->counts, (size_t)3 * a->max_blocks);
    if (!rc) rc = a->alloc(&a->offsets, (size_t)3 * a->max_blocks);
    if (!rc) rc = a->alloc(&a->book, 1);
    if (!rc) rc = a->alloc(&a->kpts, (size_t)4 * capacity);
    if (!rc) rc = a->alloc(&a->orients, (size_t)2 * capacity);
    if (!rc) rc = (int)hipMemset(a->book, 0, sizeof(NmFrameBook));
    // octave tail: first octave T = 2 (NM_FRAME_TAIL=0 switches it off, 1..3 choose T): octaves 0 and 1 are real streaming
    // work for the whole chip and keep their per-octave launches
    a->tail_ok = false; a->tail_frame = NmTailFrame{}; a->tail_state = nullptr;
    for (int o = 0; o < 20; ++o) { a->stg[o] = nullptr; a->stg_stride[o] = 0; a->cnt[o] = nullptr; }
    {
        const char *e = getenv("NM_FRAME_TAIL");
        const int T = e ? atoi(e) : 2;
        int radii[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < P._sigmas.size() && i < 5; ++i) radii[i] = a->radii[i];
        if (!rc && T >= 1 && T <= 3 && P._sigmas.size() == 5 && P._num_dog_levels == 3 && !frame_driver_writes_dog() &&
            nm_tail_plan(a->tail, width, height, P._num_octaves, T, radii)) {
            NmTailFrame &h = a->tail_frame;
            for (int o = T; !rc && o < P._num_octaves; ++o) {
                const int j = o - T;
                for (int i = 0; i < 6; ++i) h.lev[j][i] = a->lev[o][i];
                h.grad[j] = a->grad[o];
                const size_t units = (size_t)(height >> o) * nm_divup(width >> o, NM_DET_SEG_W);
                a->stg_stride[o] = units * 256;
                rc = a->alloc(&a->stg[o], 3 * a->stg_stride[o] * 4);
                if (!rc) rc = a->alloc(&a->cnt[o], 3 * units);
                h.staging[j] = a->stg[o]; h.stage_stride[j] = a->stg_stride[o]; h.counts[j] = a->cnt[o];
            }
            h.book = a->book;
            if (!rc) rc = a->alloc(&a->tail_state, NM_TAIL_STATE_INTS);
            if (!rc) rc = (int)hipMemset(a->tail_state, 0, NM_TAIL_STATE_INTS * sizeof(int));
            for (int i = 0; i < 5; ++i) a->tail.taps[i] = a->taps[i];
            a->tail.trace = nullptr;
            const char *tr = getenv("NM_TAIL_TRACE");          // diagnostic: per-item timestamps of the tail launch
            if (!rc && tr && tr[0] == '1') {
                rc = a->alloc(&a->tail.trace, (size_t)16 * NM_TAIL_MAX_FRAMES * a->tail.items_per_frame);
                if (!rc) rc = (int)hipMemset(a->tail.trace, 0, (size_t)128 * NM_TAIL_MAX_FRAMES * a->tail.items_per_frame);
            }
            a->tail_ok = !rc;
        }
    }
    if (!rc) rc = (int)hipDeviceSynchronize();
    if (rc) { nm_sift_arena_destroy(a); return rc; }
    *out = a;
    return 0;
}

void nm_sift_arena_destroy(nm_sift_arena *a)
{
    if (!a) return;
    if (a->side) { (void)hipStreamSynchronize(a->side); (void)hipStreamDestroy(a->side); }
    if (a->desc) { (void)hipStreamSynchronize(a->desc); (void)hipStreamDestroy(a->desc); }
    if (a->ev_det) (void)hipEventDestroy(a->ev_det);
    if (a->ev_desc) (void)hipEventDestroy(a->ev_desc);
    for (int o = 0; o < 20; ++o)
        if (a->ev_pyr[o]) (void)hipEventDestroy(a->ev_pyr[o]);
    if (a->ev_join) (void)hipEventDestroy(a->ev_join);
    for (void *p : a->allocs) (void)hipFree(p);
    delete a;
}

size_t nm_sift_arena_bytes(const nm_sift_arena *a) { return a ? a->bytes : 0; }

// Diagnostic: the per-item record of the arena's last octave-tail launch (NM_TAIL_TRACE=1 when the arena was created; the
// arena must have been the FIRST of its call). Synchronises the device. out: 4 words per item -- kind | slot << 8 | frame
// << 16 | index << 24 | workgroup << 48, then the 100 MHz clock when the ticket was drawn, when its inputs were ready, when it
// was done. Returns the number of items per frame (0: no trace), *n_segments / segments (5 ints each: kind, slot, items per
// frame, first item, octave) describe the plan.
int nm_sift_arena_tail_trace(const nm_sift_arena *a, unsigned long long *out, int max_items, int *segments, int max_segments)
{
    if (!a || !a->tail_ok) return 0;
    for (int i = 0; segments && i < a->tail.n_seg && i < max_segments; ++i) {
        const NmTailSeg &g = a->tail.seg[i];
        int *r = segments + 5 * i;
        r[0] = g.kind; r[1] = g.slot; r[2] = g.per_frame; r[3] = g.first_per_frame; r[4] = a->tail.oct[g.slot].o;
    }
    // the trace buffer holds NM_TAIL_MAX_FRAMES * items_per_frame records of 128 bytes: never copy past it
    max_items = std::min(max_items, NM_TAIL_MAX_FRAMES * a->tail.items_per_frame);
    if (out && a->tail.trace && max_items > 0) {
        // layout of the launch's record: 4 words per item for all n_frames * items_per_frame items, then 12 phase stamps per item
        // (conv items only); max_items must be that product, out holds 16 words per item
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        if (hipMemcpy(out, a->tail.trace, (size_t)max_items * 128, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return a->tail.items_per_frame;
}
int nm_sift_arena_tail_segments(const nm_sift_arena *a) { return (a && a->tail_ok) ? a->tail.n_seg : 0; }

// Status of the last octave-tail launch that used this arena's state words (the FIRST arena of a call of <= 2 frames lends
// them): 0 = complete, 1 = a wait inside the launch hit its spin limit and the octaves >= T of that call's frames were dropped
// (their d_num_items read -1). Synchronises `stream`. An arena without a tail plan reports 0.
int nm_sift_arena_tail_status(const nm_sift_arena *a, int *status, void *stream)
{
    if (!a || !status) return (int)hipErrorInvalidValue;
    *status = 0;
    if (!a->tail_ok) return 0;
    NM_RETURN_IF(hipStreamSynchronize(nm_stream(stream)));
    NM_RETURN_IF(hipMemcpy(status, a->tail_state + 3, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// TEST HOOK: sets the sticky error word of the arena's tail state, as a timed-out wait would, so that the NEXT tail launch on
// it drains without working (tests/test_gpu_tail.py: the call must report failure and the call after it must be correct).
int nm_sift_arena_tail_inject_error(nm_sift_arena *a)
{
    if (!a || !a->tail_ok) return (int)hipErrorInvalidValue;
    const int one = 1;
    NM_RETURN_IF(hipDeviceSynchronize());
    NM_RETURN_IF(hipMemcpy(a->tail_state + 2, &one, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// Kernel launches one nm_sift_detect_describe[_batch] call of n frames on this arena issues (HOST function): base blur, five
// Gaussian launches + detect / scan / gather per octave, orientation + descriptors; with the octave tail (calls of up to
// NM_FRAME_TAIL_MAX_BATCH = 2 frames) the octaves >= T are two launches and the description runs in two parts.
static int tail_max_batch()
{
    static const int v = [] { const char *e = getenv("NM_FRAME_TAIL_MAX_BATCH"); return e ? atoi(e) : 2; }();
    return v;
}
int nm_sift_arena_launches_per_call(const nm_sift_arena *a, int n)
{
    if (!a || n <= 0) return 0;
    const int oct = a->params._num_octaves;
    const bool tail = a->tail_ok && n <= tail_max_batch() && n <= NM_TAIL_MAX_FRAMES && !frame_driver_writes_dog();
    if (!tail) return 1 + oct * 8 + 2;
    return 1 + a->tail.T * 8 + 2 + 4;
}

// HOST function (no device access): the octave-tail plan of a width x height frame with first tail octave T -- what
// nm_sift_arena_create makes for the arena. segments: 8 ints each (kind, slot, items per frame, first item, octave, whole
// plane?, octave width, octave height); info: items per frame, LDS bytes of the tail launch, LDS bytes of the scan launch,
// tail octaves. Returns the number of segments, 0 when the geometry takes the per-octave launches.
int nm_sift_tail_plan(int width, int height, int T, int *segments, int max_segments, int info[4])
{
    if (width <= 0 || height <= 0) return 0;
    const SiftParams P(width, height);
    NmTailArgs a{};
    int radii[5] = {0, 0, 0, 0, 0};
    if (P._sigmas.size() != 5) return 0;
    for (int i = 0; i < 5; ++i) radii[i] = nm_create_kernel_for_sigma(P._sigmas[i], nullptr);
    if (!nm_tail_plan(a, width, height, P._num_octaves, T, radii)) return 0;
    for (int i = 0; segments && i < a.n_seg && i < max_segments; ++i) {
        const NmTailSeg &g = a.seg[i];
        const NmTailOct &oc = a.oct[g.slot];
        int *r = segments + 8 * i;
        r[0] = g.kind; r[1] = g.slot; r[2] = g.per_frame; r[3] = g.first_per_frame; r[4] = oc.o; r[5] = oc.whole; r[6] = oc.ow; r[7] = oc.oh;
    }
    if (info) { info[0] = a.items_per_frame; info[1] = a.lds_bytes; info[2] = a.scan_lds_bytes; info[3] = a.n_oct; }
    return a.n_seg;
}

// The reference's run-time knobs on the frame driver: SiftParams::_peak_threshold / _edge_threshold are public fields read
// per compute_keypoints call (sift/siftparams.h:97-98, siftfunctions.cu:123-125); compute_keypoints_with_mask
// (siftfunctions.cu:65-98) restricts detection to where the full-resolution mask's bilinear fetch is >= 1 (keypoint.cu:214).
int nm_sift_arena_set_params(nm_sift_arena *a, float peak_threshold, float edge_threshold)
{
    if (!a || !(edge_threshold > 0.f) || peak_threshold != peak_threshold) return (int)hipErrorInvalidValue;
    a->params._peak_threshold = peak_threshold;
    a->params._edge_threshold = edge_threshold;
    return 0;
}

int nm_sift_arena_get_params(const nm_sift_arena *a, float *peak_threshold, float *edge_threshold)
{
    if (!a) return (int)hipErrorInvalidValue;
    if (peak_threshold) *peak_threshold = a->params._peak_threshold;
    if (edge_threshold) *edge_threshold = a->params._edge_threshold;
    return 0;
}

int nm_sift_arena_set_mask(nm_sift_arena *a, const float *mask, int mask_width, int mask_height)
{
    if (!a || (mask && (mask_width != a->width || mask_height != a->height))) return (int)hipErrorInvalidValue;
    a->mask = mask;
    return 0;
}
float *nm_sift_arena_level(nm_sift_arena *a, int l) { return (a && l >= 0 && l < 6) ? a->level[l] : nullptr; }
float *nm_sift_arena_dog(nm_sift_arena *a, int d) { return (a && d >= 0 && d < 5) ? a->dog[0][d] : nullptr; }
float *nm_sift_arena_grad(nm_sift_arena *a) { return a ? a->grad[0] : nullptr; }

static int octave_pyramid(nm_sift_arena *const *as, int n, int o, int ow, int oh, bool store_top, bool decimate,
                          hipStream_t st, bool write_dog = true, bool per_octave = false, bool write_grad = true)
{
    if (o == 0) nm_prof_begin(NM_PROF_PYRAMID_O0, st);
    const size_t plane = (size_t)ow * oh;
    int rc = 0;
    for (int i = 1; i < 6 && !rc; ++i) {
        // the launch that blurs level i-1 into level i also emits DoG i-1 and, for i-1 in 1..3, the gradient plane
        // i-2 of level i-1 (compute_gradients: level l from octave[l+1], sift/siftfunctions.cu:53-63)
        NmConvBatch b{};
        b.n = n;
        for (int f = 0; f < n; ++f) {
            nm_sift_arena *a = as[f];
            float *const *lv = per_octave ? a->lev[o] : a->level;
            b.result[f] = (i < 5 || store_top) ? lv[i] : nullptr;     // level 5 is only read through DoG 4
            b.image[f] = lv[i - 1];
            b.dog[f] = write_dog ? a->dog[o][i - 1] : nullptr;
            b.grad[f] = (write_grad && i >= 2 && i <= 4) ? a->grad[o] + 2 * (size_t)(i - 2) * plane : nullptr;
            // level 3 decimated IS the next octave's level 0 (pyramidata / downsample.cu); level[0] of this octave was
            // consumed by the first launch of the sequence, so its plane can take it straight away
            b.down[f] = (decimate && i == 3) ? (per_octave ? a->lev[o + 1][0] : a->level[0]) : nullptr;
        }
        rc = nm_launch_convolve_batch(b, ow, oh, as[0]->taps[i - 1], as[0]->radii[i - 1], st);
    }
    if (o == 0) nm_prof_end(NM_PROF_PYRAMID_O0, st);
    return rc;
}

int nm_sift_octave_pyramid(nm_sift_arena *a, int ow, int oh, void *stream)
{
    if (!a || ow <= 0 || oh <= 0 || (size_t)ow * oh > a->npix) return (int)hipErrorInvalidValue;
    return octave_pyramid(&a, 1, 0, ow, oh, true, false, nm_stream(stream));
}

// Frame driver for n <= NM_MAX_BATCH equally sized frames: EVERY launch covers all frames of the call (the frame index is
// a grid dimension), so a call of 4 frames issues the same ~60 launches as a call of one, each 4x fatter. The
// scale-space chain (base blur, decimations, 5 fused Gaussian launches per octave) runs on the caller's stream; extrema +
// ordered compaction of octave o run on the first arena's side stream as soon as that octave's DoG planes exist, i.e.
// concurrently with the pyramid of octave o+1; orientation + descriptors follow on the side stream, which the caller's
// stream joins at the end. Capture-safe (events only).
int nm_sift_detect_describe_batch(nm_sift_arena *const *as, int n, const float *const *gray, float *const *desc,
                                  float *const *x, float *const *y, float *const *kpts, float *const *orients,
                                  int *const *d_num_items, void *stream)
{
    if (!as || n <= 0 || n > NM_MAX_BATCH || !gray || !desc || !x || !y) return (int)hipErrorInvalidValue;
    for (int f = 0; f < n; ++f) {
        if (!as[f] || !gray[f] || !desc[f] || !x[f] || !y[f]) return (int)hipErrorInvalidValue;
        if (as[f]->width != as[0]->width || as[f]->height != as[0]->height || as[f]->capacity != as[0]->capacity)
            return (int)hipErrorInvalidValue;
        // one set of thresholds per call (they are launch arguments); masks are per frame
        if (as[f]->params._peak_threshold != as[0]->params._peak_threshold ||
            as[f]->params._edge_threshold != as[0]->params._edge_threshold)
            return (int)hipErrorInvalidValue;
        for (int g = 0; g < f; ++g)
            if (as[g] == as[f]) return (int)hipErrorInvalidValue;
    }
    int cur = -1;
    NM_RETURN_IF(hipGetDevice(&cur));
    for (int f = 0; f < n; ++f)
        if (as[f]->device != cur) return (int)hipErrorInvalidDevice;     // arenas live on the device they were created on
    hipStream_t st = nm_stream(stream);
    const SiftParams &P = as[0]->params;
    const int W = as[0]->width, H = as[0]->height;
    NmConvBatch base{};
    base.n = n;
    for (int f = 0; f < n; ++f) { base.result[f] = as[f]->level[0]; base.image[f] = gray[f]; }
    int rc = nm_launch_convolve_batch(base, W, H, as[0]->taps_base, as[0]->base_radius, st);
    if (rc) return rc;

    NmDescribeArgs da{};
    for (int o = 0; o < 20; ++o) da.grad_off[o] = as[0]->grad_off[o];      // same geometry => same offsets in every arena
    float *kp[NM_MAX_BATCH];
    da.n = n; da.num_octaves = P._num_octaves; da.num_dogs = P._num_dog_levels;
    for (int f = 0; f < n; ++f) {
        nm_sift_arena *a = as[f];
        kp[f] = (kpts && kpts[f]) ? kpts[f] : a->kpts;
        da.book[f] = a->book; da.kpts[f] = kp[f];
        da.orients[f] = (orients && orients[f]) ? orients[f] : a->orients;
        da.desc[f] = desc[f]; da.x[f] = x[f]; da.y[f] = y[f];
    }
    hipStream_t side = as[0]->side;          // every detection / description launch covers all frames of the call
    hipStream_t dstr = as[0]->desc;
    bool forked = false, forked_desc = false;
    const bool dogs = frame_driver_writes_dog();
    // NM_FRAME_SPLIT_DESCRIBE=2 (experiment, off by default): octaves 0 and 1 hold ~98 % of a frame's keypoints; their
    // orientation + descriptor pass then starts as soon as octave 1 has been detected, on a stream of its own, beside the
    // pyramids and detections of the small octaves, and the few keypoints of the small octaves are described at the end
    // (output slots are octave-major: the passes write [oct_base[0], oct_base[2]) and [oct_base[2], num_items)). Measured
    // (MI355X, round 3): throughput unchanged (2 260 vs 2 250 frame-pairs/s), a captured graph replays in the same 445 us per
    // frame (this runtime executes a graph's branches one after the other), and eager single-frame calls, which are bound
    // by the HOST's ~55 launches on the slower boxes, got 40 us slower (545 vs 507 us): two more launches and four more
    // event operations. Not the default.
    static const int split_cfg = [] { const char *e = getenv("NM_FRAME_SPLIT_DESCRIBE"); return e ? atoi(e) : 0; }();
    int split = (split_cfg > 0 && split_cfg < P._num_octaves) ? split_cfg : 0;
    // Octave tail (nm_tail.hip): every arena of the call planned it for this geometry (same width / height => same plan).
    // It is the LATENCY path: a call of one or two frames is a chain of dependent launches that no other frame's work fills
    // (a 1080p frame: 22 launches instead of 55); in calls of many frames every per-octave launch is shared by all of them
    // and runs at a better efficiency than the tail's LDS-fused tiles (halo recomputed per tile), so those keep them
    // (16 frames per call, MI355X: 160 vs 171 us per frame). NM_FRAME_TAIL_MAX_BATCH moves the threshold.
    bool use_tail = !dogs && !split && n <= tail_max_batch() && n <= NM_TAIL_MAX_FRAMES;
    // (the first arena's plan is paired with every arena's own plane table: the plans must be the same plan -- T comes from
    // NM_FRAME_TAIL at arena creation, so arenas of one geometry CAN differ -- or the call takes the per-octave launches)
    for (int f = 0; f < n; ++f)
        use_tail = use_tail && as[f]->tail_ok && as[f]->tail.T == as[0]->tail.T && as[f]->tail.n_oct == as[0]->tail.n_oct;
    const int first_tail = use_tail ? as[0]->tail.T : P._num_octaves;
    // With the tail, the octaves < T (98 % of a frame's keypoints) are described on the description stream as soon as octave
    // T - 1 has been detected, BESIDE the tail launch; the few keypoints of the tail octaves follow behind its scans.
    if (use_tail) split = first_tail;
    NmTailArgs tail_args{};
    if (use_tail) {
        tail_args = as[0]->tail;
        tail_args.n = n;
        for (int f = 0; f < n; ++f) {
            tail_args.fr[f] = as[f]->tail_frame; tail_args.kpts[f] = kp[f];
            tail_args.d_num_items[f] = d_num_items ? d_num_items[f] : nullptr;
            tail_args.masks[f] = as[f]->mask; tail_args.any_mask |= as[f]->mask ? 1 : 0;
        }
        tail_args.mask_w = W; tail_args.mask_h = H;
        tail_args.peak = P._peak_threshold; tail_args.edge = P._edge_threshold; tail_args.sigma0 = P._sigma_0;
        tail_args.num_dogs = P._num_dog_levels; tail_args.capacity = as[0]->capacity;
        tail_args.state = as[0]->tail_state;
    }
    auto body = [&]() -> int {
        for (int o = 0; o < P._num_octaves; ++o) {
            const int ow = W >> o, oh = H >> o;
            const float xper = (float)std::pow(2.0, o);
            if (o >= first_tail) {                 // the tail launch below covers this octave; the describe pass needs its geometry
                for (int f = 0; f < n; ++f) da.grad0[f] = as[f]->grad[0];
                da.geom[o].ow = ow; da.geom[o].oh = oh; da.geom[o].xper = xper;
                continue;
            }
            int e = octave_pyramid(as, n, o, ow, oh, !dogs, o + 1 < P._num_octaves, st, dogs, true);
            if (e) return e;
            NM_RETURN_IF(hipEventRecord(as[0]->ev_pyr[o], st));
            if (use_tail && o + 1 == first_tail) {
                // The tail launch (levels, gradients, detection of the octaves >= T) goes to the CALLER's stream, straight behind
                // the pyramid of octave T - 1 whose decimated level 3 seeds it -- issued BEFORE this octave's detection launches
                // so that the host does not hold it back -- and runs beside the detection of the octaves < T on the side stream.
                e = nm_launch_tail(tail_args, st);
                if (e) return e;
            }
            NM_RETURN_IF(hipStreamWaitEvent(side, as[0]->ev_pyr[o], 0));
            forked = true;

            const int nseg = nm_divup(ow, NM_DET_SEG_W);
            const int n_blocks = oh * nseg;
            NmDetectArgs d{};
            NmScanArgs s{};
            NmGatherArgs g{};
            d.n = s.n = g.n = n;
            d.ow = ow; d.oh = oh; d.peak = P._peak_threshold; d.edge = P._edge_threshold; d.xper = xper;
            d.sigma0 = P._sigma_0; d.num_dogs = P._num_dog_levels; d.stage_stride = as[0]->stage_stride;
            d.n_blocks = n_blocks; d.nseg = nseg;
            d.from_levels = dogs ? 0 : 1;
            d.plane_stride = as[0]->plane_stride[o];
            d.mask_w = W; d.mask_h = H;
            s.n_blocks = n_blocks; s.octave = o;
            g.stage_stride = as[0]->stage_stride; g.n_blocks = n_blocks; g.octave = o;
            s.capacity = as[0]->capacity; g.capacity = as[0]->capacity;
            for (int f = 0; f < n; ++f) {
                nm_sift_arena *a = as[f];
                d.plane0[f] = dogs ? a->dog[o][0] : a->lev[o][0];
                d.staging[f] = a->staging; d.counts[f] = a->counts;
                d.masks[f] = a->mask; d.any_mask |= a->mask ? 1 : 0;
                s.counts[f] = a->counts; s.offsets[f] = a->offsets; s.book[f] = a->book;
                s.d_num_items[f] = d_num_items ? d_num_items[f] : nullptr;
                g.staging[f] = a->staging; g.counts[f] = a->counts; g.offsets[f] = a->offsets; g.book[f] = a->book;
                g.kpts[f] = kp[f];
                da.grad0[f] = a->grad[0];
            }
            e = nm_launch_detect_octave(d, s, g, side);
            if (e) return e;
            da.geom[o].ow = ow; da.geom[o].oh = oh; da.geom[o].xper = xper;
            // (Describing the octaves below T - 1 even earlier, on the description stream beside octave T - 1's pyramid and
            // detection, was measured: 435 instead of 282 us per frame -- the descriptor kernel fills every CU's wave slots and
            // the tail launch's 1 024-thread workgroups, issued at the same time, wait for whole CUs: 158 instead of 98 us.)
            if (split && o + 1 == split) {
                NM_RETURN_IF(hipEventRecord(as[0]->ev_det, side));
                da.o_begin = 0; da.o_end = split;
                if (use_tail) {
                    // with the tail the side stream has nothing left to detect: the octaves < T are described right there
                    // (one stream hand-over less on the path base blur -> ... -> descriptors), beside the tail on the caller's
                    e = nm_launch_frame_describe(da, side);
                } else {
                    NM_RETURN_IF(hipStreamWaitEvent(dstr, as[0]->ev_det, 0));
                    forked_desc = true;
                    e = nm_launch_frame_describe(da, dstr);
                }
                if (e) return e;
            }
        }
        da.o_begin = split; da.o_end = P._num_octaves;
        if (use_tail) {
            // The tail's book-keeping scans + gathers continue octave T - 1's book (ev_det: recorded on the side stream behind
            // that octave's detection, long reached by now) and stay on the CALLER's stream, straight behind the tail launch
            // -- an event hand-over to another stream costs ~10 us at the end of the chain -- as does the description of the
            // tail octaves' few keypoints.
            NM_RETURN_IF(hipStreamWaitEvent(st, as[0]->ev_det, 0));
            const int e = nm_launch_tail_scan(tail_args, st);
            if (e) return e;
            return nm_launch_frame_describe(da, st);
        }
        return nm_launch_frame_describe(da, side);
    };
    rc = body();
    if (forked_desc) {
        // the description stream joins the CALLER's stream directly (also on an error path, like the side stream below).
        // Joining it into the side stream it was forked from -- an equivalent DAG -- makes this ROCm's stream capture
        // segfault (tools/capture_shapes.py: fork s2 -> s3, join s3 -> s2 -> s1 crashes, s3 -> s1 and s2 -> s1 works).
        const hipError_t e1 = hipEventRecord(as[0]->ev_desc, dstr);
        const hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(st, as[0]->ev_desc, 0) : e1;
        if (!rc && e2 != hipSuccess) rc = (int)e2;
    }
    if (forked) {
        // also on an error path: the side stream must always be joined back, or a stream capture would be left with an
        // unjoined fork and the next call on these arenas could overtake side-stream work still in flight
        const hipError_t e1 = hipEventRecord(as[0]->ev_join, side);
        const hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(st, as[0]->ev_join, 0) : e1;
        if (!rc && e2 != hipSuccess) rc = (int)e2;
    }
    return rc;
}

// The scale-space chain of nm_sift_detect_describe_batch alone (base blur, then per octave the five fused Gaussian + DoG +
// gradient launches with the decimation in the level-3 epilogue), exactly the launches the frame driver issues on the
// caller's stream, without detection / description: what bench.py times for the whole-pyramid roofline.
int nm_sift_scale_space_batch_ex(nm_sift_arena *const *as, int n, const float *const *gray, int write_dog, void *stream)
{
    if (!as || n <= 0 || n > NM_MAX_BATCH || !gray) return (int)hipErrorInvalidValue;
    int cur = -1;
    NM_RETURN_IF(hipGetDevice(&cur));
    for (int f = 0; f < n; ++f) {
        if (!as[f] || !gray[f] || as[f]->device != cur) return (int)hipErrorInvalidValue;
        if (as[f]->width != as[0]->width || as[f]->height != as[0]->height) return (int)hipErrorInvalidValue;
    }
    hipStream_t st = nm_stream(stream);
    const SiftParams &P = as[0]->params;
    const int W = as[0]->width, H = as[0]->height;
    NmConvBatch base{};
    base.n = n;
    for (int f = 0; f < n; ++f) { base.result[f] = as[f]->level[0]; base.image[f] = gray[f]; }
    int rc = nm_launch_convolve_batch(base, W, H, as[0]->taps_base, as[0]->base_radius, st);
    // write_dog bit 0: materialise the DoG planes; bit 1: leave the gradient planes out (measurement of the plain
    // Gaussian + DoG chain, the 108 B per octave-pixel of SURVEY.md 8(d), without the fused 36 B of gradients)
    const bool dogs = (write_dog & 1) != 0, grads = (write_dog & 2) == 0;
    for (int o = 0; !rc && o < P._num_octaves; ++o)
        rc = octave_pyramid(as, n, o, W >> o, H >> o, !dogs, o + 1 < P._num_octaves, st, dogs, true, grads);
    return rc;
}

int nm_sift_scale_space_batch(nm_sift_arena *const *as, int n, const float *const *gray, void *stream)
{
    return nm_sift_scale_space_batch_ex(as, n, gray, 1, stream);
}

int nm_sift_detect_describe(nm_sift_arena *a, const float *gray, float *desc, float *x, float *y, float *kpts,
                            float *orients, int *d_num_items, void *stream)
{
    return nm_sift_detect_describe_batch(&a, 1, &gray, &desc, &x, &y, &kpts, &orients, &d_num_items, stream);
}

}  // extern "C"
