// pyramidata.cpp -- PyramidData on top of the C ABI (reference: src/gpu/sift/pyramidata.cu:18-123).
#include "../pyramidata.h"

#include <chrono>
#include <thread>

#include "../../../include/nm_abi.h"
#include "../exception.h"

PyramidData::PyramidData(const SiftParams &params) : _base_radius(0), _num_octaves(0), _num_dogs(0), _num_kernels(0), _lazy_octave(-1)
{
    for (auto &d : _dirty) d = 0;
    initialize(params);
}

void PyramidData::initialize(const SiftParams &params)
{
    clear();
    _num_octaves = params._level_max - params._level_min + 1;
    if (_num_octaves > 20) RUNTIME_EXCEPTION("Maximum bumber of levels is 20.");
    const size_t num_pixels = (size_t)params._width * params._height;

    for (int i = 0; i < _num_octaves; ++i) _octave[i] = nm::device_vector<float>(num_pixels);
    _num_dogs = params._level_max - params._level_min;
    for (int i = 0; i < _num_dogs; ++i) _dog[i] = nm::device_vector<float>(num_pixels);
    const float4 invalid = make_float4(-1, -1, -1, -1);
    for (int i = 0; i < params._num_dog_levels; ++i) {
        _key_pts[i] = nm::device_vector<float4>(num_pixels, invalid);
        _collated_kpts[i] = nm::device_vector<float4>(num_pixels, invalid);
        // room for every keypoint a level can hold, size 0: compute_orientations then sets the size without knowing it on the
        // host (lazy_count.h; the reference re-creates the vector per level with the count it has just read back,
        // pyramidata.cu:90). A keypoint is a STRICT extremum of its 3 x 3 x 3 neighbourhood (keypoint.cu:195-196): two pixels
        // that touch cannot both be maxima (or both minima), so a level holds at most ceil(w/2) ceil(h/2) of each -- half the
        // pixels, not all of them (50 MB per PyramidData at 1080p with one entry per pixel).
        _orientations[i] = nm::device_vector<float2>();
        _orientations[i].reserve_uninitialized(nm_keypoint_bound(params._width, params._height));
    }
    _lazy_rec.reset(); _lazy_octave = -1;
    _grad = nm::device_vector<float2>(num_pixels * _num_dogs, make_float2(0, 0));
    _buffer = nm::device_vector<float>(num_pixels);
    _count = nm::device_vector<int>(4);
    _compact_ws = nm::device_vector<int>(nm_compact3_workspace_bytes((int)num_pixels) / sizeof(int) + 1);
    for (auto &d : _dirty) d = 0;                      // every dense map is all -1 now
    (void)_host_counts.get();
    generate_kernels(params);
}

void PyramidData::clear()
{
    for (int i = 0; i < _num_octaves; ++i) _octave[i].clear();
    _grad.clear();
    for (int i = 0; i < _num_dogs; ++i) _dog[i].clear();
    for (int i = 0; i < _num_kernels; ++i) _kernels[i].clear();
    for (int i = 0; i < _num_dogs - 2; ++i) {
        _key_pts[i].clear();
        _collated_kpts[i].clear();
        _orientations[i].clear();
    }
    _lazy_rec.reset(); _lazy_octave = -1;
    _buffer.clear();
    _base_kernel.clear();
    _kernel_radii.clear();
    _num_octaves = _num_dogs = _num_kernels = 0;
}

static bool nm_stream_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Wait until the words `need` of a slot have left -1: a short spin (the usual case: the kernel has long finished), then
// yielding; after 20 s without the words the device is drained once and, if they are still missing, the error convention of
// the C++ layer applies (the producing launch never ran: a failed launch or a destroyed stream's dropped work).
static void nm_wait_words(volatile int *host, unsigned need, const char *what)
{
    auto arrived = [&]() {
        for (int i = 0; i < 4; ++i)
            if (((need >> i) & 1u) && host[i] == -1) return false;
        return true;
    };
    for (int spin = 0; spin < 2000; ++spin)
        if (arrived()) return;
    const auto t0 = std::chrono::steady_clock::now();
    while (!arrived()) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            nm_check((int)hipDeviceSynchronize(), what);
            if (!arrived()) nm_check((int)hipErrorNotReady, what);
            return;
        }
    }
}

void nm::pending_counts::resolve()
{
    if (_resolved.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lock(_m);
    if (_resolved.load(std::memory_order_relaxed)) return;
    nm_wait_words(host, need, "keypoint count read-back");
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < 4; ++i) values[i] = ((need >> i) & 1u) ? host[i] : 0;
    _resolved.store(true, std::memory_order_release);
}

std::shared_ptr<nm::pending_counts> nm::pinned_ring::take(hipStream_t stream, unsigned need, int **dev_words)
{
    if (!_host) {
        nm_check((int)hipHostMalloc(reinterpret_cast<void **>(&_host), SLOTS * 4 * sizeof(int), hipHostMallocMapped),
                 "pinned counter ring allocation");
        void *d = nullptr;
        nm_check((int)hipHostGetDevicePointer(&d, _host, 0), "pinned counter ring mapping");
        _dev = static_cast<int *>(d);
        for (int i = 0; i < SLOTS * 4; ++i) _host[i] = 0;
    }
    const int slot = _next;
    _next = (_next + 1) % SLOTS;
    volatile int *words = _host + 4 * slot;
    if (!nm_stream_capturing(stream)) {
        // (no host waits while a capture is open: a captured launch does not run now, its words arrive at the first replay)
        if (auto old = _last[slot].lock()) old->resolve();    // 64 calls later and still unread: latch it before the words are reused
        // the slot's last producer must have WRITTEN even if its record was dropped unread: it would otherwise overwrite what
        // the new owner's kernel is about to write
        if (_need[slot]) nm_wait_words(words, _need[slot], "pinned counter ring slot");
        for (int i = 0; i < 4; ++i) words[i] = -1;
        std::atomic_thread_fence(std::memory_order_release);
    }
    _need[slot] = need;
    auto rec = std::make_shared<pending_counts>(words, need);
    _last[slot] = rec;
    *dev_words = _dev + 4 * slot;
    return rec;
}

void nm::pinned_ring::release()
{
    for (int i = 0; i < SLOTS; ++i) {
        if (auto old = _last[i].lock()) old->resolve();       // records outlive the ring: give them their values first
        // a producer whose record was dropped may still be about to write into the memory freed below
        if (_host && _need[i]) nm_wait_words(_host + 4 * i, _need[i], "pinned counter ring release");
        _need[i] = 0;
    }
    if (_host) (void)hipHostFree(_host);
    _host = _dev = nullptr;
}

void PyramidData::gpu_collate_keypoints_for_octave_dev(int num_pixels, hipStream_t stream)
{
    const float *dense[3];
    float *out[3];
    for (int l = 0; l < 3; ++l) {
        dense[l] = reinterpret_cast<const float *>(_key_pts[l].data());
        out[l] = reinterpret_cast<float *>(_collated_kpts[l].data());
    }
    nm_check(nm_compact_keypoints3(dense, num_pixels, out, _count.data(), _compact_ws.data(), stream), "Keypoint collation failed");
}

void PyramidData::gpu_collate_keypoints_for_level(int level, int num_pixels)
{
    _lazy_rec.reset(); _lazy_octave = -1;
    nm_check(nm_compact_keypoints(reinterpret_cast<const float *>(_key_pts[level].data()), num_pixels,
                                  reinterpret_cast<float *>(_collated_kpts[level].data()), _count.data(),
                                  _compact_ws.data(), nullptr),
             "Keypoint collation failed");
    int new_size = 0;   // the reference's copy_if returns a host iterator: same implicit synchronisation point
    nm_check((int)hipMemcpy(&new_size, _count.data(), sizeof(int), hipMemcpyDeviceToHost), "Keypoint count D2H");
    _orientations[level].resize_uninitialized((size_t)new_size);     // (-1,-1) per keypoint, without re-allocating
    if (new_size > 0) {
        nm_check(nm_fill_u32(_orientations[level].data(), (size_t)new_size * 2, 0xBF800000u, nullptr), "orientation reset");
        nm_check((int)hipStreamSynchronize(nullptr), "orientation reset");
    }
}

void PyramidData::gpu_collate_keypoints_for_octave(int num_pixels, int counts[3], hipStream_t stream)
{
    _lazy_rec.reset(); _lazy_octave = -1;
    const float *dense[3];
    float *out[3];
    for (int l = 0; l < 3; ++l) {
        dense[l] = reinterpret_cast<const float *>(_key_pts[l].data());
        out[l] = reinterpret_cast<float *>(_collated_kpts[l].data());
    }
    nm_check(nm_compact_keypoints3(dense, num_pixels, out, _count.data(), _compact_ws.data(), stream),
             "Keypoint collation failed");
    int *host = _host_counts.get();
    nm_check((int)hipMemcpyAsync(host, _count.data(), 3 * sizeof(int), hipMemcpyDeviceToHost, stream), "Keypoint count D2H");
    nm_check((int)hipStreamSynchronize(stream), "Keypoint count D2H");
    for (int l = 0; l < 3; ++l) counts[l] = host[l];
}

void PyramidData::generate_kernels(const SiftParams &params)
{
    create_kernel_for_sigma(params._base_smooth, _base_kernel, _base_radius);
    _num_kernels = (int)params._sigmas.size();
    for (int i = 0; i < _num_kernels; ++i) {
        int rad = 0;
        create_kernel_for_sigma(params._sigmas[i], _kernels[i], rad);
        _kernel_radii.push_back(rad);
    }
}

void PyramidData::create_kernel_for_sigma(float sigma, nm::device_vector<float> &result, int &radius)
{
    radius = nm_create_kernel_for_sigma(sigma, nullptr);
    std::vector<float> taps(2 * radius + 1);
    nm_create_kernel_for_sigma(sigma, taps.data());
    result = nm::device_vector<float>(taps);
}
