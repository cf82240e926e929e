// pyramidata.cpp -- PyramidData on top of the C ABI (reference: src/gpu/sift/pyramidata.cu:18-123).
#include "../pyramidata.h"

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
        // room for every pixel's keypoint, size 0: compute_orientations then sets the size without knowing it on the host
        // (lazy_count.h; the reference re-creates the vector per level with the count it has just read back, pyramidata.cu:90)
        _orientations[i] = nm::device_vector<float2>();
        _orientations[i].reserve_uninitialized(num_pixels);
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

void nm::pending_counts::resolve()
{
    if (resolved) return;
    // the kernel that fills the words runs on `stream`; a stream the client has destroyed since is covered by the device
    if (hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipGetLastError();
        nm_check((int)hipDeviceSynchronize(), "keypoint count read-back");
    }
    for (int i = 0; i < 4; ++i) values[i] = host[i];
    resolved = true;
}

std::shared_ptr<nm::pending_counts> nm::pinned_ring::take(hipStream_t stream, int **dev_words)
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
    if (auto old = _last[slot].lock()) old->resolve();        // 64 calls later and still unread: latch it before the words are reused
    auto rec = std::make_shared<pending_counts>(_host + 4 * slot, stream);
    _last[slot] = rec;
    *dev_words = _dev + 4 * slot;
    return rec;
}

void nm::pinned_ring::release()
{
    for (int i = 0; i < SLOTS; ++i)
        if (auto old = _last[i].lock()) old->resolve();       // records outlive the ring: give them their values first
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
