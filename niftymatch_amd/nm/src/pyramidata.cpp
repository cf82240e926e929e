// pyramidata.cpp -- PyramidData on top of the C ABI (reference: src/gpu/sift/pyramidata.cu:18-123).
#include "../pyramidata.h"

#include "../../../include/nm_abi.h"
#include "../exception.h"

PyramidData::PyramidData(const SiftParams &params) : _base_radius(0), _num_octaves(0), _num_dogs(0), _num_kernels(0)
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
    }
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
    _buffer.clear();
    _base_kernel.clear();
    _kernel_radii.clear();
    _num_octaves = _num_dogs = _num_kernels = 0;
}

void PyramidData::gpu_collate_keypoints_for_level(int level, int num_pixels)
{
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
