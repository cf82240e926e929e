// gpuutils.cpp -- the `gpuutils` library: CudaTimer and CudaUtils on HIP (reference: src/gpu/utils/cudatimer.cu:3-22,
// src/gpu/utils/cudautils.cpp:8-28).
#include "../cudatimer.h"
#include "../cudautils.h"
#include "../exception.h"

CudaTimer::CudaTimer(hipStream_t stream) : _stream(stream), _start(nullptr), _stop(nullptr)
{
    nm_check((int)hipEventCreate(&_start), "CudaTimer event");
    nm_check((int)hipEventCreate(&_stop), "CudaTimer event");
}

CudaTimer::~CudaTimer()
{
    if (_start) (void)hipEventDestroy(_start);
    if (_stop) (void)hipEventDestroy(_stop);
}

void CudaTimer::start() { nm_check((int)hipEventRecord(_start, _stream), "CudaTimer start"); }

float CudaTimer::stop()
{
    float ms = 0.f;
    nm_check((int)hipEventRecord(_stop, _stream), "CudaTimer stop");
    nm_check((int)hipEventSynchronize(_stop), "CudaTimer stop");
    nm_check((int)hipEventElapsedTime(&ms, _start, _stop), "CudaTimer elapsed");
    return ms;
}

int CudaUtils::_max_gflops_device_id = -1;

void CudaUtils::setup_CUDA()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) RUNTIME_EXCEPTION("No HIP device found.");
    int best = 0, best_cus = -1;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount * p.clockRate > best_cus) {
            best_cus = p.multiProcessorCount * p.clockRate;
            best = d;
        }
    }
    _max_gflops_device_id = best;
    if (hipSetDevice(best) != hipSuccess) RUNTIME_EXCEPTION("Could not set the HIP device.");
}
