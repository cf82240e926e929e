// cudatimer.h -- stream timer (drop-in for NiftyMatch src/gpu/utils/cudatimer.h; hipEvent pair instead of cudaEvent).
#ifndef __CUDA_TIMER_H__
#define __CUDA_TIMER_H__

#include <hip/hip_runtime_api.h>

class CudaTimer {
public:
    CudaTimer(hipStream_t stream = 0);
    ~CudaTimer();
    void start();          //!< record the start event on the stream
    float stop();          //!< record the stop event, wait for it, return elapsed milliseconds

private:
    hipStream_t _stream;
    hipEvent_t _start, _stop;
};

#endif
