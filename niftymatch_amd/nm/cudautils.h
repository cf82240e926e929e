// cudautils.h -- device selection (drop-in for NiftyMatch src/gpu/utils/cudautils.h). The reference also binds the
// device to OpenGL (cudautils.cpp:25); there is no GL interop here.
#ifndef __CUDA_UTILS_H__
#define __CUDA_UTILS_H__

class CudaUtils {
public:
    //! Select the device with the most compute units and make it current. Throws Exception<std::runtime_error>
    //! when no device is present.
    static void setup_CUDA();
    static int max_gflops_device_id() { return _max_gflops_device_id; }

private:
    static int _max_gflops_device_id;
};

#endif
