// buffer_range.hip -- what the buffer descriptor's range check covers on this device: voffset only, or voffset + soffset?
// (LLVM documents soffset as "excluded from bounds checking"; the matcher's tile loads and the distance pass's stores are
// written for that reading. This probe confirms it on the hardware: a load whose voffset is in range but whose soffset
// carries it past num_records must return DATA if soffset is excluded, ZERO if it is included.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/buffer_range.hip -o buffer_range && ./buffer_range
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(const unsigned *buf, unsigned *out)
{
    // descriptor over the first 256 bytes of a 4 KiB allocation filled with 0xA5A5A5A5
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, 256, 0x00020000);
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);          // in range
    out[1] = __builtin_amdgcn_raw_buffer_load_b32(rs, 1024, 0, 0);       // voffset out of range -> 0
    out[2] = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 1024, 0);       // soffset past the range: 0 if checked, data if not
    out[3] = __builtin_amdgcn_raw_buffer_load_b32(rs, 252, 1024, 0);     // last in-range voffset + soffset
    out[4] = __builtin_amdgcn_raw_buffer_load_b32(rs, 0x80000000u, 16, 0);   // the distance pass's poisoned voffset
}

int main()
{
    unsigned *buf, *out, h[5];
    hipMalloc(&buf, 4096); hipMalloc(&out, 64);
    hipMemset(buf, 0xA5, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, buf, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("in range %08x | voffset out %08x | soffset past range %08x (%s) | voffset 252 + soffset %08x | poisoned voffset %08x\n", h[0], h[1],
           h[2], h[2] ? "soffset is EXCLUDED from the range check" : "soffset is INCLUDED in the range check", h[3], h[4]);
    return 0;
}
