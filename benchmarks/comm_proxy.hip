// A kernel with the FOOTPRINT of an RCCL collective kernel -- a few workgroups ("channels") that each stream a slice of a buffer for a
// long time -- for concurrency experiments on one GPU (benchmarks/stream_concurrency.py, benchmarks/rccl_overlap.py).  Not part of the
// library.  hipcc --offload-arch=gfx950 -O3 -fPIC -shared benchmarks/comm_proxy.hip -o benchmarks/_alt/libcomm_proxy.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void proxy_kernel(const u32x4_t* src, u32x4_t* dst, long nvec, int repeats) {
    const long stride = (long)gridDim.x * 256;
    for (int r = 0; r < repeats; ++r)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) dst[i] = src[i];
}
extern "C" int comm_proxy_copy(const void* src, void* dst, long bytes, int workgroups, int repeats, void* stream) {
    hipLaunchKernelGGL(proxy_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (u32x4_t*)dst, bytes / 16, repeats);
    return (int)hipGetLastError();
}
