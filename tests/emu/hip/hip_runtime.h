// Stand-in for <hip/hip_runtime.h> when a whole .hip translation unit (kernels + C entry points) is compiled for the host lockstep
// emulator: kernel launches become emu::launch, streams and error codes are inert.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <hip_emu.h>
typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu::launch(dim3(grid), dim3(block), [&] { kern(__VA_ARGS__); })
// one fiber runs at a time, so a plain read-modify-write is atomic here
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated HIP error"; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
