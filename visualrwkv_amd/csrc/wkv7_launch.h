// State and helpers shared by the WKV7 launchers (wkv7_capi.hip) and the profiling entry point (wkv7_profile.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/visualrwkv_hip.h"
#include <wkv7_kernels.h>

namespace wkv7launch {

extern std::atomic<int> g_fwd_variant, g_bwd_variant;      // vrwkv_wkv7_set_{forward,backward}_variant; -1 = default
// few heads (B*H <= 128: at most half of the 256 CUs would be busy): the forward runs two workgroups per head, 32 value rows each
constexpr long FWD_ISPLIT_MAX_HEADS = 128;
// T chain on the bf16 matrix core (2) + producer priority 2 (4; same-box A/B: 1.18 -> 1.09 ms) + priorities swapped in
// segment 1, where the producers have ~1.2k cycles of slack per chunk and the consumers none (128; 1.09 -> 1.04 ms)
constexpr int BWD_V5_MODE = 2 + 4 + 128;

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
inline int check_common(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return VRWKV_EINVAL;
    if (T % VRWKV_CHUNK_LEN != 0) return VRWKV_ESHAPE;   // cuda_backward asserts this, wkv7_cuda.cu:136
    return VRWKV_OK;
}
inline int finish_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
// dynamic-LDS launch of a kernel taking one argument block
template <class Args>
inline int launch_lds(void (*kern)(Args), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const Args& p) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
    return finish_launch();
}

}  // namespace wkv7launch
