// Host launcher + C-ABI of the ViT attention forward (attention_kernels.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <attention_kernels.h>

extern "C" int vrwkv_attention_fwd_bf16(int B, int L, int H, int D, const void* q, const void* k, const void* v,
                                        long stride_b, long stride_l, long stride_h, void* o, void* stream) {
    if (B <= 0 || L <= 0 || H <= 0 || !q || !k || !v || !o) return VRWKV_EINVAL;
    if (D != 64 && D != 72) return VRWKV_ESHAPE;
    if ((stride_b | stride_l | stride_h) % 8 != 0) return VRWKV_EALIGN;       // 16-byte rows
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(o)) & 15u) return VRWKV_EALIGN;
    vattn::Args p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, stride_b, stride_l, stride_h, L, H,
                  (float)(1.4426950408889634 / sqrt((double)D))};
    const dim3 grid((unsigned)((L + 63) / 64), (unsigned)(B * H));
    if (D == 64) hipLaunchKernelGGL(vattn::fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(vattn::fwd_kernel<72>, grid, dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
