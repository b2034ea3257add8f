// C-ABI launchers of the WKV6 kernels (csrc/wkv6_chunked.h).  Same conventions as wkv7_capi.hip: plain device
// pointers, sizes and a hipStream_t; nothing is allocated; 0 / positive hipError_t / negative VRWKV_E*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <wkv6_chunked.h>
#include <wkv6_bwd_v2.h>

namespace {
inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
inline int check6(int B, int T, int C, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return VRWKV_EINVAL;
    if (C != H * 64) return VRWKV_ESHAPE;               // head size 64 (RWKV_HEAD_SIZE_A, -D_N_=64 in the reference build)
    return VRWKV_OK;
}
int g_bwd6_variant = -1;            // -1: default (2); 1: bwd6_kernel (wkv6_chunked.h), 2: bwd6_kernel_v2 (wkv6_bwd_v2.h)
inline int done6() { hipError_t e = hipGetLastError(); return e == hipSuccess ? VRWKV_OK : (int)e; }
}  // namespace

extern "C" {

int vrwkv_wkv6_set_backward_variant(int variant) {
    if (variant != -1 && variant != 1 && variant != 2) return VRWKV_EINVAL;
    g_bwd6_variant = variant;
    return VRWKV_OK;
}

long vrwkv_wkv6_ckpt_floats(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    return (long)B * H * ((T + 15) / 16) * 64 * 64;
}

int vrwkv_wkv6_forward_bf16(int B, int T, int C, int H, const void* r, const void* k, const void* v, const float* ew,
                            const void* u, void* y, float* s_ckpt, void* stream) {
    int rc = check6(B, T, C, H);
    if (rc) return rc;
    if (!r || !k || !v || !ew || !u || !y) return VRWKV_EINVAL;
    if (misaligned16(r) || misaligned16(k) || misaligned16(v) || misaligned16(ew) || misaligned16(u) || misaligned16(y) ||
        (s_ckpt && misaligned16(s_ckpt)))
        return VRWKV_EALIGN;
    wkv6c::Fwd6Args p{T, H, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, ew, (const uint16_t*)u, (uint16_t*)y, s_ckpt};
    hipLaunchKernelGGL(wkv6c::fwd6_kernel, dim3((unsigned)((long)B * H)), dim3(256), 0, (hipStream_t)stream, p);
    return done6();
}

int vrwkv_wkv6_backward_bf16(int B, int T, int C, int H, const void* r, const void* k, const void* v, const float* ew,
                             const void* u, const void* gy, const float* s_ckpt, void* gr, void* gk, void* gv, void* gw,
                             void* gu, void* stream) {
    int rc = check6(B, T, C, H);
    if (rc) return rc;
    if (!r || !k || !v || !ew || !u || !gy || !s_ckpt || !gr || !gk || !gv || !gw || !gu) return VRWKV_EINVAL;
    if (misaligned16(r) || misaligned16(k) || misaligned16(v) || misaligned16(ew) || misaligned16(u) || misaligned16(gy) ||
        misaligned16(s_ckpt) || misaligned16(gr) || misaligned16(gk) || misaligned16(gv) || misaligned16(gw) || misaligned16(gu))
        return VRWKV_EALIGN;
    wkv6c::Bwd6Args p{T, H, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, ew, (const uint16_t*)u,
                      (const uint16_t*)gy, s_ckpt, (uint16_t*)gr, (uint16_t*)gk, (uint16_t*)gv, (uint16_t*)gw, (uint16_t*)gu};
    const bool fits32 = (unsigned long long)B * T * C * 4ull < (1ull << 32);      // wkv6_bwd_v2.h forms 32-bit byte offsets (ew is fp32)
    if (g_bwd6_variant == 1 || !fits32) {
        hipLaunchKernelGGL(wkv6c::bwd6_kernel, dim3((unsigned)((long)B * H)), dim3(256), 0, (hipStream_t)stream, p);
        return done6();
    }
    void (*kern)(wkv6c::Bwd6Args) = &wkv6v2::bwd6_kernel_v2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv6v2::Lds6V2));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * H)), dim3(768), sizeof(wkv6v2::Lds6V2), (hipStream_t)stream, p);
    return done6();
}

}  // extern "C"
