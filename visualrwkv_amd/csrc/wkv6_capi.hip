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
    // wkv6_bwd_v2.h forms 32-bit byte offsets inside a tensor (the largest is the fp32 ew): a launch that reaches 4 GiB runs as batch slices of the
    // same kernel on offset pointers (every tensor is batch-major; gu is per sample); one sample that large goes to the four-wave kernel (64-bit)
    const unsigned long long per_sample = (unsigned long long)T * C * 4ull, limit = 1ull << 32;
    int bmax = B;
    bool v1 = g_bwd6_variant == 1;
    if (!v1 && (unsigned long long)B * per_sample >= limit) {
        bmax = (int)((limit - 1) / per_sample);
        if (bmax < 1) { v1 = true; bmax = B; }
    }
    void (*kern)(wkv6c::Bwd6Args) = &wkv6v2::bwd6_kernel_v2;
    if (!v1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv6v2::Lds6V2));
        if (e != hipSuccess) return (int)e;
    }
    const size_t act = (size_t)T * C, ck = (size_t)H * ((T + 15) / 16) * 64 * 64;
    for (int b0 = 0; b0 < B; b0 += bmax) {
        const int nb = B - b0 < bmax ? B - b0 : bmax;
        const size_t o = (size_t)b0 * act;
        wkv6c::Bwd6Args p{T, H, (const uint16_t*)r + o, (const uint16_t*)k + o, (const uint16_t*)v + o, ew + o, (const uint16_t*)u,
                          (const uint16_t*)gy + o, s_ckpt + (size_t)b0 * ck, (uint16_t*)gr + o, (uint16_t*)gk + o, (uint16_t*)gv + o, (uint16_t*)gw + o,
                          (uint16_t*)gu + (size_t)b0 * C};
        if (v1) hipLaunchKernelGGL(wkv6c::bwd6_kernel, dim3((unsigned)((long)nb * H)), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(kern, dim3((unsigned)((long)nb * H)), dim3(768), sizeof(wkv6v2::Lds6V2), (hipStream_t)stream, p);
        const int rc2 = done6();
        if (rc2) return rc2;
    }
    return VRWKV_OK;
}

}  // extern "C"
