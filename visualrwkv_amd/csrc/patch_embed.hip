// Host launcher + C-ABI of the implicit-GEMM patch embedding (patch_embed_kernels.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <patch_embed_kernels.h>

namespace {
template <int P>
int launch(const vpe::Args& a, int B, hipStream_t stream) {
    using G = vpe::Geo<P>;
    auto kern = vpe::kernel<P>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * (a.Mimg / 64))), dim3(256), G::LDS_BYTES, stream, a);
    e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
}  // namespace

extern "C" int vrwkv_patch_embed_kp(int P) { return P > 0 ? (3 * P * P + 31) / 32 * 32 : VRWKV_EINVAL; }

extern "C" int vrwkv_patch_embed_bf16(int B, int Himg, int Wimg, int P, int N, const void* pixels, const void* w_padded,
                                      const void* bias, const void* pos, void* out, int tokens_per_image, int prefix,
                                      void* stream) {
    if (B <= 0 || !pixels || !w_padded || !out || prefix < 0) return VRWKV_EINVAL;
    if (P != 14 && P != 16) return VRWKV_ESHAPE;
    if (Himg % P || Wimg % P || N % 32) return VRWKV_ESHAPE;
    const int gh = Himg / P, gw = Wimg / P, M = gh * gw;
    if (M % 64 || tokens_per_image < prefix + M) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(pixels) & 3u) || (reinterpret_cast<uintptr_t>(w_padded) & 15u) ||
        (reinterpret_cast<uintptr_t>(out) & 7u) || (reinterpret_cast<uintptr_t>(bias) & 7u) || (reinterpret_cast<uintptr_t>(pos) & 7u))
        return VRWKV_EALIGN;
    vpe::Args a{(const uint16_t*)pixels, (const uint16_t*)w_padded, (const uint16_t*)bias, (const uint16_t*)pos, (uint16_t*)out,
                Himg, Wimg, N, gw, M, tokens_per_image, prefix};
    if (P == 14) return launch<14>(a, B, (hipStream_t)stream);
    return launch<16>(a, B, (hipStream_t)stream);
}
