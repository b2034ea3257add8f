// Host launcher + C-ABI of the tower image transform (image_kernels.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <image_kernels.h>

extern "C" int vrwkv_resize_normalize_u8(int H, int W, const void* src_hwc_u8, int S, const float* mean3, const float* std3,
                                         void* dst_chw, int dst_is_f32, void* stream) {
    if (H <= 0 || W <= 0 || S <= 0 || !src_hwc_u8 || !dst_chw || !mean3 || !std3) return VRWKV_EINVAL;
    vimg::Args a{};
    a.src = (const uint8_t*)src_hwc_u8; a.dst = dst_chw; a.H = H; a.W = W; a.S = S; a.out_f32 = dst_is_f32 ? 1 : 0;
    a.scale_x = (float)W / (float)S; a.scale_y = (float)H / (float)S;
    a.taps_x = vimg::max_taps(a.scale_x); a.taps_y = vimg::max_taps(a.scale_y);
    for (int c = 0; c < 3; ++c) {
        if (!(std3[c] > 0.f)) return VRWKV_EINVAL;
        a.mul[c] = 1.f / (255.f * std3[c]);
        a.add[c] = -mean3[c] / std3[c];
    }
    const size_t lds = (size_t)(vimg::BX * a.taps_x + vimg::BY * a.taps_y) * 4 + (size_t)(vimg::BX + vimg::BY) * 8;
    if (lds > 150 * 1024) return VRWKV_ESHAPE;             // down-sampling ratios beyond ~500: not an image transform any more
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vimg::resize_normalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)((S + vimg::BX - 1) / vimg::BX), (unsigned)((S + vimg::BY - 1) / vimg::BY));
    hipLaunchKernelGGL(vimg::resize_normalize_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
