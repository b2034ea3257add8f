// Host launchers + C-ABI of the ViT attention forward (attention_kernels.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <attention_kernels.h>

namespace {

int g_qtiles = 0;      // 0 = default (2)

template <int D, int QT, int S>
int launch(const vattn::Args& a, int B, hipStream_t stream) {
    using G = vattn::Geo<D, QT, S>;
    vattn::Args p = a;
    p.nqb = (a.L + G::NQ - 1) / G::NQ;
    p.BH = B * a.H;
    auto kern = vattn::fwd_kernel<D, QT, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nqb * p.BH)), dim3(256), G::LDS_BYTES, stream, p);
    e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

template <int D, int S>
int launch_qt(const vattn::Args& a, int B, hipStream_t stream) {
    if (g_qtiles == 1) return launch<D, 1, S>(a, B, stream);
    return launch<D, 2, S>(a, B, stream);
}

int check(int B, int L, int H, const void* q, const void* k, const void* v, const void* o, long sb, long sl, long sh) {
    if (B <= 0 || L <= 0 || H <= 0 || !q || !k || !v || !o) return VRWKV_EINVAL;
    if ((sb | sl | sh) % 8 != 0) return VRWKV_EALIGN;       // 16-byte rows
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(o)) & 15u) return VRWKV_EALIGN;
    return VRWKV_OK;
}

}  // namespace

extern "C" int vrwkv_attention_set_qtiles(int qt) {
    if (qt != 0 && qt != 1 && qt != 2) return VRWKV_EINVAL;
    g_qtiles = qt;
    return VRWKV_OK;
}

extern "C" int vrwkv_attention_fwd_bf16(int B, int L, int H, int D, const void* q, const void* k, const void* v,
                                        long stride_b, long stride_l, long stride_h, void* o, void* stream) {
    if (int rc = check(B, L, H, q, k, v, o, stride_b, stride_l, stride_h)) return rc;
    if (D != 64 && D != 72) return VRWKV_ESHAPE;
    vattn::Args p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, stride_b, stride_l, stride_h, L, H,
                  (float)(1.4426950408889634 / sqrt((double)D)), nullptr, nullptr, 0, 0};
    if (D == 64) return launch_qt<64, 0>(p, B, (hipStream_t)stream);
    return launch_qt<72, 0>(p, B, (hipStream_t)stream);
}

extern "C" int vrwkv_attention_relpos_fwd_bf16(int B, int S, int H, int D, const void* q, const void* k, const void* v,
                                               long stride_b, long stride_l, long stride_h, const void* rel_h,
                                               const void* rel_w, void* o, void* stream) {
    const int L = S * S;
    if (int rc = check(B, L, H, q, k, v, o, stride_b, stride_l, stride_h)) return rc;
    if (!rel_h || !rel_w) return VRWKV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(rel_h) | reinterpret_cast<uintptr_t>(rel_w)) & 15u) return VRWKV_EALIGN;
    if (D != 64 || (S != 14 && S != 64)) return VRWKV_ESHAPE;
    vattn::Args p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, stride_b, stride_l, stride_h, L, H,
                  (float)(1.4426950408889634 / sqrt((double)D)), (const uint16_t*)rel_h, (const uint16_t*)rel_w, 0, 0};
    if (S == 14) return launch_qt<64, 14>(p, B, (hipStream_t)stream);
    return launch_qt<64, 64>(p, B, (hipStream_t)stream);
}
