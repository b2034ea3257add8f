// C-ABI of the skinny weight-gradient product (lora_wgrad.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <lora_wgrad.h>
#include <wgrad_big.h>

namespace {

// M-slices: ~2 workgroups per CU of the 256; more slices cost more partial-sum traffic than they hide latency
int splits(long M, int Nw, int D) {
    const long nsteps = (M + lwg::KS - 1) / lwg::KS;
    long s = (D <= 160 ? 512 : 256) / (Nw / lwg::CT);     // measured at Nw = 2048: 32 slices (16 for D = 256, which runs
                                                          // two column groups per slice) beat 24, 48 and 64
    if (s < 1) s = 1;
    return (int)(s < nsteps ? s : nsteps);
}
bool supported_d(int D) { return D == 32 || D == 64 || D == 96 || D == 128 || D == 160 || D == 256; }

template <int ND>
int launch(const lwg::Args& a, int S, hipStream_t st) {
    constexpr int D = 16 * ND;                           // Narrow columns per workgroup; a.D / D column groups
    const size_t lds = 2 * lwg::KS * (size_t)(lwg::WS + D + 8) * sizeof(uint16_t);
    hipLaunchKernelGGL((lwg::wgrad_kernel<ND>), dim3((unsigned)(a.Nw / lwg::CT * (a.D / D)), (unsigned)S), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" long vrwkv_wgrad_skinny_ws_floats(long M, int Nw, int D) {
    if (M <= 0 || Nw <= 0 || Nw % lwg::CT != 0 || !supported_d(D)) return -1;
    return (long)splits(M, Nw, D) * Nw * D;
}

extern "C" int vrwkv_wgrad_skinny_bf16(long M, int Nw, int D, const void* wide, const void* narrow, void* out, int transposed,
                                       float* ws, void* stream) {
    if (M <= 0 || !wide || !narrow || !out || !ws) return VRWKV_EINVAL;
    if (Nw <= 0 || Nw % lwg::CT != 0 || !supported_d(D)) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(wide) | reinterpret_cast<uintptr_t>(narrow) | reinterpret_cast<uintptr_t>(ws)) & 15u) return VRWKV_EALIGN;
    const hipStream_t st = (hipStream_t)stream;
    const int S = splits(M, Nw, D);
    const lwg::Args a{M, Nw, D, (const uint16_t*)wide, (const uint16_t*)narrow, ws};
    int e = 0;
    switch (D / 16) {
        case 2: e = launch<2>(a, S, st); break;
        case 4: e = launch<4>(a, S, st); break;
        case 6: e = launch<6>(a, S, st); break;
        case 8: e = launch<8>(a, S, st); break;
        case 10: e = launch<10>(a, S, st); break;
        default: e = launch<8>(a, S, st); break;      // D = 256: two column groups of 128
    }
    if (e) return e;
    const long n = (long)Nw * D;
    hipLaunchKernelGGL(lwg::reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, ws, S, Nw, D, transposed, (uint16_t*)out);
    hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? VRWKV_OK : (int)e2;
}

// ---- the square / wide weight gradients (wgrad_big.h): C (N1 x N2) = A^T B, A (M x N1), B (M x N2)
namespace {
// M-slices so that the grid has at least one workgroup per CU (256): the C x C shapes have 64 tiles -> 4 slices
int big_splits(long M, int N1, int N2) {
    const long tiles = (long)(N1 / wgb::TM) * (N2 / wgb::TN);
    long s = (256 + tiles - 1) / tiles;
    const long nst = M / wgb::KT;
    if (s > 8) s = 8;
    return (int)(s < 1 ? 1 : (s < nst ? s : nst));
}
}  // namespace

extern "C" long vrwkv_wgrad_big_ws_floats(long M, int N1, int N2) {
    if (M <= 0 || M % wgb::KT != 0 || N1 <= 0 || N2 <= 0 || N1 % wgb::TM != 0 || N2 % wgb::TN != 0) return -1;
    const int S = big_splits(M, N1, N2);
    return S > 1 ? (long)S * N1 * N2 : 0;
}

extern "C" int vrwkv_wgrad_big_bf16(long M, int N1, int N2, const void* A, const void* B, void* out, float* ws, void* stream) {
    if (M <= 0 || !A || !B || !out) return VRWKV_EINVAL;
    if (M % wgb::KT != 0 || N1 <= 0 || N2 <= 0 || N1 % wgb::TM != 0 || N2 % wgb::TN != 0) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ws)) & 15u) return VRWKV_EALIGN;
    const int S = big_splits(M, N1, N2);
    if (S > 1 && !ws) return VRWKV_EINVAL;
    const hipStream_t st = (hipStream_t)stream;
    const wgb::Args a{M, N1, N2, S, (const uint16_t*)A, (const uint16_t*)B, ws, (uint16_t*)out};
    const size_t lds = (size_t)wgb::STAGES * 2 * wgb::OPB;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgb::wgrad_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wgb::wgrad_big_kernel, dim3((unsigned)((N1 / wgb::TM) * (N2 / wgb::TN) * S)), dim3(512), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (S > 1) {
        const long n = (long)N1 * N2;
        hipLaunchKernelGGL(wgb::wgrad_big_reduce, dim3((unsigned)((n / 4 + 255) / 256 > 8192 ? 8192 : (n / 4 + 255) / 256)), dim3(256), 0, st, ws, S, n, (uint16_t*)out);
        e = hipGetLastError();
    }
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
