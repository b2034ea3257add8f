// C-ABI of the skinny weight-gradient product (lora_wgrad.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <lora_wgrad.h>

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
