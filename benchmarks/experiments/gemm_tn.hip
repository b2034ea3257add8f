// C-ABI of the T,N GEMM with an element-wise epilogue (gemm_tn.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"      // error codes only: the entry point below is not declared there
#include <gemm_tn.h>

extern "C" int vrwkv_gemm_tn_bf16(long M, int N, int K, const void* A, const void* B, void* C, int epilogue, const void* aux, void* stream) {
    if (M <= 0 || !A || !B || !C) return VRWKV_EINVAL;
    if (epilogue < 0 || epilogue > 2 || (epilogue == gtn::EPI_DRELUSQ && !aux)) return VRWKV_EINVAL;
    if (M % gtn::TM != 0 || N <= 0 || N % gtn::TN != 0 || K <= 0 || K % gtn::KT != 0 || (long)K * 2 * 256 >= (1L << 31)) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(aux)) & 15u) return VRWKV_EALIGN;
    const gtn::Args a{M, N, K, (const uint16_t*)A, (const uint16_t*)B, (uint16_t*)C, (const uint16_t*)aux};
    const int Tn = N / gtn::TN, nper = (Tn + 7) / 8;
    const dim3 grid((unsigned)(8L * nper * (M / gtn::TM)));
    void (*kern)(gtn::Args) = epilogue == gtn::EPI_RELUSQ ? &gtn::gemm_tn_kernel<gtn::EPI_RELUSQ>
                              : epilogue == gtn::EPI_DRELUSQ ? &gtn::gemm_tn_kernel<gtn::EPI_DRELUSQ> : &gtn::gemm_tn_kernel<gtn::EPI_NONE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, gtn::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, dim3(512), gtn::LDS_BYTES, (hipStream_t)stream, a);
    e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
