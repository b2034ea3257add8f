// WKV7 single-token step with carried state (stateful generation).
//
// The reference has no stateful RWKV-7 path: VisualRWKV.generate re-runs the full training forward for every new
// token (VisualRWKV-v7/v7.00/src/model.py:513-529).  Same recurrence as forward_kernel (cuda/wkv7_cuda.cu:17-42)
// for T = 1, with the per-head state S (fp32, [i][j], i = value row, j = key column) read and written in place:
//     sa = S z ;  S = S diag(w) + sa a^T + v k^T ;  y = S q          w = exp(-exp(w_raw))
// One wave per (b,h): lane i owns row i.  The 16 KB state tile is staged through LDS with 16-byte coalesced
// accesses (row stride padded to 65 floats so that the per-lane row walk is bank-conflict free).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

__global__ __launch_bounds__(64) void wkv7_step_kernel(int H, const uint16_t* __restrict__ w_, const uint16_t* __restrict__ q_,
                                                       const uint16_t* __restrict__ k_, const uint16_t* __restrict__ v_,
                                                       const uint16_t* __restrict__ z_, const uint16_t* __restrict__ a_,
                                                       float* __restrict__ state, uint16_t* __restrict__ y_) {
    __shared__ float S[64][65];
    __shared__ float vec[5][64];                   // w q k z a
    const int bh = blockIdx.x, i = threadIdx.x;
    const size_t vb = (size_t)bh * 64;
    float* sp = state + (size_t)bh * 64 * 64;
    for (int e = i; e < 1024; e += 64) {           // 1024 float4 = the 64x64 tile
        const float4 x = reinterpret_cast<const float4*>(sp)[e];
        const int r = e >> 4, c = (e & 15) * 4;
        S[r][c] = x.x; S[r][c + 1] = x.y; S[r][c + 2] = x.z; S[r][c + 3] = x.w;
    }
    vec[0][i] = fast_exp(-fast_exp(bf16_to_f32(w_[vb + i])));
    vec[1][i] = bf16_to_f32(q_[vb + i]);
    vec[2][i] = bf16_to_f32(k_[vb + i]);
    vec[3][i] = bf16_to_f32(z_[vb + i]);
    vec[4][i] = bf16_to_f32(a_[vb + i]);
    const float vi = bf16_to_f32(v_[vb + i]);
    __syncthreads();
    float sa = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) sa = fmaf(S[i][j], vec[3][j], sa);
    float y = 0.f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const float s = fmaf(S[i][j], vec[0][j], fmaf(sa, vec[4][j], vec[2][j] * vi));
        S[i][j] = s;
        y = fmaf(s, vec[1][j], y);
    }
    y_[vb + i] = (uint16_t)f32_to_bf16_bits(y);
    __syncthreads();
    for (int e = i; e < 1024; e += 64) {
        const int r = e >> 4, c = (e & 15) * 4;
        reinterpret_cast<float4*>(sp)[e] = make_float4(S[r][c], S[r][c + 1], S[r][c + 2], S[r][c + 3]);
    }
}

}  // namespace

extern "C" int vrwkv_wkv7_step_bf16(int B, int H, const void* w, const void* q, const void* k, const void* v,
                                    const void* z, const void* a, float* state, void* y, void* stream) {
    if (B <= 0 || H <= 0 || !w || !q || !k || !v || !z || !a || !state || !y) return VRWKV_EINVAL;
    if (reinterpret_cast<uintptr_t>(state) & 15u) return VRWKV_EALIGN;
    hipLaunchKernelGGL(wkv7_step_kernel, dim3((unsigned)(B * H)), dim3(64), 0, (hipStream_t)stream, H,
                       (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                       (const uint16_t*)z, (const uint16_t*)a, state, (uint16_t*)y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
