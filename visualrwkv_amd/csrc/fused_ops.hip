// Streaming (HBM-bound) kernels around the WKV7 operator: fused AdamW on a ZeRO-1 shard, squared-norm
// reduction for gradient clipping.  16-byte accesses, grid-stride, one pass over the data.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

// AdamW (decoupled weight decay, bias-corrected) on a flat shard:
//   DeepSpeed FusedAdam(adam_w_mode=True) as configured by the reference (src/model.py:410):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr * ( (m/bc1) / (sqrt(v/bc2) + eps) + wd p )
// master/m/v fp32, gradient and published parameter bf16.  Elements with global index >= wd_boundary get
// no weight decay (tensors that are < 2-D after squeeze(), src/model.py:391-393).
__global__ __launch_bounds__(256) void adamw_kernel(long n, float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, const uint16_t* __restrict__ grad,
                                                    uint16_t* __restrict__ param, float lr, float b1, float b2, float eps,
                                                    float wd, float inv_bc1, float inv_sqrt_bc2, float grad_scale,
                                                    long global_offset, long wd_boundary, const float* __restrict__ sqnorm,
                                                    float clip) {
    // sqnorm != nullptr: the global squared gradient norm lives on the device (summed over ranks, of the UNSCALED
    // gradient sum); the clip factor min(1, clip / (|g| + 1e-6)) is formed here instead of on the host, so the
    // optimizer step needs no device -> host synchronisation.  grad_scale then is 1 / world.
    if (sqnorm) {
        const float gnorm = sqrtf(*sqnorm) * grad_scale;
        if (clip > 0.f) grad_scale *= fminf(1.f, clip / (gnorm + 1e-6f));
    }
    // one 4-element vector per thread, no loop, non-temporal accesses (nothing here is read again before ~20 GB of other traffic):
    // the grid-stride loop over 2048 workgroups ran at 5.4 TB/s; see relusq_fwd_kernel (tmix_fused.hip) for the measurement
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (n >> 2)) return;
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2_nt __attribute__((ext_vector_type(2)));
    const f32x4_nt p4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(master) + i);
    const f32x4_nt m4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(m) + i);
    const f32x4_nt v4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(v) + i);
    const u32x2_nt g2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(grad) + i);
    float p[4] = {p4[0], p4[1], p4[2], p4[3]}, mm[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]};
    const float g[4] = {bf16_lo(g2[0]) * grad_scale, bf16_hi(g2[0]) * grad_scale, bf16_lo(g2[1]) * grad_scale, bf16_hi(g2[1]) * grad_scale};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float decay = (global_offset + 4 * i + e) < wd_boundary ? wd : 0.f;
        mm[e] = b1 * mm[e] + (1.f - b1) * g[e];
        vv[e] = b2 * vv[e] + (1.f - b2) * g[e] * g[e];
        const float upd = (mm[e] * inv_bc1) / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps) + decay * p[e];
        p[e] -= lr * upd;
    }
    const f32x4_nt po = {p[0], p[1], p[2], p[3]}, mo = {mm[0], mm[1], mm[2], mm[3]}, vo = {vv[0], vv[1], vv[2], vv[3]};
    const u32x2_nt bo = {cvt_pk_bf16(p[0], p[1]), cvt_pk_bf16(p[2], p[3])};
    __builtin_nontemporal_store(po, reinterpret_cast<f32x4_nt*>(master) + i);
    __builtin_nontemporal_store(mo, reinterpret_cast<f32x4_nt*>(m) + i);
    __builtin_nontemporal_store(vo, reinterpret_cast<f32x4_nt*>(v) + i);
    __builtin_nontemporal_store(bo, reinterpret_cast<u32x2_nt*>(param) + i);
}

// out[0] += sum x^2 over a bf16 buffer (n % 8 == 0); one atomic per workgroup
__global__ __launch_bounds__(256) void sqnorm_bf16_kernel(long n, const uint16_t* __restrict__ x, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    const long nvec = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const uint4 u = reinterpret_cast<const uint4*>(x)[i];
        const float f[8] = {bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y), bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(f[e], f[e], acc);
    }
    acc = group_sum<6>(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

int grid_for(long nvec) {
    long b = (nvec + 255) / 256;
    return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);      // <= 8 workgroups per CU, grid-stride the rest
}

}  // namespace

extern "C" {

int vrwkv_adamw_step_bf16(long n, float* master, float* m, float* v, const void* grad, void* param,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                          float grad_scale, long global_offset, long wd_boundary, void* stream) {
    if (n <= 0 || !master || !m || !v || !grad || !param || step < 1) return VRWKV_EINVAL;
    if (n % 4 != 0) return VRWKV_ESHAPE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(((n >> 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, master, m, v,
                       (const uint16_t*)grad, (uint16_t*)param, lr, beta1, beta2, eps, weight_decay,
                       (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, global_offset, wd_boundary, (const float*)nullptr, 0.f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_adamw_step_clip_bf16(long n, float* master, float* m, float* v, const void* grad, void* param,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               const float* sqnorm, float inv_world, float clip, long global_offset, long wd_boundary,
                               void* stream) {
    if (n <= 0 || !master || !m || !v || !grad || !param || !sqnorm || step < 1) return VRWKV_EINVAL;
    if (n % 4 != 0) return VRWKV_ESHAPE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(((n >> 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, master, m, v,
                       (const uint16_t*)grad, (uint16_t*)param, lr, beta1, beta2, eps, weight_decay,
                       (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), inv_world, global_offset, wd_boundary, sqnorm, clip);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_sqnorm_bf16(long n, const void* x, float* out, void* stream) {
    if (n <= 0 || !x || !out) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(sqnorm_bf16_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n, (const uint16_t*)x, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // extern "C"

// ---------------------------------------------------------------- bf16 matrix transpose (weights for the T,N input-gradient GEMMs)
// out[c][r] = in[r][c] for a row-major (rows, cols) bf16 matrix, rows % 64 == cols % 64 == 0.  One workgroup per 64 x 64
// tile: 16-byte coalesced reads of tile rows into LDS (row stride 66 elements: column walks hit distinct banks), 16-byte
// coalesced writes of tile columns.  torch's strided copy reaches 0.7-0.9 TB/s on these 8-33 MB matrices.
namespace {
__global__ __launch_bounds__(256) void transpose_bf16_kernel(int rows, int cols, const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
    __shared__ uint16_t tile[64][66];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, r = q >> 3, cc = (q & 7) * 8;
        const uint4 u = *reinterpret_cast<const uint4*>(in + (r0 + r) * cols + c0 + cc);
        uint32_t* t = reinterpret_cast<uint32_t*>(&tile[r][cc]);      // 4-byte aligned: 66 r + cc is even
        t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, c = q >> 3, rr = (q & 7) * 8;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rr + 2 * e][c] | ((uint32_t)tile[rr + 2 * e + 1][c] << 16);
        *reinterpret_cast<uint4*>(out + (c0 + c) * rows + r0 + rr) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
}  // namespace

extern "C" int vrwkv_transpose_bf16(long rows, long cols, const void* in, void* out, void* stream) {
    if (rows <= 0 || cols <= 0 || !in || !out) return VRWKV_EINVAL;
    if (rows % 64 || cols % 64 || rows > 0x7fffffffL || cols > 0x7fffffffL) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) return VRWKV_EALIGN;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(cols / 64), (unsigned)(rows / 64)), dim3(256), 0, (hipStream_t)stream,
                       (int)rows, (int)cols, (const uint16_t*)in, (uint16_t*)out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
