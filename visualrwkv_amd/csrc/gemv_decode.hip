// Batched GEMV for single-token decode (stateful generation): several independent products
// y_j = act_j(W_j x_j) (+ res_j) in ONE launch.  At decode time every projection of an RWKV-7 layer is a GEMV that
// streams its weight matrix once (HBM-bound: 3 GB of bf16 weights per token for the 1.5B stack) and the step is
// otherwise launch-bound -- the library issues one kernel per projection and activation.
//
// W_j: (N_j, K_j) bf16 row-major (nn.Linear layout; LoRA factors are pre-transposed by the caller), x_j: (B, K_j) bf16,
// y_j: (B, N_j) bf16, B <= 4.  A row is reduced by a group of G = min(64, K/8) lanes (16-byte loads, 8 bf16 per lane
// and step), so a wave handles 64/G rows at a time; x is staged once per workgroup in LDS as fp32.  fp32 accumulation,
// one rounding at the end (the library GEMM it replaces also accumulates in fp32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

constexpr int GV_MAX_JOBS = 8;
constexpr int GV_MAX_B = 4;
constexpr int GV_ROWS_PER_WG = 32;
constexpr int GV_THREADS = 256;

struct GemvJob {
    const uint16_t* W;
    const uint16_t* x;
    const uint16_t* res;      // optional (B,N): added after the activation
    uint16_t* y;
    int N, K, act;            // act: 0 none, 1 tanh, 2 sigmoid, 3 relu^2
    int wg_begin;             // first workgroup of this job
};
struct GemvArgs { GemvJob job[GV_MAX_JOBS]; int n_jobs, B; };

DEVFN float apply_act(float v, int act) {
    if (act == 1) return 1.f - 2.f / (1.f + fast_exp(2.f * v));          // tanh
    if (act == 2) return 1.f / (1.f + fast_exp(-v));
    if (act == 3) { const float r = fmaxf(v, 0.f); return r * r; }
    return v;
}

__global__ __launch_bounds__(GV_THREADS) void gemv_multi_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];         // [B][K]
    int j = 0;
#pragma unroll
    for (int t = 1; t < GV_MAX_JOBS; ++t) j += (t < a.n_jobs && (int)blockIdx.x >= a.job[t].wg_begin) ? 1 : 0;
    const GemvJob job = a.job[j];
    const int K = job.K, N = job.N, B = a.B;
    for (int i = threadIdx.x * 8; i < B * K; i += GV_THREADS * 8) {
        const uint4 u = *reinterpret_cast<const uint4*>(job.x + i);
        float* d = xs + i;
        d[0] = bf16_lo(u.x); d[1] = bf16_hi(u.x); d[2] = bf16_lo(u.y); d[3] = bf16_hi(u.y);
        d[4] = bf16_lo(u.z); d[5] = bf16_hi(u.z); d[6] = bf16_lo(u.w); d[7] = bf16_hi(u.w);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kchunks = K / 8;
    const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_ROWS_PER_WG + wave * (GV_ROWS_PER_WG / 4);   // 8 rows per wave
    if (kchunks >= 64) {
        // long rows: the whole wave walks a row; 4 rows at a time keep four 16-byte loads per lane in flight
        for (int rb = 0; rb < GV_ROWS_PER_WG / 4; rb += 4) {
            float acc[4][GV_MAX_B];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < GV_MAX_B; ++b) acc[i][b] = 0.f;
            for (int c = lane; c < kchunks; c += 64) {
                uint4 u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = row0 + rb + i;
                    u[i] = n < N ? *reinterpret_cast<const uint4*>(job.W + (size_t)n * K + c * 8) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float wv[8] = {bf16_lo(u[i].x), bf16_hi(u[i].x), bf16_lo(u[i].y), bf16_hi(u[i].y),
                                         bf16_lo(u[i].z), bf16_hi(u[i].z), bf16_lo(u[i].w), bf16_hi(u[i].w)};
#pragma unroll
                    for (int b = 0; b < GV_MAX_B; ++b) {
                        if (b < B) {
                            const float4 x0 = *reinterpret_cast<const float4*>(xs + b * K + c * 8);
                            const float4 x1 = *reinterpret_cast<const float4*>(xs + b * K + c * 8 + 4);
                            acc[i][b] += wv[0] * x0.x + wv[1] * x0.y + wv[2] * x0.z + wv[3] * x0.w
                                       + wv[4] * x1.x + wv[5] * x1.y + wv[6] * x1.z + wv[7] * x1.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = row0 + rb + i;
#pragma unroll
                for (int b = 0; b < GV_MAX_B; ++b) {
                    if (b < B) {
                        float v = group_sum<6>(acc[i][b]);
                        if (lane == 0 && n < N) {
                            v = apply_act(v, job.act);
                            if (job.res) v += bf16_to_f32(job.res[(size_t)b * N + n]);
                            job.y[(size_t)b * N + n] = (uint16_t)f32_to_bf16_bits(v);
                        }
                    }
                }
            }
        }
    } else {
        // short rows (LoRA up-projections, K = 64..256): a group of G lanes per row, 64/G rows at a time
        int G = 32;
        while (G > kchunks) G >>= 1;
        const int rows_at_once = 64 / G, sub = lane / G, gl = lane % G;
        for (int rb = 0; rb < GV_ROWS_PER_WG / 4; rb += rows_at_once) {
            const int n = row0 + rb + sub;
            const bool live = sub + rb < GV_ROWS_PER_WG / 4 && n < N;
            float acc[GV_MAX_B] = {0.f, 0.f, 0.f, 0.f};
            for (int c = gl; c < kchunks; c += G) {
                const uint4 u = live ? *reinterpret_cast<const uint4*>(job.W + (size_t)n * K + c * 8) : make_uint4(0, 0, 0, 0);
                const float wv[8] = {bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y), bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w)};
#pragma unroll
                for (int b = 0; b < GV_MAX_B; ++b) {
                    if (b < B) {
                        const float* xb = xs + b * K + c * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[b] = fmaf(wv[e], xb[e], acc[b]);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < GV_MAX_B; ++b) {
                if (b < B) {
                    float v = acc[b];                    // all-reduce inside the G-lane group (G is uniform)
                    v = G >= 32 ? group_sum<5>(v) : G >= 16 ? group_sum<4>(v) : G >= 8 ? group_sum<3>(v)
                      : G >= 4 ? group_sum<2>(v) : G >= 2 ? group_sum<1>(v) : v;
                    if (gl == 0 && live) {
                        v = apply_act(v, job.act);
                        if (job.res) v += bf16_to_f32(job.res[(size_t)b * N + n]);
                        job.y[(size_t)b * N + n] = (uint16_t)f32_to_bf16_bits(v);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int vrwkv_gemv_multi_bf16(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res,
                                     void* const* y, const int* N, const int* K, const int* act, void* stream) {
    if (n_jobs <= 0 || n_jobs > GV_MAX_JOBS || B <= 0 || B > GV_MAX_B || !W || !x || !y || !N || !K || !act) return VRWKV_EINVAL;
    GemvArgs a{};
    a.n_jobs = n_jobs; a.B = B;
    int wg = 0, kmax = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!W[j] || !x[j] || !y[j] || N[j] <= 0 || K[j] <= 0 || K[j] % 8 != 0) return VRWKV_ESHAPE;
        if ((reinterpret_cast<uintptr_t>(W[j]) | reinterpret_cast<uintptr_t>(x[j])) & 15u) return VRWKV_EALIGN;
        a.job[j] = GemvJob{(const uint16_t*)W[j], (const uint16_t*)x[j], res ? (const uint16_t*)res[j] : nullptr, (uint16_t*)y[j],
                           N[j], K[j], act[j], wg};
        wg += (N[j] + GV_ROWS_PER_WG - 1) / GV_ROWS_PER_WG;
        kmax = K[j] > kmax ? K[j] : kmax;
    }
    const size_t lds = (size_t)B * kmax * sizeof(float);
    if (lds > 64 * 1024) return VRWKV_ESHAPE;                          // B*K <= 16384 (e.g. B = 2 at K = 8192)
    hipLaunchKernelGGL(gemv_multi_kernel, dim3((unsigned)wg), dim3(GV_THREADS), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
