// Batched GEMV for single-token decode (stateful generation): several independent products
// y_j = act_j(W_j x_j) (+ res_j) in ONE launch.  At decode time every projection of an RWKV-7 layer is a GEMV that
// streams its weight matrix once (HBM-bound: 3 GB of bf16 weights per token for the 1.5B stack) and the step is
// otherwise launch-bound -- the library issues one kernel per projection and activation.
//
// W_j: (N_j, K_j) bf16 row-major (nn.Linear layout; LoRA factors are pre-transposed by the caller), x_j: (B, K_j) bf16,
// y_j: (B, N_j) bf16, B <= 4.  fp32 accumulation, one rounding at the end (the library GEMM it replaces also
// accumulates in fp32).
//   long rows (K >= 512): a wave owns 2 rows and walks them with 16-byte loads, 8 weight loads per lane in flight
//   before the first use; x comes straight from global memory (every wave reads the same few KB: L1/L2 hits), so
//   nothing is staged and no barrier delays the weight stream.  8 rows per workgroup: a 2048 x 2048 projection is
//   256 workgroups, and the whole matrix is in flight at once -- the step is bound by one HBM round trip per launch.
//   short rows (K < 512): a group of G = K/8 lanes per row, 64/G rows at a time, x staged in LDS as fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

constexpr int GV_MAX_JOBS = 8;
constexpr int GV_MAX_B = 4;
constexpr int GV_ROWS_PER_WG = 32;      // short-row jobs
constexpr int GV_LONG_ROWS_PER_WG = 8;  // long-row jobs: 2 per wave
constexpr int GV_LONG_KCHUNKS = 64;     // K >= 512
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

DEVFN void unpack8(const uint4& u, float* f) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}

__global__ __launch_bounds__(GV_THREADS) void gemv_multi_kernel(GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];         // [B][K], short-row jobs only
    int j = 0;
#pragma unroll
    for (int t = 1; t < GV_MAX_JOBS; ++t) j += (t < a.n_jobs && (int)blockIdx.x >= a.job[t].wg_begin) ? 1 : 0;
    const GemvJob job = a.job[j];
    const int K = job.K, N = job.N, B = a.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kchunks = K / 8;
    if (kchunks >= GV_LONG_KCHUNKS) {
        constexpr int R = GV_LONG_ROWS_PER_WG / 4, U = 4;
        const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_LONG_ROWS_PER_WG + wave * R;
        float acc[R][GV_MAX_B];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < GV_MAX_B; ++b) acc[r][b] = 0.f;
        for (int c0 = lane; c0 < kchunks; c0 += 64 * U) {
            uint4 u[R][U];
#pragma unroll
            for (int q = 0; q < U; ++q)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = row0 + r, c = c0 + 64 * q;
                    u[r][q] = (n < N && c < kchunks) ? *reinterpret_cast<const uint4*>(job.W + (size_t)n * K + c * 8) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int c = c0 + 64 * q;
                if (c < kchunks) {
                    float wv[R][8];
#pragma unroll
                    for (int r = 0; r < R; ++r) unpack8(u[r][q], wv[r]);
#pragma unroll
                    for (int b = 0; b < GV_MAX_B; ++b) {
                        if (b < B) {
                            float xv[8];
                            unpack8(*reinterpret_cast<const uint4*>(job.x + (size_t)b * K + c * 8), xv);
#pragma unroll
                            for (int r = 0; r < R; ++r)
#pragma unroll
                                for (int e = 0; e < 8; ++e) acc[r][b] = fmaf(wv[r][e], xv[e], acc[r][b]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = row0 + r;
#pragma unroll
            for (int b = 0; b < GV_MAX_B; ++b) {
                if (b < B) {
                    float v = group_sum<6>(acc[r][b]);
                    if (lane == 0 && n < N) {
                        v = apply_act(v, job.act);
                        if (job.res) v += bf16_to_f32(job.res[(size_t)b * N + n]);
                        job.y[(size_t)b * N + n] = (uint16_t)f32_to_bf16_bits(v);
                    }
                }
            }
        }
    } else {
        // short rows (LoRA up-projections, K = 64..256): a group of G lanes per row, 64/G rows at a time
        for (int i = threadIdx.x * 8; i < B * K; i += GV_THREADS * 8) unpack8(*reinterpret_cast<const uint4*>(job.x + i), xs + i);
        __syncthreads();
        const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_ROWS_PER_WG + wave * (GV_ROWS_PER_WG / 4);   // 8 rows per wave
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
        const bool long_rows = K[j] / 8 >= GV_LONG_KCHUNKS;
        const int rows = long_rows ? GV_LONG_ROWS_PER_WG : GV_ROWS_PER_WG;
        wg += (N[j] + rows - 1) / rows;
        if (!long_rows) kmax = K[j] > kmax ? K[j] : kmax;
    }
    const size_t lds = (size_t)B * kmax * sizeof(float);
    hipLaunchKernelGGL(gemv_multi_kernel, dim3((unsigned)wg), dim3(GV_THREADS), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
