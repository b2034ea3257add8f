// Batched GEMV for single-token decode (stateful generation): several independent products
// y_j = act_j(W_j x_j) (+ res_j) in ONE launch.  At decode time every projection of an RWKV-7 layer is a GEMV that
// streams its weight matrix once (HBM-bound: 3 GB of bf16 weights per token for the 1.5B stack) and the step is
// otherwise launch-bound -- the library issues one kernel per projection and activation, and a captured kernel costs
// ~5 us whatever it does.
//
// W_j: (N_j, K_j) bf16 row-major (nn.Linear layout; LoRA factors are pre-transposed by the caller), x_j: (B, K_j) bf16,
// y_j: (B, N_j) bf16, B <= 4.  fp32 accumulation, one rounding at the end (the library GEMM it replaces also
// accumulates in fp32).
//   long rows (K >= 512): a wave owns 2 rows and walks them with 16-byte loads, 8 weight loads per lane in flight
//   before the first use; x comes straight from global memory (every wave reads the same few KB: L1/L2 hits), so
//   nothing is staged and no barrier delays the weight stream.  8 rows per workgroup: a 2048 x 2048 projection is
//   256 workgroups, and the whole matrix is in flight at once -- the step is bound by one HBM round trip per launch.
//   short rows (K < 512): a group of G = K/8 lanes per row, 64/G rows at a time, x staged in LDS as fp32.
//
// LayerNorm fold (vrwkv_gemv_ln_multi_bf16, long rows, K <= 4096): the jobs share one raw residual row x; every
// workgroup rebuilds its job's input  in_j = h + (x_prev - h) mu_j,  h = LayerNorm(x)  (Block.forward's ln1 / ln2 + the
// token-shift lerps of src/model.py:169-173,222-223,250,253) from a few KB that sit in L2 -- cheaper than the ~7 us a
// separate captured kernel costs.  One workgroup also writes h to `h_out`; the carried row x_prev itself is replaced by
// a later launch (`copy` side job of vrwkv_gemv_multi_copy_bf16 / vrwkv_decode_tmix_head_bf16), because other
// workgroups of this launch still read it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>
// 16-byte register pieces as the compiler's NATIVE vector type: private arrays of HIP's uint4 (a struct of unions) are not
// promoted to registers and ended up in scratch memory (32 B/lane in the LayerNorm prologues of this file).
typedef uint32_t vrwkv_u4 __attribute__((ext_vector_type(4)));
#define uint4 vrwkv_u4
#define make_uint4(a_, b_, c_, d_) (vrwkv_u4{(uint32_t)(a_), (uint32_t)(b_), (uint32_t)(c_), (uint32_t)(d_)})

namespace {

constexpr int GV_MAX_JOBS = 8;
constexpr int GV_MAX_B = 4;
constexpr int GV_ROWS_PER_WG = 32;      // short-row jobs
constexpr int GV_LONG_ROWS_PER_WG = 8;  // long-row jobs: 2 per wave
constexpr int GV_LONG_KCHUNKS = 64;     // K >= 512
constexpr int GV_LN_MAX_K = 4096;
constexpr int GV_THREADS = 256;

struct GemvJob {
    const uint16_t* W;
    const uint16_t* x;
    const uint16_t* res;      // optional (B,N): added after the activation
    const uint16_t* mu;       // LayerNorm fold: this job's lerp coefficients (K)
    uint16_t* y;
    int N, K, act;            // act: 0 none, 1 tanh, 2 sigmoid, 3 relu^2
    int wg_begin;             // first workgroup of this job
};
struct LnFold {
    const uint16_t *ln_w, *ln_b, *x_prev;   // (K), (K), (B,K); null ln_w = no fold
    uint16_t* h_out;                        // (B,K): LayerNorm(x), written by the first workgroup
    float eps;
};
struct GemvArgs {
    GemvJob job[GV_MAX_JOBS];
    int n_jobs, B;
    LnFold ln;
    const uint4* copy_src; uint4* copy_dst; long copy_vec;      // side job: 16-byte vectors copied by workgroup 0
};

DEVFN float rb(float x) { return __uint_as_float(f32_to_bf16_bits(x) << 16); }     // round to bf16, keep as fp32

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
DEVFN uint4 pack8(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

constexpr int GV_R = GV_LONG_ROWS_PER_WG / 4, GV_U = 4;      // rows per wave, 16-byte loads per row and lane in flight

// Rows longer than 2048 (the 4C -> C projection of the channel mix) are split over the workgroup's waves instead:
// 2 rows per workgroup, wave w walks quarter w of both, partial sums meet in LDS.  Four times the workgroups, and
// again the whole matrix is in flight at once instead of four dependent batches per wave.
__host__ __device__ inline bool split_k(int K) { return K > 2048 && K % 32 == 0; }
__host__ __device__ inline int long_rows_per_wg(int K) { return split_k(K) ? GV_LONG_ROWS_PER_WG / 4 : GV_LONG_ROWS_PER_WG; }

DEVFN void load_batch(const GemvJob& job, int row0, int c0, int kchunks, uint4 (&u)[GV_R][GV_U]) {
#pragma unroll
    for (int q = 0; q < GV_U; ++q)
#pragma unroll
        for (int r = 0; r < GV_R; ++r) {
            const int n = row0 + r, c = c0 + 64 * q;
            u[r][q] = (n < job.N && c < kchunks) ? *reinterpret_cast<const uint4*>(job.W + (size_t)n * job.K + c * 8) : make_uint4(0, 0, 0, 0);
        }
}

// One long-row wave: rows row0, row0+1 against B input vectors; `input(b, c, xv)` yields chunk c of vector b as 8 floats.
// `first` holds the weights of the first batch (c0 = lane), loaded by the caller before whatever it had to wait for.
// The walk covers chunks [c_begin, kchunks); acc holds per-lane partial sums.
template <int BB, typename F>
DEVFN void long_rows_acc(const GemvJob& job, int B, int row0, int lane, int c_begin, int kchunks, uint4 (&first)[GV_R][GV_U],
                         float (&acc)[GV_R][BB], F&& input) {
    constexpr int R = GV_R, U = GV_U;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < BB; ++b) acc[r][b] = 0.f;
    for (int c0 = c_begin + lane; c0 < kchunks; c0 += 64 * U) {
        uint4 u[R][U];
        if (c0 == c_begin + lane) {
#pragma unroll
            for (int q = 0; q < U; ++q)
#pragma unroll
                for (int r = 0; r < R; ++r) u[r][q] = first[r][q];
        } else {
            load_batch(job, row0, c0, kchunks, u);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int c = c0 + 64 * q;
            if (c < kchunks) {
                float wv[R][8];
#pragma unroll
                for (int r = 0; r < R; ++r) unpack8(u[r][q], wv[r]);
#pragma unroll
                for (int b = 0; b < BB; ++b) {
                    if (b < B) {
                        float xv[8];
                        input(b, c, xv);
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[r][b] = fmaf(wv[r][e], xv[e], acc[r][b]);
                    }
                }
            }
        }
    }
}

template <int BB, typename F>
DEVFN void long_rows(const GemvJob& job, int B, int row0, int lane, uint4 (&first)[GV_R][GV_U], F&& input) {
    constexpr int R = GV_R;
    const int N = job.N;
    float acc[R][BB];
    long_rows_acc<BB>(job, B, row0, lane, 0, job.K / 8, first, acc, input);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = row0 + r;
#pragma unroll
        for (int b = 0; b < BB; ++b) {
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
}

DEVFN int find_job(const GemvArgs& a) {
    int j = 0;
#pragma unroll
    for (int t = 1; t < GV_MAX_JOBS; ++t) j += (t < a.n_jobs && (int)blockIdx.x >= a.job[t].wg_begin) ? 1 : 0;
    return j;
}

__global__ __launch_bounds__(GV_THREADS) void gemv_multi_kernel(GemvArgs a) {
    float* xs = reinterpret_cast<float*>(dyn_lds());                    // [B][K], short-row jobs only
    const GemvJob job = a.job[find_job(a)];
    const int K = job.K, N = job.N, B = a.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kchunks = K / 8;
    if (a.copy_vec && blockIdx.x == 0)
        for (long i = threadIdx.x; i < a.copy_vec; i += GV_THREADS) a.copy_dst[i] = a.copy_src[i];
    if (split_k(K)) {
        __shared__ float part[4][GV_R][GV_MAX_B];
        const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_R, kq = kchunks / 4;
        uint4 first[GV_R][GV_U];
        load_batch(job, row0, wave * kq + lane, (wave + 1) * kq, first);
        float acc[GV_R][GV_MAX_B];
        long_rows_acc<GV_MAX_B>(job, B, row0, lane, wave * kq, (wave + 1) * kq, first, acc, [&](int b, int c, float* xv) {
            unpack8(*reinterpret_cast<const uint4*>(job.x + (size_t)b * K + c * 8), xv);
        });
#pragma unroll
        for (int r = 0; r < GV_R; ++r)
#pragma unroll
            for (int b = 0; b < GV_MAX_B; ++b) {
                const float v = b < B ? group_sum<6>(acc[r][b]) : 0.f;
                if (lane == 0) part[wave][r][b] = v;
            }
        __syncthreads();
        if (threadIdx.x < GV_R * GV_MAX_B) {
            const int r = threadIdx.x / GV_MAX_B, b = threadIdx.x % GV_MAX_B, n = row0 + r;
            if (b < B && n < N) {
                float v = apply_act(part[0][r][b] + part[1][r][b] + part[2][r][b] + part[3][r][b], job.act);
                if (job.res) v += bf16_to_f32(job.res[(size_t)b * N + n]);
                job.y[(size_t)b * N + n] = (uint16_t)f32_to_bf16_bits(v);
            }
        }
    } else if (kchunks >= GV_LONG_KCHUNKS) {
        const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_LONG_ROWS_PER_WG + wave * GV_R;
        uint4 first[GV_R][GV_U];
        load_batch(job, row0, lane, kchunks, first);
        long_rows<GV_MAX_B>(job, B, row0, lane, first, [&](int b, int c, float* xv) {
            unpack8(*reinterpret_cast<const uint4*>(job.x + (size_t)b * K + c * 8), xv);
        });
    } else {
        // short rows (LoRA up-projections, K = 64..256): a group of G lanes per row, 64/G rows at a time
        for (int i = threadIdx.x * 8; i < B * K; i += GV_THREADS * 8) unpack8(*reinterpret_cast<const uint4*>(job.x + i), xs + i);
        __syncthreads();
        const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_ROWS_PER_WG + wave * (GV_ROWS_PER_WG / 4);   // 8 rows per wave
        int G = 32;
        while (G > kchunks) G >>= 1;
        const int rows_at_once = 64 / G, sub = lane / G, gl = lane % G;
        for (int rb_ = 0; rb_ < GV_ROWS_PER_WG / 4; rb_ += rows_at_once) {
            const int n = row0 + rb_ + sub;
            const bool live = sub + rb_ < GV_ROWS_PER_WG / 4 && n < N;
            float acc[GV_MAX_B] = {0.f, 0.f, 0.f, 0.f};
            for (int c = gl; c < kchunks; c += G) {
                float wv[8];
                unpack8(live ? *reinterpret_cast<const uint4*>(job.W + (size_t)n * K + c * 8) : make_uint4(0, 0, 0, 0), wv);
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

// LayerNorm fold.  The workgroup builds its job's input vectors once, cooperatively: thread t owns chunks t, t+256 of the
// row (K <= 4096), everything it needs is loaded in one go next to nothing else, the row statistics are one LDS exchange,
// and the bf16 inputs are parked in LDS for the four waves' row walks.  Sums are taken around the row's first element so
// that E[d^2] - E[d]^2 does not cancel when the row has a large mean.
template <int CH, int BB>      // chunks per thread (K <= 2048 CH), batch rows compiled in: registers follow the real shape
__global__ __launch_bounds__(GV_THREADS) void gemv_ln_kernel(GemvArgs a) {
    uint4* xin = reinterpret_cast<uint4*>(dyn_lds());                   // [B][K/8] packed bf16 inputs
    __shared__ float red[4][BB][2];
    const GemvJob job = a.job[find_job(a)];
    const int K = job.K, B = a.B, kchunks = K / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = ((int)blockIdx.x - job.wg_begin) * GV_LONG_ROWS_PER_WG + wave * GV_R;
    uint4 first[GV_R][GV_U];
    load_batch(job, row0, lane, kchunks, first);         // the weight stream starts before the prologue's round trip
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 xr[CH][BB], pr[CH][BB], lwr[CH], lbr[CH], mur[CH];
    float x0[BB];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int ch = threadIdx.x + i * GV_THREADS;
        const bool live = ch < kchunks;
        lwr[i] = live ? *reinterpret_cast<const uint4*>(a.ln.ln_w + ch * 8) : zero;
        lbr[i] = live ? *reinterpret_cast<const uint4*>(a.ln.ln_b + ch * 8) : zero;
        mur[i] = live ? *reinterpret_cast<const uint4*>(job.mu + ch * 8) : zero;
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            const bool lb_ = live && b < B;
            xr[i][b] = lb_ ? *reinterpret_cast<const uint4*>(job.x + (size_t)b * K + ch * 8) : zero;
            pr[i][b] = lb_ ? *reinterpret_cast<const uint4*>(a.ln.x_prev + (size_t)b * K + ch * 8) : zero;
        }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b) x0[b] = b < B ? bf16_to_f32(job.x[(size_t)b * K]) : 0.f;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        if (b < B) {
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (threadIdx.x + i * GV_THREADS < kchunks) {
                    float xv[8];
                    unpack8(xr[i][b], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = xv[e] - x0[b]; s += d; s2 = fmaf(d, d, s2); }
                }
            }
            s = group_sum<6>(s); s2 = group_sum<6>(s2);
            if (lane == 0) { red[wave][b][0] = s; red[wave][b][1] = s2; }
        }
    }
    __syncthreads();
    const bool writes_h = blockIdx.x == 0;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        if (b < B) {
            const float s = red[0][b][0] + red[1][b][0] + red[2][b][0] + red[3][b][0];
            const float s2 = red[0][b][1] + red[1][b][1] + red[2][b][1] + red[3][b][1];
            const float m = s / (float)K, mean = x0[b] + m;
            const float rstd = 1.f / sqrtf(fmaxf(s2 / (float)K - m * m, 0.f) + a.ln.eps);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ch = threadIdx.x + i * GV_THREADS;
                if (ch < kchunks) {
                    float xv[8], xp[8], lw[8], lb[8], mu[8], h[8];
                    unpack8(xr[i][b], xv); unpack8(pr[i][b], xp); unpack8(lwr[i], lw); unpack8(lbr[i], lb); unpack8(mur[i], mu);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = rb((xv[e] - mean) * rstd * lw[e] + lb[e]);
                        xv[e] = fmaf(xp[e] - h[e], mu[e], h[e]);
                    }
                    xin[b * kchunks + ch] = pack8(xv);
                    if (writes_h) *reinterpret_cast<uint4*>(a.ln.h_out + (size_t)b * K + ch * 8) = pack8(h);
                }
            }
        }
    }
    __syncthreads();
    long_rows<BB>(job, B, row0, lane, first, [&](int b, int c, float* xv) { unpack8(xin[b * kchunks + c], xv); });
}

int launch(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res, const void* const* mu,
           void* const* y, const int* N, const int* K, const int* act, const LnFold* ln, const void* copy_src,
           void* copy_dst, long copy_elems, void* stream) {
    if (n_jobs <= 0 || n_jobs > GV_MAX_JOBS || B <= 0 || B > GV_MAX_B || !W || !x || !y || !N || !K || !act) return VRWKV_EINVAL;
    if (copy_elems < 0 || (copy_elems > 0 && (!copy_src || !copy_dst))) return VRWKV_EINVAL;
    if (copy_elems % 8 != 0) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(copy_src) | reinterpret_cast<uintptr_t>(copy_dst)) & 15u) return VRWKV_EALIGN;
    GemvArgs a{};
    a.n_jobs = n_jobs; a.B = B;
    a.copy_src = (const uint4*)copy_src; a.copy_dst = (uint4*)copy_dst; a.copy_vec = copy_elems / 8;
    if (ln) a.ln = *ln;
    int wg = 0, kmax = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!W[j] || !x[j] || !y[j] || N[j] <= 0 || K[j] <= 0 || K[j] % 8 != 0) return VRWKV_ESHAPE;
        if ((reinterpret_cast<uintptr_t>(W[j]) | reinterpret_cast<uintptr_t>(x[j])) & 15u) return VRWKV_EALIGN;
        const bool long_rows = K[j] / 8 >= GV_LONG_KCHUNKS;
        if (ln) {
            if (!long_rows || K[j] > GV_LN_MAX_K || K[j] != K[0] || x[j] != x[0]) return VRWKV_ESHAPE;
            if (!mu || !mu[j]) return VRWKV_EINVAL;
            if (reinterpret_cast<uintptr_t>(mu[j]) & 15u) return VRWKV_EALIGN;
        }
        a.job[j] = GemvJob{(const uint16_t*)W[j], (const uint16_t*)x[j], res ? (const uint16_t*)res[j] : nullptr,
                           ln ? (const uint16_t*)mu[j] : nullptr, (uint16_t*)y[j], N[j], K[j], act[j], wg};
        const int rows = ln ? GV_LONG_ROWS_PER_WG : long_rows ? long_rows_per_wg(K[j]) : GV_ROWS_PER_WG;
        wg += (N[j] + rows - 1) / rows;
        if (!long_rows) kmax = K[j] > kmax ? K[j] : kmax;
    }
    const size_t lds = (size_t)B * kmax * sizeof(float);
    if (ln) {
        const dim3 g((unsigned)wg), t(GV_THREADS);
        const size_t l = (size_t)B * K[0] * 2;
        const hipStream_t st = (hipStream_t)stream;
        const bool wide = K[0] > 8 * GV_THREADS;             // two chunks per thread
        if (B == 1) { if (wide) hipLaunchKernelGGL((gemv_ln_kernel<2, 1>), g, t, l, st, a); else hipLaunchKernelGGL((gemv_ln_kernel<1, 1>), g, t, l, st, a); }
        else if (B == 2) { if (wide) hipLaunchKernelGGL((gemv_ln_kernel<2, 2>), g, t, l, st, a); else hipLaunchKernelGGL((gemv_ln_kernel<1, 2>), g, t, l, st, a); }
        else { if (wide) hipLaunchKernelGGL((gemv_ln_kernel<2, 4>), g, t, l, st, a); else hipLaunchKernelGGL((gemv_ln_kernel<1, 4>), g, t, l, st, a); }
    }
    else hipLaunchKernelGGL(gemv_multi_kernel, dim3((unsigned)wg), dim3(GV_THREADS), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // namespace

extern "C" int vrwkv_gemv_multi_bf16(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res,
                                     void* const* y, const int* N, const int* K, const int* act, void* stream) {
    return launch(n_jobs, B, W, x, res, nullptr, y, N, K, act, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int vrwkv_gemv_multi_copy_bf16(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res,
                                          void* const* y, const int* N, const int* K, const int* act, const void* copy_src,
                                          void* copy_dst, long copy_elems, void* stream) {
    return launch(n_jobs, B, W, x, res, nullptr, y, N, K, act, nullptr, copy_src, copy_dst, copy_elems, stream);
}

extern "C" int vrwkv_gemv_ln_multi_bf16(int n_jobs, int B, int K, const void* const* W, const void* x, const void* ln_w,
                                        const void* ln_b, float eps, const void* x_prev, const void* const* mu, void* h_out,
                                        void* const* y, const int* N, const int* act, void* stream) {
    if (n_jobs <= 0 || n_jobs > GV_MAX_JOBS || !x || !ln_w || !ln_b || !x_prev || !mu || !h_out) return VRWKV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ln_w) | reinterpret_cast<uintptr_t>(ln_b) | reinterpret_cast<uintptr_t>(x_prev) |
         reinterpret_cast<uintptr_t>(h_out)) & 15u) return VRWKV_EALIGN;
    const void* xs_[GV_MAX_JOBS];
    int Ks[GV_MAX_JOBS];
    for (int j = 0; j < n_jobs; ++j) { xs_[j] = x; Ks[j] = K; }
    const LnFold ln{(const uint16_t*)ln_w, (const uint16_t*)ln_b, (const uint16_t*)x_prev, (uint16_t*)h_out, eps};
    return launch(n_jobs, B, W, xs_, nullptr, mu, y, N, Ks, act, &ln, nullptr, nullptr, 0, stream);
}
