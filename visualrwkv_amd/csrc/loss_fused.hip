// Cross-entropy over the 65536-token vocabulary + the reference's L2Wrap gradient term, one pass over the logits in
// each direction (VisualRWKV-v7/v7.00/src/model.py:418-434 `training_step`, :257-271 `L2Wrap`).
//
// The reference runs: contiguous copy of the shifted logits, log-softmax forward, NLL, log-softmax backward, a max over
// the vocabulary, a zero-fill of a logits-sized tensor, a scatter and a logits-sized add of the two gradients -- about
// 40 B per logit of HBM traffic (5.5 GB logits at 16 x 2624 tokens) against 2 (forward) + 4 (backward) here.
//   forward : per row  m = max_c x_c (first arg-max kept),  lse = m + log sum_c e^{x_c - m},  loss = lse - x_label
//   backward: dx_c = w_row (e^{x_c - lse} - [c = label]) + [c = argmax] m l2_factor
// w_row already contains the upstream gradient, 1/max(valid_b,1), 1/B and is 0 for ignored rows (label -100 or the
// last position of a sample); the L2Wrap term is NOT scaled by the upstream gradient (the reference returns it as is,
// model.py:270).  One workgroup per row, online soft-max (one read of the row), fp32 arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

constexpr int CE_THREADS = 256;

struct MS { float m, s; int idx; };
DEVFN MS ms_merge(MS a, MS b) {             // lower index wins ties
    MS r;
    const bool a_ge = a.m > b.m || (a.m == b.m && a.idx <= b.idx);
    r.m = a_ge ? a.m : b.m;
    r.idx = a_ge ? a.idx : b.idx;
    r.s = a.s * fast_exp(a.m - r.m) + b.s * fast_exp(b.m - r.m);
    return r;
}

__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(int V, const uint16_t* __restrict__ logits,
                                                            const long* __restrict__ labels, float* __restrict__ row_loss,
                                                            float* __restrict__ row_max, float* __restrict__ row_lse,
                                                            int* __restrict__ row_arg) {
    __shared__ float sm[CE_THREADS / 64], ss[CE_THREADS / 64];
    __shared__ int si[CE_THREADS / 64];
    const long row = blockIdx.x;
    const uint16_t* x = logits + row * (long)V;
    MS acc{-3.0e38f, 0.f, 0x7fffffff};
    for (int c0 = threadIdx.x * 8; c0 < V; c0 += CE_THREADS * 8) {
        const uint4 u = *reinterpret_cast<const uint4*>(x + c0);
        const float f[8] = {bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y), bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w)};
        float m8 = f[0];
        int i8 = 0;
#pragma unroll
        for (int e = 1; e < 8; ++e) if (f[e] > m8) { m8 = f[e]; i8 = e; }
        float s8 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s8 += fast_exp(f[e] - m8);
        acc = ms_merge(acc, MS{m8, s8, c0 + i8});
    }
    // wave reduction through LDS-free shuffles (ds_bpermute), then across the 4 waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int src = (threadIdx.x & 63) ^ off;
        MS o{lane_bcast(acc.m, src), lane_bcast(acc.s, src), (int)__float_as_uint(lane_bcast(__uint_as_float((uint32_t)acc.idx), src))};
        acc = ms_merge(acc, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wave] = acc.m; ss[wave] = acc.s; si[wave] = acc.idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MS t{sm[0], ss[0], si[0]};
        for (int w = 1; w < CE_THREADS / 64; ++w) t = ms_merge(t, MS{sm[w], ss[w], si[w]});
        const float lse = t.m + fast_log(t.s);
        const long lab = labels[row];
        row_max[row] = t.m;
        row_lse[row] = lse;
        row_arg[row] = t.idx;
        row_loss[row] = (lab >= 0 && lab < V) ? lse - bf16_to_f32(x[lab]) : 0.f;
    }
}

__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(int V, const uint16_t* __restrict__ logits,
                                                            const long* __restrict__ labels, const float* __restrict__ row_w,
                                                            const float* __restrict__ row_max, const float* __restrict__ row_lse,
                                                            const int* __restrict__ row_arg, float l2_factor,
                                                            uint16_t* __restrict__ dlogits) {
    const long row = blockIdx.x;
    const uint16_t* x = logits + row * (long)V;
    uint16_t* d = dlogits + row * (long)V;
    const float w = row_w[row], lse = row_lse[row], l2 = row_max[row] * l2_factor;
    const int arg = row_arg[row];
    const long lab = labels[row];
    for (int c0 = threadIdx.x * 8; c0 < V; c0 += CE_THREADS * 8) {
        const uint4 u = *reinterpret_cast<const uint4*>(x + c0);
        const float f[8] = {bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y), bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w)};
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float v = w != 0.f ? w * (fast_exp(f[e] - lse) - (c == lab ? 1.f : 0.f)) : 0.f;
            v += c == arg ? l2 : 0.f;
            g[e] = v;
        }
        *reinterpret_cast<uint4*>(d + c0) = make_uint4(cvt_pk_bf16(g[0], g[1]), cvt_pk_bf16(g[2], g[3]),
                                                       cvt_pk_bf16(g[4], g[5]), cvt_pk_bf16(g[6], g[7]));
    }
}

}  // namespace

extern "C" {

int vrwkv_ce_fwd_bf16(long nrows, int V, const void* logits, const long* labels, float* row_loss, float* row_max,
                      float* row_lse, int* row_argmax, void* stream) {
    if (nrows <= 0 || !logits || !labels || !row_loss || !row_max || !row_lse || !row_argmax) return VRWKV_EINVAL;
    if (V <= 0 || V % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)nrows), dim3(CE_THREADS), 0, (hipStream_t)stream, V,
                       (const uint16_t*)logits, labels, row_loss, row_max, row_lse, row_argmax);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_ce_bwd_bf16(long nrows, int V, const void* logits, const long* labels, const float* row_w, const float* row_max,
                      const float* row_lse, const int* row_argmax, float l2_factor, void* dlogits, void* stream) {
    if (nrows <= 0 || !logits || !labels || !row_w || !row_max || !row_lse || !row_argmax || !dlogits) return VRWKV_EINVAL;
    if (V <= 0 || V % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)nrows), dim3(CE_THREADS), 0, (hipStream_t)stream, V,
                       (const uint16_t*)logits, labels, row_w, row_max, row_lse, row_argmax, l2_factor, (uint16_t*)dlogits);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // extern "C"
