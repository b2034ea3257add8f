// Residual add + LayerNorm, forward and backward, one pass each (RWKV-7 Block: x = x + att(ln1(x)); x = x + ffn(ln2(x)),
// VisualRWKV-v7/v7.00/src/model.py:247-254, and ln_out, :318).
//
//   forward :  xn = bf16(x + delta)            (skipped when delta == nullptr: xn = x)
//              y  = (xn - mean) * rstd * w + b  with the statistics of the ROUNDED xn (what the reference's separate
//                                               bf16 add followed by nn.LayerNorm sees), fp32 arithmetic, one rounding
//   backward:  dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (xn - mean) * rstd
//              dw = sum_rows dy * xhat,  db = sum_rows dy
// The reference runs these as separate eager kernels (add, LayerNorm forward, LayerNorm backward x3, gradient add):
// 13 B/element forward and ~24 B/element backward against 8 + 8 here.
//
// A workgroup owns a contiguous range of token rows and walks it row by row; a thread owns 8 consecutive channels
// (16-byte accesses), so the per-channel parameter gradients stay in registers for the whole range and leave as one
// fp32 partial row per workgroup (summed by colsum_kernel in a fixed order: deterministic, no atomics).  Row
// statistics: DPP wave all-reduce, then one LDS slot per wave (double buffered by row parity -> one barrier per
// reduction).  The next row's loads are issued before the current row is reduced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>
#include <ln_kernels.h>

namespace {

using namespace vln;

#ifndef VRWKV_LN_BWD_GRID
#define VRWKV_LN_BWD_GRID 1024
#endif
#ifndef VRWKV_LN_MIX_BWD_GRID
#define VRWKV_LN_MIX_BWD_GRID 768
#endif
// Forward kernels: MANY short-lived workgroups keep more requests in flight than 2048 resident ones that walk 20 rows each
// (profiles/r4_eltwise_micro_ab.jsonl): add + LayerNorm one row per workgroup (-20 %), the lerp kernels four (the shifted row
// x[n-1] is the previous iteration's row in registers: with one row per workgroup it would be read and normalised twice)
constexpr int LN_ROWS_PER_WG = 1, LN_MIX_ROWS_PER_WG = 4;
inline int ln_grid(long ntok, int rows = LN_ROWS_PER_WG) {
    const long g = (ntok + rows - 1) / rows;
    return (int)(g < 1 ? 1 : g > (1L << 22) ? (1L << 22) : g);
}
// Backward kernels: one workgroup per resident slot (add_ln_bwd: 104 VGPRs, 4 workgroups of 256 threads per CU; ln_mix_bwd<1>: 162, 3 per
// CU) -- a single round of equal token ranges, and half / a third of the partial rows for ln_colsum_kernel to read (2048 rows cost
// 46 us per call, 2.3 ms per step)
constexpr int LN_BWD_GRID = VRWKV_LN_BWD_GRID, LN_MIX_BWD_GRID = VRWKV_LN_MIX_BWD_GRID;
inline int ln_bwd_grid(long ntok) { return (int)(ntok < LN_BWD_GRID ? ntok : LN_BWD_GRID); }
inline int ln_mix_bwd_grid(long ntok) { return (int)(ntok < LN_MIX_BWD_GRID ? ntok : LN_MIX_BWD_GRID); }
inline int ln_ok(int C) { return C > 0 && C % 64 == 0 && C <= 8192; }
inline int ln_threads(int C) { return (C / 8 + 63) / 64 * 64; }

}  // namespace

extern "C" {

long vrwkv_add_ln_ws_floats(long ntok, int C) { return (long)ln_bwd_grid(ntok) * 2 * C; }

int vrwkv_add_ln_fwd_bf16(long ntok, int C, float eps, const void* x, const void* delta, const void* w, const void* b,
                          void* xn, void* y, float* mean, float* rstd, void* stream) {
    if (ntok <= 0 || !x || !w || !b || !y || !mean || !rstd || (delta && !xn)) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(ln_grid(ntok)), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, eps,
                       (const uint16_t*)x, (const uint16_t*)delta, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn,
                       (uint16_t*)y, mean, rstd, (const long*)nullptr, (const uint16_t*)nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

// Inference form for the frozen ViT towers (timm pre-LN blocks, src/vision.py:123-134; SAM blocks, src/sam.py:231-247):
// xn = x + delta * dscale (dscale: LayerScale gamma, may be NULL), y = LayerNorm(xn); no statistics are kept.
int vrwkv_add_ln_scaled_fwd_bf16(long ntok, int C, float eps, const void* x, const void* delta, const void* dscale, const void* w,
                                 const void* b, void* xn, void* y, void* stream) {
    if (ntok <= 0 || !x || !w || !b || !y || (delta && !xn) || (dscale && !delta)) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(ln_grid(ntok)), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, eps,
                       (const uint16_t*)x, (const uint16_t*)delta, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn,
                       (uint16_t*)y, (float*)nullptr, (float*)nullptr, (const long*)nullptr, (const uint16_t*)dscale);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

// LayerNorm of the projector output written straight into the rows of the token-embedding tensor that hold the image
// placeholders (MLPWithContextGating's ln_v + the masked scatter of preparing_embedding, src/model.py:338,485-493):
// out[row_index[n]] = LN(x[n]).  row_index: device int64, distinct rows; a NEGATIVE entry drops feature row n (the sample
// had fewer placeholders than features -- the reference truncates the features, src/model.py:487-491): nothing is written
// for it and in the backward it receives a zero gradient and does not contribute to dgamma / dbeta.
int vrwkv_ln_scatter_fwd_bf16(long ntok, int C, float eps, const void* x, const void* w, const void* b, const long* row_index,
                              void* out, float* mean, float* rstd, void* stream) {
    if (ntok <= 0 || !x || !w || !b || !row_index || !out || !mean || !rstd) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(ln_grid(ntok)), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, eps,
                       (const uint16_t*)x, (const uint16_t*)nullptr, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)nullptr,
                       (uint16_t*)out, mean, rstd, row_index, (const uint16_t*)nullptr);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

// ... and its backward: dx[n] = LN'(dout[row_index[n]]), dwb = (dgamma, dbeta); ws as for vrwkv_add_ln_bwd_bf16
int vrwkv_ln_gather_bwd_bf16(long ntok, int C, const void* dout, const long* row_index, const void* x, const float* mean,
                             const float* rstd, const void* w, void* dx, float* dwb, float* ws, void* stream) {
    if (ntok <= 0 || !dout || !row_index || !x || !mean || !rstd || !w || !dx || !dwb || !ws) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    const int G = ln_bwd_grid(ntok);
    hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(G), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)dout,
                       (const uint16_t*)nullptr, (const uint16_t*)x, mean, rstd, (const uint16_t*)w, (uint16_t*)dx, ws, row_index);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)(2L * C / 16)), dim3(256), 0, (hipStream_t)stream, G, 2L * C, ws, dwb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_add_ln_bwd_bf16(long ntok, int C, const void* dy, const void* dres, const void* xn, const float* mean,
                          const float* rstd, const void* w, void* dx, float* dwb, float* ws, void* stream) {
    if (ntok <= 0 || !dy || !xn || !mean || !rstd || !w || !dx || !dwb || !ws) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    const int G = ln_bwd_grid(ntok);
    hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(G), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)dy,
                       (const uint16_t*)dres, (const uint16_t*)xn, mean, rstd, (const uint16_t*)w, (uint16_t*)dx, ws, (const long*)nullptr);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)(2L * C / 16)), dim3(256), 0, (hipStream_t)stream, G, 2L * C, ws, dwb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

// Residual add + LayerNorm + token shift + M lerps (M = 1: channel-mix, M = 6: time-mix), see ln_mix_fwd_kernel above.
long vrwkv_ln_mix_ws_floats(long ntok, int C, int M) { return (long)ln_mix_bwd_grid(ntok) * (2 + M) * C; }

int vrwkv_ln_mix_fwd_bf16(long ntok, int T, int C, float eps, int M, const void* x, const void* delta, const void* w, const void* b,
                          const void* const* mu, void* xn, void* const* out, float* mean, float* rstd, void* stream) {
    if (ntok <= 0 || T <= 0 || ntok % T != 0 || !x || !w || !b || !mu || !out || !mean || !rstd || (delta && !xn)) return VRWKV_EINVAL;
    if (!ln_ok(C) || (M != 1 && M != 6)) return VRWKV_ESHAPE;
    LmPtrs pm{}; LmOuts po{};
    for (int j = 0; j < M; ++j) {
        if (!mu[j] || !out[j]) return VRWKV_EINVAL;
        pm.p[j] = (const uint16_t*)mu[j]; po.p[j] = (uint16_t*)out[j];
    }
    const dim3 grid(ln_grid(ntok, LN_MIX_ROWS_PER_WG)), block(ln_threads(C));
    hipStream_t st = (hipStream_t)stream;
    if (M == 1) hipLaunchKernelGGL(ln_mix_fwd_kernel<1>, grid, block, 0, st, ntok, T, C, eps, (const uint16_t*)x, (const uint16_t*)delta,
                                   (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn, mean, rstd, pm, po);
    else hipLaunchKernelGGL(ln_mix_fwd_kernel<6>, grid, block, 0, st, ntok, T, C, eps, (const uint16_t*)x, (const uint16_t*)delta,
                            (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn, mean, rstd, pm, po);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

// dx = dres + LN'(gradient of the lerps' input), dwb = (dgamma, dbeta) (2, C) fp32, dmu (M, C) fp32; dres may be NULL;
// ws: vrwkv_ln_mix_ws_floats(ntok, C, M) floats.  M = 1 only: with six lerps the kernel keeps ~200 values per thread (48 gradient
// accumulators, the prefetched rows) and hipcc spills inside the token loop at two workgroups per CU (1.02 ms against 0.72 ms for
// the two kernels); the time-mix backward is vrwkv_mix_bwd_ln_bf16 (tmix_fused.hip) followed by vrwkv_add_ln_bwd_bf16.
int vrwkv_ln_mix_bwd_bf16(long ntok, int T, int C, int M, const void* xn, const float* mean, const float* rstd, const void* w,
                          const void* b, const void* const* mu, const void* const* dout, const void* dout3_second, const void* dres,
                          void* dx, float* dwb, float* dmu, float* ws, void* stream) {
    if (ntok <= 0 || T <= 0 || ntok % T != 0 || !xn || !mean || !rstd || !w || !b || !mu || !dout || !dx || !dwb || !dmu || !ws) return VRWKV_EINVAL;
    if (!ln_ok(C) || M != 1 || dout3_second) return VRWKV_ESHAPE;        // M = 6: vrwkv_mix_bwd_ln_bf16 + vrwkv_add_ln_bwd_bf16 (below)
    LmPtrs pm{}, pd{};
    for (int j = 0; j < M; ++j) {
        if (!mu[j] || !dout[j]) return VRWKV_EINVAL;
        pm.p[j] = (const uint16_t*)mu[j]; pd.p[j] = (const uint16_t*)dout[j];
    }
    const int G = ln_mix_bwd_grid(ntok);
    const dim3 grid(G), block(ln_threads(C));
    hipStream_t st = (hipStream_t)stream;
    float* part_ln = ws; float* part_mu = ws + (size_t)G * 2 * C;
#define LN_MIX_BWD_LB(MM, DUP, LB) hipLaunchKernelGGL((ln_mix_bwd_kernel<MM, DUP, LB>), grid, block, 0, st, ntok, T, C, (const uint16_t*)xn, mean, rstd, \
        (const uint16_t*)w, (const uint16_t*)b, pm, pd, (const uint16_t*)dout3_second, (const uint16_t*)dres, (uint16_t*)dx, part_ln, part_mu)
#define LN_MIX_BWD(MM, DUP) do { if (block.x <= 256) LN_MIX_BWD_LB(MM, DUP, 256); else if (block.x <= 512) LN_MIX_BWD_LB(MM, DUP, 512); \
                                 else LN_MIX_BWD_LB(MM, DUP, 1024); } while (0)
    LN_MIX_BWD(1, false);
#undef LN_MIX_BWD
#undef LN_MIX_BWD_LB
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)(2L * C / 16)), dim3(256), 0, st, G, 2L * C, part_ln, dwb);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)((long)M * C / 16)), dim3(256), 0, st, G, (long)M * C, part_mu, dmu);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // extern "C"
