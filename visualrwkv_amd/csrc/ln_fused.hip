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

namespace {

struct V8 { float f[8]; };
DEVFN V8 unpack8(uint4 u) {
    V8 r;
    r.f[0] = bf16_lo(u.x); r.f[1] = bf16_hi(u.x); r.f[2] = bf16_lo(u.y); r.f[3] = bf16_hi(u.y);
    r.f[4] = bf16_lo(u.z); r.f[5] = bf16_hi(u.z); r.f[6] = bf16_lo(u.w); r.f[7] = bf16_hi(u.w);
    return r;
}
DEVFN uint4 pack8(const V8& v) {
    return make_uint4(cvt_pk_bf16(v.f[0], v.f[1]), cvt_pk_bf16(v.f[2], v.f[3]), cvt_pk_bf16(v.f[4], v.f[5]), cvt_pk_bf16(v.f[6], v.f[7]));
}
DEVFN uint4 ldg(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

constexpr int MAXW = 16;          // waves per workgroup (C <= 8192)

// all-reduce of NV values over the workgroup; `slot` alternates between consecutive calls
template <int NV>
DEVFN void block_sum(float (*red)[MAXW][2], int slot, int wave, int lane, int nw, float* v) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = group_sum<6>(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[slot][wave][i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    for (int w = 0; w < nw; ++w) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += red[slot][w][i];
    }
}

__global__ __launch_bounds__(1024) void add_ln_fwd_kernel(long ntok, int C, float eps, const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ delta, const uint16_t* __restrict__ w,
                                                          const uint16_t* __restrict__ b, uint16_t* __restrict__ xn,
                                                          uint16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                          const long* __restrict__ yrow, const uint16_t* __restrict__ dscale = nullptr) {
    __shared__ float red[4][MAXW][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c0 = threadIdx.x * 8;
    const bool act = c0 < C;
    const long lo = ntok * blockIdx.x / gridDim.x, hi = ntok * (blockIdx.x + 1) / gridDim.x;
    if (lo >= hi) return;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const V8 wv = unpack8(act ? ldg(w + c0) : z4), bv = unpack8(act ? ldg(b + c0) : z4);
    const V8 sv = unpack8((act && dscale) ? ldg(dscale + c0) : z4);         // optional per-channel scale of delta (ViT LayerScale)
    const float inv_c = 1.f / (float)C;
    uint4 nx = act ? ldg(x + lo * C + c0) : z4, nd = (act && delta) ? ldg(delta + lo * C + c0) : z4;
    for (long n = lo; n < hi; ++n) {
        const uint4 cx = nx, cd = nd;
        if (n + 1 < hi && act) {
            nx = ldg(x + (n + 1) * C + c0);
            if (delta) nd = ldg(delta + (n + 1) * C + c0);
        }
        V8 v = unpack8(cx);
        if (delta) {
            const V8 d = unpack8(cd);
            if (dscale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.f[e] = fmaf(d.f[e], sv.f[e], v.f[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.f[e] += d.f[e];
            }
            const uint4 r = pack8(v);
            if (act) *reinterpret_cast<uint4*>(xn + n * C + c0) = r;
            v = unpack8(r);
        }
        const int par = (int)(n & 1) * 2;
        float s[1] = {0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) s[0] += v.f[e];
        block_sum<1>(red, par, wave, lane, nw, s);
        const float mu = s[0] * inv_c;
        float q[1] = {0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = act ? v.f[e] - mu : 0.f; q[0] = fmaf(t, t, q[0]); }
        block_sum<1>(red, par + 1, wave, lane, nw, q);
        const float rs = rsqrtf(q[0] * inv_c + eps);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.f[e] = fmaf((v.f[e] - mu) * rs, wv.f[e], bv.f[e]);
        const long orow = yrow ? yrow[n] : n;                  // yrow: scatter into a larger tensor; a negative row is dropped
        if (act && orow >= 0) *reinterpret_cast<uint4*>(y + orow * C + c0) = pack8(o);
        if (threadIdx.x == 0 && mean) { mean[n] = mu; rstd[n] = rs; }
    }
}

__global__ __launch_bounds__(1024) void add_ln_bwd_kernel(long ntok, int C, const uint16_t* __restrict__ dy,
                                                          const uint16_t* __restrict__ dres, const uint16_t* __restrict__ xn,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const uint16_t* __restrict__ w, uint16_t* __restrict__ dx,
                                                          float* __restrict__ part, const long* __restrict__ yrow) {
    __shared__ float red[2][MAXW][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c0 = threadIdx.x * 8;
    const bool act = c0 < C;
    const long lo = ntok * blockIdx.x / gridDim.x, hi = ntok * (blockIdx.x + 1) / gridDim.x;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const V8 wv = unpack8(act ? ldg(w + c0) : z4);
    const float inv_c = 1.f / (float)C;
    V8 gw, gb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { gw.f[e] = 0.f; gb.f[e] = 0.f; }
    uint4 ny = z4, nx = z4, nr = z4;
    float nmu = 0.f, nrs = 0.f;
    if (lo < hi) {
        if (act) { const long r = yrow ? yrow[lo] : lo; ny = r >= 0 ? ldg(dy + r * C + c0) : z4; nx = ldg(xn + lo * C + c0); if (dres) nr = ldg(dres + lo * C + c0); }
        nmu = mean[lo]; nrs = rstd[lo];
    }
    for (long n = lo; n < hi; ++n) {
        const uint4 cy = ny, cx = nx, cr = nr;
        const float mu = nmu, rs = nrs;
        if (n + 1 < hi) {
            if (act) { const long r = yrow ? yrow[n + 1] : n + 1; ny = r >= 0 ? ldg(dy + r * C + c0) : z4; nx = ldg(xn + (n + 1) * C + c0); if (dres) nr = ldg(dres + (n + 1) * C + c0); }
            nmu = mean[n + 1]; nrs = rstd[n + 1];
        }
        const V8 d = unpack8(cy), xv = unpack8(cx);
        V8 xh, g;
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh.f[e] = act ? (xv.f[e] - mu) * rs : 0.f;
            g.f[e] = d.f[e] * wv.f[e];
            s[0] += g.f[e];
            s[1] = fmaf(g.f[e], xh.f[e], s[1]);
            gw.f[e] = fmaf(d.f[e], xh.f[e], gw.f[e]);
            gb.f[e] += d.f[e];
        }
        block_sum<2>(red, (int)(n & 1), wave, lane, nw, s);
        const float c1 = s[0] * inv_c, c2 = s[1] * inv_c;
        V8 o = unpack8(cr);                         // zeros when there is no residual gradient
#pragma unroll
        for (int e = 0; e < 8; ++e) o.f[e] = fmaf(rs, g.f[e] - c1 - xh.f[e] * c2, o.f[e]);
        if (act) *reinterpret_cast<uint4*>(dx + n * C + c0) = pack8(o);
    }
    if (act) {
        float* dst = part + (size_t)blockIdx.x * 2 * C + c0;
        *reinterpret_cast<float4*>(dst) = make_float4(gw.f[0], gw.f[1], gw.f[2], gw.f[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(gw.f[4], gw.f[5], gw.f[6], gw.f[7]);
        *reinterpret_cast<float4*>(dst + C) = make_float4(gb.f[0], gb.f[1], gb.f[2], gb.f[3]);
        *reinterpret_cast<float4*>(dst + C + 4) = make_float4(gb.f[4], gb.f[5], gb.f[6], gb.f[7]);
    }
}

// out[j] = sum_g part[g][j], fixed order (same scheme as tmix_fused.hip's colsum_kernel); width % 64 == 0
__global__ __launch_bounds__(256) void ln_colsum_kernel(int G, long width, const float* __restrict__ part, float* __restrict__ out) {
    __shared__ float4 red[16][16];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const long col = (long)blockIdx.x * 64 + 4 * cq;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = rg; g < G; g += 16) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)g * width + col);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    red[rg][cq] = a;
    __syncthreads();
    if (rg == 0) {
        float4 t = red[0][cq];
#pragma unroll
        for (int r = 1; r < 16; ++r) { const float4 v = red[r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + col) = t;
    }
}

constexpr int LN_GRID = 2048;
inline int ln_grid(long ntok) { return (int)(ntok < LN_GRID ? ntok : LN_GRID); }
inline int ln_ok(int C) { return C > 0 && C % 64 == 0 && C <= 8192; }
inline int ln_threads(int C) { return (C / 8 + 63) / 64 * 64; }

}  // namespace

extern "C" {

long vrwkv_add_ln_ws_floats(long ntok, int C) { return (long)ln_grid(ntok) * 2 * C; }

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
    const int G = ln_grid(ntok);
    hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(G), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)dout,
                       (const uint16_t*)nullptr, (const uint16_t*)x, mean, rstd, (const uint16_t*)w, (uint16_t*)dx, ws, row_index);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)(2L * C / 64)), dim3(256), 0, (hipStream_t)stream, G, 2L * C, ws, dwb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_add_ln_bwd_bf16(long ntok, int C, const void* dy, const void* dres, const void* xn, const float* mean,
                          const float* rstd, const void* w, void* dx, float* dwb, float* ws, void* stream) {
    if (ntok <= 0 || !dy || !xn || !mean || !rstd || !w || !dx || !dwb || !ws) return VRWKV_EINVAL;
    if (!ln_ok(C)) return VRWKV_ESHAPE;
    const int G = ln_grid(ntok);
    hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(G), dim3(ln_threads(C)), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)dy,
                       (const uint16_t*)dres, (const uint16_t*)xn, mean, rstd, (const uint16_t*)w, (uint16_t*)dx, ws, (const long*)nullptr);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)(2L * C / 64)), dim3(256), 0, (hipStream_t)stream, G, 2L * C, ws, dwb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // extern "C"
