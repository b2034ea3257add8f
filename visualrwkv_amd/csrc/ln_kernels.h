// Kernels of csrc/ln_fused.hip (residual add + LayerNorm, and the same fused with the token shift + lerps of the time-mix /
// channel-mix): kept in a header so that the host lockstep emulator (tests/emu) can run them on the CPU.  See ln_fused.hip for
// the math, the byte counts and the C entry points.
#pragma once
#include <gfx950_prims.h>

namespace vln {

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
// Row pieces are streamed: every activation row is read once and written once per kernel, and the tensors (172 MB and more) do not
// survive in L2 / MALL until their consumer runs -- non-temporal accesses (measured per kernel in profiles/r4_eltwise_micro_ab.jsonl)
#ifndef VRWKV_NT
#define VRWKV_NT 1
#endif
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
DEVFN uint4 ldg(const uint16_t* p) {
#if VRWKV_NT
    const u32x4_nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(u[0], u[1], u[2], u[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
DEVFN void stg(uint16_t* p, uint4 v) {
#if VRWKV_NT
    const u32x4_nt u = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(u, reinterpret_cast<u32x4_nt*>(p));
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
// empty asm that "redefines" a packed loop-invariant row: keeps the compiler from hoisting its unpacked form (twice the registers)
// out of the token loop
DEVFN void keep_packed(uint4& u) { pin_vgpr4(u.x, u.y, u.z, u.w); }

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
            if (act) stg(xn + n * C + c0, r);
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
        if (act && orow >= 0) stg(y + orow * C + c0, pack8(o));
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
        if (act) stg(dx + n * C + c0, pack8(o));
    }
    if (act) {
        float* dst = part + (size_t)blockIdx.x * 2 * C + c0;
        *reinterpret_cast<float4*>(dst) = make_float4(gw.f[0], gw.f[1], gw.f[2], gw.f[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(gw.f[4], gw.f[5], gw.f[6], gw.f[7]);
        *reinterpret_cast<float4*>(dst + C) = make_float4(gb.f[0], gb.f[1], gb.f[2], gb.f[3]);
        *reinterpret_cast<float4*>(dst + C + 4) = make_float4(gb.f[4], gb.f[5], gb.f[6], gb.f[7]);
    }
}

// out[j] = sum_g part[g][j], fixed order (same scheme as tmix_fused.hip's colsum_kernel); width % 16 == 0
__global__ __launch_bounds__(256) void ln_colsum_kernel(int G, long width, const float* __restrict__ part, float* __restrict__ out) {
    // A workgroup owns 16 columns; thread (cq = tid & 3, rg = tid >> 2) sums rows rg, rg+64, ... of 4 adjacent columns (float4); the 64
    // row groups are combined through LDS in a fixed order (deterministic).  (64 columns per workgroup left half of the CUs without
    // work for the 2-8 k columns of a parameter gradient: 23 / 46 us per call, 5 ms per training step.)
    __shared__ float4 red[64][4];
    __shared__ float4 red2[8][4];
    const int cq = threadIdx.x & 3, rg = threadIdx.x >> 2;
    const long col = (long)blockIdx.x * 16 + 4 * cq;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = rg; g < G; g += 64) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)g * width + col);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    red[rg][cq] = a;
    __syncthreads();
    if (rg < 8) {
        float4 t = red[rg][cq];
#pragma unroll
        for (int r = rg + 8; r < 64; r += 8) { const float4 v = red[r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        red2[rg][cq] = t;
    }
    __syncthreads();
    if (rg == 0) {
        float4 t = red2[0][cq];
#pragma unroll
        for (int r = 1; r < 8; ++r) { const float4 v = red2[r][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + col) = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Residual add + LayerNorm + token shift + M lerps in one pass (Block: x = x + att(ln1(x)) with RWKV_Tmix_x070's six lerps,
// x = x + ffn(ln2(x)) with RWKV_CMix_x070's one: VisualRWKV-v7/v7.00/src/model.py:247-254,166-173,222-223).  The LayerNorm output is
// used by nothing but the lerps, so it is never written: forward 4 B/element less than add_ln + mix (8 + 14 -> 18 for M = 6,
// 8 + 4 -> 8 for M = 1), backward 6 B/element less (16 + 8 -> 18, 6 + 8 -> 8), and one saved activation less per LayerNorm.
// Same arithmetic as the two-kernel path, rounding included (the LayerNorm output and the lerps' input gradient are rounded to
// bf16 where that path stores them), so outputs and input gradients are bit-identical to it.
//   forward : xn = bf16(x + delta);  y = bf16(LN(xn));  out_j[n] = y[n] + (y[n-1] - y[n]) mu_j   (y[-1] = 0 at the start of a sample)
//   backward: dyl[n] = bf16(A[n] + Bv[n+1]),  A = sum_j d_j (1 - mu_j),  Bv = sum_j d_j mu_j;  dx = dres + LN'(dyl);
//             dmu_j = sum_n d_j[n] (y[n-1] - y[n]),  dw, db as in add_ln_bwd;  y is recomputed from xn and the saved statistics.
// A workgroup walks a contiguous token range in order (the row before the range is recomputed / the row after it read once).
struct LmPtrs { const uint16_t* p[6]; };
struct LmOuts { uint16_t* p[6]; };

template <int M>
__global__ __launch_bounds__(1024) void ln_mix_fwd_kernel(long ntok, int T, int C, float eps, const uint16_t* __restrict__ x,
                                                          const uint16_t* __restrict__ delta, const uint16_t* __restrict__ w,
                                                          const uint16_t* __restrict__ b, uint16_t* __restrict__ xn,
                                                          float* __restrict__ mean, float* __restrict__ rstd, LmPtrs mu, LmOuts out) {
    __shared__ float red[4][MAXW][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c0 = threadIdx.x * 8;
    const bool act = c0 < C;
    const long lo = ntok * blockIdx.x / gridDim.x, hi = ntok * (blockIdx.x + 1) / gridDim.x;
    if (lo >= hi) return;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const V8 wv = unpack8(act ? ldg(w + c0) : z4), bv = unpack8(act ? ldg(b + c0) : z4);
    uint4 mp[M];                                                 // lerp weights, packed (unpacked where used: registers)
#pragma unroll
    for (int j = 0; j < M; ++j) mp[j] = act ? ldg(mu.p[j] + c0) : z4;
    const float inv_c = 1.f / (float)C;
    const long n0 = (lo % T != 0) ? lo - 1 : lo;                 // the row before the range: only its LayerNorm output is needed
    uint4 nx = act ? ldg(x + n0 * C + c0) : z4, nd = (act && delta) ? ldg(delta + n0 * C + c0) : z4;
    V8 prev;
#pragma unroll
    for (int e = 0; e < 8; ++e) prev.f[e] = 0.f;
    int tpos = (int)(n0 % T);                                    // position of row n inside its sample (one division per workgroup)
    for (long n = n0; n < hi; ++n) {
        const uint4 cx = nx, cd = nd;
        if (n + 1 < hi && act) {
            nx = ldg(x + (n + 1) * C + c0);
            if (delta) nd = ldg(delta + (n + 1) * C + c0);
        }
        const bool own = n >= lo;
        V8 v = unpack8(cx);
        if (delta) {
            const V8 d = unpack8(cd);
#pragma unroll
            for (int e = 0; e < 8; ++e) v.f[e] += d.f[e];
            const uint4 r = pack8(v);
            if (act && own) stg(xn + n * C + c0, r);
            v = unpack8(r);
        }
        const int par = (int)(n & 1) * 2;
        float s[1] = {0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) s[0] += v.f[e];
        block_sum<1>(red, par, wave, lane, nw, s);
        const float mu_ = s[0] * inv_c;
        float q[1] = {0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = act ? v.f[e] - mu_ : 0.f; q[0] = fmaf(t, t, q[0]); }
        block_sum<1>(red, par + 1, wave, lane, nw, q);
        const float rs = rsqrtf(q[0] * inv_c + eps);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.f[e] = fmaf((v.f[e] - mu_) * rs, wv.f[e], bv.f[e]);
        const V8 cur = unpack8(pack8(o));                        // the bf16 value the two-kernel path stores and re-reads
        if (own) {
            const bool first = tpos == 0;
            V8 xx;
#pragma unroll
            for (int e = 0; e < 8; ++e) xx.f[e] = (first ? 0.f : prev.f[e]) - cur.f[e];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const V8 m = unpack8(mp[j]);
                V8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) r.f[e] = fmaf(xx.f[e], m.f[e], cur.f[e]);
                if (act) stg(out.p[j] + n * C + c0, pack8(r));
            }
            if (threadIdx.x == 0) { mean[n] = mu_; rstd[n] = rs; }
        }
        prev = cur;
        if (++tpos == T) tpos = 0;
    }
}

// LayerNorm backward of one finished row (the body of add_ln_bwd_kernel's loop): dyl = gradient of the LayerNorm output (already
// rounded to bf16), xh / rs of that row, res = the residual path's gradient (packed; zeros: none)
DEVFN void ln_row_bwd(float (*red)[MAXW][2], int slot, int wave, int lane, int nw, bool act, float inv_c, const V8& wv, const V8& dyl,
                      const V8& xh, float rs, uint4 res, V8& gw, V8& gb, uint16_t* dst) {
    V8 g;
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        g.f[e] = dyl.f[e] * wv.f[e];
        s[0] += g.f[e];
        s[1] = fmaf(g.f[e], xh.f[e], s[1]);
        gw.f[e] = fmaf(dyl.f[e], xh.f[e], gw.f[e]);
        gb.f[e] += dyl.f[e];
    }
    block_sum<2>(red, slot, wave, lane, nw, s);
    const float c1 = s[0] * inv_c, c2 = s[1] * inv_c;
    V8 o = unpack8(res);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.f[e] = fmaf(rs, g.f[e] - c1 - xh.f[e] * c2, o.f[e]);
    if (act) stg(dst, pack8(o));
}

// DUP3: output 3 (x_v) has two consumers; their gradients arrive as dout.p[3] and dout3b (see mix_bwd_kernel in tmix_fused.hip).
// LB: threads per workgroup the instantiation is compiled for (C / 8 rounded up to a wave).  Register discipline (M = 6 keeps 48
// gradient accumulators per thread): rows travel packed (bf16) and are unpacked where used, the previous row is carried as packed
// xn + its two statistics and its LayerNorm output is recomputed, and the next row's loads are issued after this row's values
// have been consumed and before the reduction -- the kernel needs ~2 us of HBM time per row and CU, the arithmetic ~0.3.
template <int M, bool DUP3, int LB>
__global__ __launch_bounds__(LB, LB <= 256 ? 2 : 1) void ln_mix_bwd_kernel(long ntok, int T, int C, const uint16_t* __restrict__ xn,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const uint16_t* __restrict__ w, const uint16_t* __restrict__ b, LmPtrs mu,
                                                        LmPtrs dout, const uint16_t* __restrict__ dout3b,
                                                        const uint16_t* __restrict__ dres, uint16_t* __restrict__ dx,
                                                        float* __restrict__ part_ln, float* __restrict__ part_mu) {
    __shared__ float red[2][MAXW][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c0 = threadIdx.x * 8;
    const bool act = c0 < C;
    const long lo = ntok * blockIdx.x / gridDim.x, hi = ntok * (blockIdx.x + 1) / gridDim.x;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const int cc = act ? c0 : 0;                                 // inactive lanes (C / 8 not a multiple of 64) read column 0, store nothing
    uint4 wp = act ? ldg(w + c0) : z4, bp = act ? ldg(b + c0) : z4;
    uint4 mp[M];
#pragma unroll
    for (int j = 0; j < M; ++j) mp[j] = act ? ldg(mu.p[j] + c0) : z4;
    const float inv_c = 1.f / (float)C;
    V8 gw, gb, gm[M];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gw.f[e] = 0.f; gb.f[e] = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) gm[j].f[e] = 0.f;
    }
    // LayerNorm output of a row as the forward rounded it, from its packed xn and statistics
    auto ln_out = [&](uint4 xp, float m0, float r0) {
        const V8 xv = unpack8(xp), wv = unpack8(wp), bv = unpack8(bp);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.f[e] = fmaf((xv.f[e] - m0) * r0, wv.f[e], bv.f[e]);
        return unpack8(pack8(o));
    };
    if (lo < hi) {
        V8 aprev;
#pragma unroll
        for (int e = 0; e < 8; ++e) aprev.f[e] = 0.f;
        uint4 xprev = z4, resprev = z4;                          // previous row: packed xn, residual gradient, statistics
        float muprev = 0.f, rsprev = 0.f;
        if (lo % T != 0) { xprev = ldg(xn + (lo - 1) * C + cc); muprev = mean[lo - 1]; rsprev = rstd[lo - 1]; }
        // rows lo .. hi-1 in full; row hi (if it continues the last sequence) contributes only Bv to the gradient of row hi-1
        const long last = (hi < ntok && hi % T != 0) ? hi : hi - 1;
        uint4 nd[M], nd3 = z4, nx, nr = z4;
        float nmu, nrs;
#pragma unroll
        for (int j = 0; j < M; ++j) nd[j] = ldg(dout.p[j] + lo * C + cc);
        if (DUP3) nd3 = ldg(dout3b + lo * C + cc);
        nx = ldg(xn + lo * C + cc);
        if (dres) nr = ldg(dres + lo * C + cc);
        nmu = mean[lo]; nrs = rstd[lo];
        int slot = 0;
        int tpos = (int)(lo % T);                                // position of row n inside its sample
        for (long n = lo; n <= last; ++n) {
            const bool inside = n < hi, cont = tpos != 0;
            keep_packed(wp); keep_packed(bp);
#pragma unroll
            for (int j = 0; j < M; ++j) keep_packed(mp[j]);
            const uint4 cx = nx, cr = nr;
            const float mu_ = nmu, rs = nrs;
            // shift difference of the lerps at this row: y[n-1] - y[n]
            V8 xx;
            {
                const V8 y = ln_out(cx, mu_, rs);
                const V8 yp = ln_out(xprev, muprev, rsprev);
#pragma unroll
                for (int e = 0; e < 8; ++e) xx.f[e] = inside ? (cont ? yp.f[e] : 0.f) - y.f[e] : 0.f;
            }
            V8 dsum, bvv;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dsum.f[e] = 0.f; bvv.f[e] = 0.f; }
#pragma unroll
            for (int j = 0; j < M; ++j) {
                V8 d = unpack8(nd[j]);
                if (DUP3 && j == (M > 3 ? 3 : 0)) {
                    const V8 d2 = unpack8(nd3);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d.f[e] += d2.f[e];
                }
                const V8 m = unpack8(mp[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dsum.f[e] += d.f[e];
                    bvv.f[e] = fmaf(d.f[e], m.f[e], bvv.f[e]);
                    gm[j].f[e] = fmaf(d.f[e], xx.f[e], gm[j].f[e]);
                }
            }
            if (n + 1 <= last) {                                 // next row's loads: this row's registers are free, the reduction is ahead
                const long o = (n + 1) * C + cc;
#pragma unroll
                for (int j = 0; j < M; ++j) nd[j] = ldg(dout.p[j] + o);
                if (DUP3) nd3 = ldg(dout3b + o);
                if (n + 1 < hi) {
                    nx = ldg(xn + o);
                    if (dres) nr = ldg(dres + o);
                    nmu = mean[n + 1]; nrs = rstd[n + 1];
                }
            }
            if (n > lo) {                                        // the previous row is complete: its LayerNorm backward
                V8 t, xh;
                const V8 xv = unpack8(xprev);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    t.f[e] = aprev.f[e] + (cont ? bvv.f[e] : 0.f);
                    xh.f[e] = act ? (xv.f[e] - muprev) * rsprev : 0.f;
                }
                ln_row_bwd(red, slot, wave, lane, nw, act, inv_c, unpack8(wp), unpack8(pack8(t)), xh, rsprev, resprev, gw, gb, dx + (n - 1) * C + cc);
                slot ^= 1;
            }
            if (inside) {
#pragma unroll
                for (int e = 0; e < 8; ++e) aprev.f[e] = dsum.f[e] - bvv.f[e];
                xprev = cx; resprev = cr; muprev = mu_; rsprev = rs;
            }
            if (++tpos == T) tpos = 0;
        }
        if (last == hi - 1) {                                    // no successor row: the gradient of the last row is A
            V8 xh;
            const V8 xv = unpack8(xprev);
#pragma unroll
            for (int e = 0; e < 8; ++e) xh.f[e] = act ? (xv.f[e] - muprev) * rsprev : 0.f;
            ln_row_bwd(red, slot, wave, lane, nw, act, inv_c, unpack8(wp), unpack8(pack8(aprev)), xh, rsprev, resprev, gw, gb, dx + (hi - 1) * C + cc);
        }
    }
    if (act) {
        float* dst = part_ln + (size_t)blockIdx.x * 2 * C + c0;
        *reinterpret_cast<float4*>(dst) = make_float4(gw.f[0], gw.f[1], gw.f[2], gw.f[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(gw.f[4], gw.f[5], gw.f[6], gw.f[7]);
        *reinterpret_cast<float4*>(dst + C) = make_float4(gb.f[0], gb.f[1], gb.f[2], gb.f[3]);
        *reinterpret_cast<float4*>(dst + C + 4) = make_float4(gb.f[4], gb.f[5], gb.f[6], gb.f[7]);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float* dm = part_mu + ((size_t)blockIdx.x * M + j) * C + c0;
            *reinterpret_cast<float4*>(dm) = make_float4(gm[j].f[0], gm[j].f[1], gm[j].f[2], gm[j].f[3]);
            *reinterpret_cast<float4*>(dm + 4) = make_float4(gm[j].f[4], gm[j].f[5], gm[j].f[6], gm[j].f[7]);
        }
    }
}

}  // namespace vln
