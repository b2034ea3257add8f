// Fused streaming kernels for the element-wise glue of RWKV-7 time-mix / channel-mix (forward + backward).
//
// The reference runs this glue as ~45 separate PyTorch eager kernels per layer and direction
// (VisualRWKV-v7/v7.00/src/model.py:166-173 token-shift + 6 lerps, :176 decay soft-clamp, :179-188 value
// residual / a-gate / kk normalise / k modulate, :191-194 GroupNorm + bonus + gate, :222-225 channel-mix lerp +
// relu^2); on MI355X that is pure HBM traffic (measured: 242 ms of a 510 ms training step).  Here each group is
// one pass: a workgroup owns TPB consecutive tokens, a thread owns 8 consecutive channels (16-byte bf16
// accesses; a 64-channel head = 8 adjacent lanes, reduced with DPP), arithmetic in fp32, one rounding to bf16
// at the end.  Per-channel parameter gradients are accumulated in registers over the workgroup's tokens, written
// as one fp32 partial row per workgroup and summed by a second tiny kernel (no atomics: same-address fp32
// atomics from ~1300 workgroups serialised and made the backward kernels 3-4x slower than HBM-bound;
// this is also deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

constexpr int TPB = 16;      // tokens per workgroup (granularity of the backward kernels' token ranges)
// Forward kernels: a workgroup owns ceil(ntok / gridDim.x) consecutive tokens.  MANY short-lived workgroups keep more requests in
// flight than fewer resident ones that walk 16 tokens each (per kernel in profiles/r4_eltwise_micro_ab.jsonl: post -14 % at one token
// per workgroup, kva -8 % at two; the lerps want four -- the shifted row x[n-1] is the previous iteration's row)
constexpr int TPB_MIX = 4, TPB_DECAY = 2, TPB_KVA = 2, TPB_POST = 1, TPB_GN = 1;
DEVFN long own_lo(long ntok) { return (long)blockIdx.x * ((ntok + gridDim.x - 1) / gridDim.x); }
DEVFN long own_hi(long ntok) { const long t = (ntok + gridDim.x - 1) / gridDim.x, e = ((long)blockIdx.x + 1) * t; return e < ntok ? e : ntok; }
constexpr int MAXM = 6;

struct V8 { float f[8]; };

// Activation rows are streamed (read once, written once, 172 MB and more per tensor: nothing survives in L2 / MALL until its consumer
// runs): non-temporal accesses, measured per kernel in profiles/r4_eltwise_micro_ab.jsonl
#ifndef VRWKV_NT
#define VRWKV_NT 1
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
DEVFN u32x4_t ld8raw(const uint16_t* p) {
#if VRWKV_NT
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#else
    return *reinterpret_cast<const u32x4_t*>(p);
#endif
}
DEVFN V8 cvt8(u32x4_t u) {
    V8 r;
    r.f[0] = bf16_lo(u[0]); r.f[1] = bf16_hi(u[0]); r.f[2] = bf16_lo(u[1]); r.f[3] = bf16_hi(u[1]);
    r.f[4] = bf16_lo(u[2]); r.f[5] = bf16_hi(u[2]); r.f[6] = bf16_lo(u[3]); r.f[7] = bf16_hi(u[3]);
    return r;
}
DEVFN V8 ld8f(const uint16_t* p) { return cvt8(ld8raw(p)); }
DEVFN void st8f(uint16_t* p, const V8& v) {
    const u32x4_t u = {cvt_pk_bf16(v.f[0], v.f[1]), cvt_pk_bf16(v.f[2], v.f[3]), cvt_pk_bf16(v.f[4], v.f[5]), cvt_pk_bf16(v.f[6], v.f[7])};
#if VRWKV_NT
    __builtin_nontemporal_store(u, reinterpret_cast<u32x4_t*>(p));
#else
    *reinterpret_cast<u32x4_t*>(p) = u;
#endif
}
DEVFN V8 zero8() { V8 r; for (int e = 0; e < 8; ++e) r.f[e] = 0.f; return r; }
// software-pipelined backward loops (kva_bwd, post_bwd): the NEXT token's raw 16-byte pieces are requested before the current token's
// arithmetic and stores, converted when used
#ifndef VRWKV_BWD_PF
#define VRWKV_BWD_PF 1
#endif
// per-workgroup partial of a column sum: part[blockIdx.x][vec][c]
DEVFN void put_partial(float* part, int nvec, int vec, int C, int c0, const V8& v) {
    float* dst = part + ((size_t)blockIdx.x * nvec + vec) * C + c0;
    *reinterpret_cast<float4*>(dst) = make_float4(v.f[0], v.f[1], v.f[2], v.f[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v.f[4], v.f[5], v.f[6], v.f[7]);
}
// out[j] = sum_g part[g][j];  width % 16 == 0.
__global__ __launch_bounds__(256) void colsum_kernel(int G, long width, const float* __restrict__ part, float* __restrict__ out) {
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
// Backward kernels: workgroup g of G owns the contiguous token range [ntok g / G, ntok (g+1) / G) -- balanced to one
// token (a 16-token tile grid left 2 or 3 tiles per workgroup at the benchmark shape: 15 % idle).
DEVFN long range_lo(long ntok) { return ntok * blockIdx.x / gridDim.x; }
DEVFN long range_hi(long ntok) { return ntok * (blockIdx.x + 1) / gridDim.x; }
// kva_bwd / post_bwd (token-local arithmetic): tokens blockIdx.x, blockIdx.x + gridDim.x, ... instead of a contiguous range
#ifndef VRWKV_BWD_STRIDED
#define VRWKV_BWD_STRIDED 1
#endif
DEVFN long tok_first(long ntok) { return VRWKV_BWD_STRIDED ? (long)blockIdx.x : range_lo(ntok); }
DEVFN long tok_end(long ntok) { return VRWKV_BWD_STRIDED ? ntok : range_hi(ntok); }
DEVFN long tok_step() { return VRWKV_BWD_STRIDED ? (long)gridDim.x : 1L; }
DEVFN float sigmoidf_(float x) { return 1.f / (1.f + fast_exp(-x)); }

struct Ptrs6 { const uint16_t* p[MAXM]; };
struct MPtrs6 { uint16_t* p[MAXM]; };

// ---------------------------------------------------------------------------------------------- F1 / F5: shift + lerps
// DD (RWKV-6, VisualRWKV-v6/v6.0/src/model.py:150-160): the lerp weight of output j is mu_j + mm_j[n] with a per-token
// tensor mm_j (the 5-way data-dependent LoRA of the time-mix) instead of the channel vector alone.
template <int M, bool DD = false>
__global__ void mix_fwd_kernel(long ntok, int T, int C, const uint16_t* __restrict__ x, const uint16_t* __restrict__ x_prev,
                               Ptrs6 mu, MPtrs6 out, Ptrs6 mm = Ptrs6{}) {
    const int c0 = threadIdx.x * 8;
    V8 m[M];
#pragma unroll
    for (int i = 0; i < M; ++i) m[i] = ld8f(mu.p[i] + c0);
    for (long n = own_lo(ntok), n_end = own_hi(ntok); n < n_end; ++n) {
        const V8 xv = ld8f(x + n * C + c0);
        V8 xx;
        if (n % T != 0) {
            const V8 xp = ld8f(x + (n - 1) * C + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) xx.f[e] = xp.f[e] - xv.f[e];
        } else if (x_prev) {                       // stateful inference: the token before the first one of sample n / T
            const V8 xp = ld8f(x_prev + (n / T) * C + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) xx.f[e] = xp.f[e] - xv.f[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xx.f[e] = -xv.f[e];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            V8 o, w = m[j];
            if (DD) {
                const V8 t = ld8f(mm.p[j] + n * C + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) w.f[e] += t.f[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o.f[e] = fmaf(xx.f[e], w.f[e], xv.f[e]);
            st8f(out.p[j] + n * C + c0, o);
        }
    }
}

// dx[n] = A[n] + [t < T-1] Bv[n+1],  A[n] = sum_m d_m[n] (1 - mu_m) = Dsum[n] - Bv[n],  Bv[n] = sum_m d_m[n] mu_m ;
// dmu_m += d_m[n] * (x[n-1] - x[n]).
// A workgroup owns one contiguous token range (balanced to +-1 token over the grid); a thread owns 4 channels and
// walks the range once: every row of x and of the M gradients is read exactly once (plus one boundary row per
// range), and the only loop-carried values are A[n-1] and x[n-1] (8 registers) -- the loads of later rows do not
// depend on them, so the unrolled loop keeps many rows in flight.  (Reading row n+1 again for every token, as the
// first version did, doubled the L2->CU traffic and needed 200+ VGPRs: 1.9 ms per call instead of ~0.4 ms.)
struct V4 { float f[4]; };
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
DEVFN V4 ld4f(const uint16_t* p) {
#if VRWKV_NT
    const u32x2_t u = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
#else
    const u32x2_t u = *reinterpret_cast<const u32x2_t*>(p);
#endif
    V4 r;
    r.f[0] = bf16_lo(u[0]); r.f[1] = bf16_hi(u[0]); r.f[2] = bf16_lo(u[1]); r.f[3] = bf16_hi(u[1]);
    return r;
}
DEVFN void st4f(uint16_t* p, const V4& v) {
    const u32x2_t u = {cvt_pk_bf16(v.f[0], v.f[1]), cvt_pk_bf16(v.f[2], v.f[3])};
#if VRWKV_NT
    __builtin_nontemporal_store(u, reinterpret_cast<u32x2_t*>(p));
#else
    *reinterpret_cast<u32x2_t*>(p) = u;
#endif
}
// DUP3: output 3 (x_v of the time-mix) has two consumers (value projection, v-gate LoRA); their gradients arrive as
// dout.p[3] and dout3b and are summed here instead of by a separate element-wise kernel (3 x 172 MB per layer).
// DD: per-token lerp weights mu_j + mm_j[n]; additionally writes dmm_j[n] = d_j[n] (x[n-1] - x[n]).
// LNX: x is not stored -- it is the LayerNorm output of the fused add + LayerNorm + lerps forward (ln_fused.hip: ln_mix_fwd_kernel),
// recomputed here from xn and the row statistics exactly as that kernel rounded it: bf16(fma((xn - mean) rstd, gamma, beta)).
struct LnX { const float* mean; const float* rstd; const uint16_t* w; const uint16_t* b; };
template <int M, bool DUP3, bool DD = false, bool LNX = false>
__global__ __launch_bounds__(512) void mix_bwd_kernel(long ntok, int T, int C, const uint16_t* __restrict__ x, Ptrs6 mu, Ptrs6 dout,
                                                      const uint16_t* __restrict__ dout3b, uint16_t* __restrict__ dx,
                                                      float* __restrict__ dmu, Ptrs6 mm = Ptrs6{}, MPtrs6 dmm = MPtrs6{}, LnX ln = LnX{}) {
    const long lo = ntok * blockIdx.x / gridDim.x, hi = ntok * (blockIdx.x + 1) / gridDim.x;
    for (int c0 = threadIdx.x * 4; c0 < C; c0 += blockDim.x * 4) {
        uint2 lnw = make_uint2(0u, 0u), lnb = lnw;
        if (LNX) { lnw = *reinterpret_cast<const uint2*>(ln.w + c0); lnb = *reinterpret_cast<const uint2*>(ln.b + c0); }
        auto ldx = [&](long row) {                           // row `row` of x, 4 channels
            V4 t = ld4f(x + row * C + c0);
            if (LNX) {
                const float m0 = ln.mean[row], r0 = ln.rstd[row];
                const float wf[4] = {bf16_lo(lnw.x), bf16_hi(lnw.x), bf16_lo(lnw.y), bf16_hi(lnw.y)};
                const float bf[4] = {bf16_lo(lnb.x), bf16_hi(lnb.x), bf16_lo(lnb.y), bf16_hi(lnb.y)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf((t.f[e] - m0) * r0, wf[e], bf[e]);
                const uint32_t p0 = cvt_pk_bf16(o[0], o[1]), p1 = cvt_pk_bf16(o[2], o[3]);
                t.f[0] = bf16_lo(p0); t.f[1] = bf16_hi(p0); t.f[2] = bf16_lo(p1); t.f[3] = bf16_hi(p1);
            }
            return t;
        };
        V4 m[M], gm[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            m[i] = ld4f(mu.p[i] + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) gm[i].f[e] = 0.f;
        }
        V4 xprev, aprev;
        {
            const V4 t = ldx(lo > 0 ? lo - 1 : 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xprev.f[e] = t.f[e]; aprev.f[e] = 0.f; }
        }
        // rows lo .. hi-1 in full; row hi (if it continues the last sequence) contributes only Bv to dx[hi-1]
        const long last = (hi < ntok && hi % T != 0) ? hi : hi - 1;
#pragma unroll 4
        for (long n = lo; n <= last; ++n) {
            const bool inside = n < hi;
            const bool cont = n % T != 0;                   // row n-1 belongs to the same sequence
            V4 d[M];
#pragma unroll
            for (int j = 0; j < M; ++j) d[j] = ld4f(dout.p[j] + n * C + c0);
            if (DUP3) {
                const V4 d2 = ld4f(dout3b + n * C + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[M > 3 ? 3 : 0].f[e] += d2.f[e];
            }
            const V4 xv = ldx(inside ? n : n - 1);
            V4 dsum, bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { dsum.f[e] = 0.f; bv.f[e] = 0.f; }
#pragma unroll
            for (int j = 0; j < M; ++j) {
                V4 w = m[j];
                if (DD) {
                    const V4 t = ld4f(mm.p[j] + n * C + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) w.f[e] += t.f[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { dsum.f[e] += d[j].f[e]; bv.f[e] = fmaf(d[j].f[e], w.f[e], bv.f[e]); }
            }
            if (n > lo) {
                V4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.f[e] = aprev.f[e] + (cont ? bv.f[e] : 0.f);
                st4f(dx + (n - 1) * C + c0, o);
            }
            if (inside) {
                V4 xxv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xx = (cont ? xprev.f[e] : 0.f) - xv.f[e];
                    xxv.f[e] = xx;
#pragma unroll
                    for (int j = 0; j < M; ++j) gm[j].f[e] = fmaf(d[j].f[e], xx, gm[j].f[e]);
                    aprev.f[e] = dsum.f[e] - bv.f[e];
                    xprev.f[e] = xv.f[e];
                }
                if (DD) {
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        V4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.f[e] = d[j].f[e] * xxv.f[e];
                        st4f(dmm.p[j] + n * C + c0, o);
                    }
                }
            }
        }
        if (last == hi - 1 && hi > lo) st4f(dx + (hi - 1) * C + c0, aprev);      // no successor row: dx = A
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float* dst = dmu + ((size_t)blockIdx.x * M + j) * C + c0;
            *reinterpret_cast<float4*>(dst) = make_float4(gm[j].f[0], gm[j].f[1], gm[j].f[2], gm[j].f[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------- F2: decay soft-clamp
// w = -softplus(-(w0 + h)) - 0.5
__global__ void decay_fwd_kernel(long ntok, int C, const uint16_t* __restrict__ h, const uint16_t* __restrict__ w0,
                                 uint16_t* __restrict__ w) {
    const int c0 = threadIdx.x * 8;
    const V8 b = ld8f(w0 + c0);
    for (long n = own_lo(ntok), n_end = own_hi(ntok); n < n_end; ++n) {
        const V8 hv = ld8f(h + n * C + c0);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = hv.f[e] + b.f[e];
            const float sp = fmaxf(-u, 0.f) + fast_log(1.f + fast_exp(-fabsf(u)));   // softplus(-u)
            o.f[e] = -sp - 0.5f;
        }
        st8f(w + n * C + c0, o);
    }
}
__global__ void decay_bwd_kernel(long ntok, int C, const uint16_t* __restrict__ h, const uint16_t* __restrict__ w0,
                                 const uint16_t* __restrict__ dw, uint16_t* __restrict__ dh, float* __restrict__ dw0) {
    const int c0 = threadIdx.x * 8;
    const V8 b = ld8f(w0 + c0);
    V8 g0 = zero8();
    for (long n = range_lo(ntok), hi = range_hi(ntok); n < hi; ++n) {
        const V8 hv = ld8f(h + n * C + c0), d = ld8f(dw + n * C + c0);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o.f[e] = d.f[e] * sigmoidf_(-(hv.f[e] + b.f[e]));
            g0.f[e] += o.f[e];
        }
        st8f(dh + n * C + c0, o);
    }
    put_partial(dw0, 1, 0, C, c0, g0);
}

// ---------------------------------------------------------------------------------------------- F3: k / v / a glue
struct KvaFwd {
    long ntok; int C; int has_vres;
    const uint16_t *k, *v, *vfirst, *vl, *al;          // activations
    const uint16_t *k_k, *k_a, *a0, *v0;               // parameters (C)
    uint16_t *k2, *v2, *z, *b;                         // outputs
};
__global__ void kva_fwd_kernel(KvaFwd p) {
    const int c0 = threadIdx.x * 8, C = p.C;
    const V8 kk_p = ld8f(p.k_k + c0), ka_p = ld8f(p.k_a + c0), a0 = ld8f(p.a0 + c0);
    V8 v0 = zero8();
    if (p.has_vres) v0 = ld8f(p.v0 + c0);
    for (long n = own_lo(p.ntok), n_end = own_hi(p.ntok); n < n_end; ++n) {
        const long o = n * C + c0;
        const V8 k = ld8f(p.k + o), al = ld8f(p.al + o);
        V8 a, kk, k2, z, b;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a.f[e] = sigmoidf_(a0.f[e] + al.f[e]);
            kk.f[e] = k.f[e] * kk_p.f[e];
            ss = fmaf(kk.f[e], kk.f[e], ss);
            k2.f[e] = k.f[e] * (1.f + (a.f[e] - 1.f) * ka_p.f[e]);
        }
        ss = group_sum<3>(ss);                                   // 64 channels of the head = 8 adjacent lanes
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);       // F.normalize(p=2, eps=1e-12)
#pragma unroll
        for (int e = 0; e < 8; ++e) { kk.f[e] *= inv; z.f[e] = -kk.f[e]; b.f[e] = kk.f[e] * a.f[e]; }
        st8f(p.k2 + o, k2); st8f(p.z + o, z); st8f(p.b + o, b);
        if (p.has_vres) {
            const V8 v = ld8f(p.v + o), vf = ld8f(p.vfirst + o), vl = ld8f(p.vl + o);
            V8 v2;
#pragma unroll
            for (int e = 0; e < 8; ++e) v2.f[e] = v.f[e] + (vf.f[e] - v.f[e]) * sigmoidf_(v0.f[e] + vl.f[e]);
            st8f(p.v2 + o, v2);
        }
    }
}
struct KvaBwd {
    long ntok; int C; int has_vres;
    const uint16_t *k, *v, *vfirst, *vl, *al, *k_k, *k_a, *a0, *v0;
    const uint16_t *dk2, *dv2, *dz, *db;               // incoming
    uint16_t *dk, *dv, *dvfirst, *dvl, *dal;           // outgoing
    float* part;                                       // [grid][4][C] partials of dk_k dk_a da0 dv0
    const uint16_t *dk2b, *dv2b;                       // optional second gradients of k2 / v2 (two consumers: WKV7 and the
                                                       // bonus term of `post`), summed here instead of by autograd
    const uint16_t* dvf_in;                            // optional: gradient of v_first collected by the layers after this one
};
// Two passes over the workgroup's token range -- the k / a-gate part, then the value-residual part: they share nothing but the
// loop, and as one loop the kernel held 4 parameter vectors + 4 gradient accumulators + both parts' rows: 128 VGPRs and 88 B of
// scratch per lane inside the token loop (3.5 TB/s where post_bwd reaches 5.4).  Per pass: 3 + 3 vectors (1 + 1 in the second).
template <int LB>
__global__ __launch_bounds__(LB) void kva_bwd_kernel(KvaBwd p) {
    constexpr bool PF = VRWKV_BWD_PF && LB <= 256;        // wider rows: 8 or 16 waves per workgroup, the 145 registers of the prefetching loop would leave one workgroup per CU
    const int c0 = threadIdx.x * 8, C = p.C;
    const long lo = tok_first(p.ntok), hi = tok_end(p.ntok), step = tok_step();
    {
        const V8 kk_p = ld8f(p.k_k + c0), ka_p = ld8f(p.k_a + c0), a0 = ld8f(p.a0 + c0);
        V8 g_kk = zero8(), g_ka = zero8(), g_a0 = zero8();
        struct Row { u32x4_t k, al, dk2, dz, db, dk2b; };
        auto fetch = [&](long n) {
            const long o = n * C + c0;
            Row r;
            r.k = ld8raw(p.k + o); r.al = ld8raw(p.al + o); r.dk2 = ld8raw(p.dk2 + o); r.dz = ld8raw(p.dz + o); r.db = ld8raw(p.db + o);
            if (p.dk2b) r.dk2b = ld8raw(p.dk2b + o);
            return r;
        };
        Row nxt{};
        if (PF && lo < hi) nxt = fetch(lo);
        for (long n = lo; n < hi; n += step) {
            const long o = n * C + c0;
            Row cur;
            if (PF) { cur = nxt; if (n + step < hi) nxt = fetch(n + step); } else cur = fetch(n);
            const V8 k = cvt8(cur.k), al = cvt8(cur.al);
            V8 dk2 = cvt8(cur.dk2);
            const V8 dz = cvt8(cur.dz), db = cvt8(cur.db);
            if (p.dk2b) {
                const V8 t = cvt8(cur.dk2b);
#pragma unroll
                for (int e = 0; e < 8; ++e) dk2.f[e] += t.f[e];
            }
            V8 a, u, kk, dkk, dk, dal;
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a.f[e] = sigmoidf_(a0.f[e] + al.f[e]);
                u.f[e] = k.f[e] * kk_p.f[e];
                ss = fmaf(u.f[e], u.f[e], ss);
            }
            ss = group_sum<3>(ss);
            const float nrm = fmaxf(sqrtf(ss), 1e-12f), inv = 1.f / nrm;
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kk.f[e] = u.f[e] * inv;
                dkk.f[e] = db.f[e] * a.f[e] - dz.f[e];              // z = -kk, b = kk*a
                dot = fmaf(kk.f[e], dkk.f[e], dot);
            }
            dot = group_sum<3>(dot);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float du = (dkk.f[e] - kk.f[e] * dot) * inv;  // d normalize
                const float mod = 1.f + (a.f[e] - 1.f) * ka_p.f[e];
                dk.f[e] = dk2.f[e] * mod + du * kk_p.f[e];
                g_kk.f[e] = fmaf(du, k.f[e], g_kk.f[e]);
                g_ka.f[e] = fmaf(dk2.f[e] * k.f[e], a.f[e] - 1.f, g_ka.f[e]);
                const float da = db.f[e] * kk.f[e] + dk2.f[e] * k.f[e] * ka_p.f[e];
                dal.f[e] = da * a.f[e] * (1.f - a.f[e]);
                g_a0.f[e] += dal.f[e];
            }
            st8f(p.dk + o, dk); st8f(p.dal + o, dal);
        }
        put_partial(p.part, 4, 0, C, c0, g_kk); put_partial(p.part, 4, 1, C, c0, g_ka); put_partial(p.part, 4, 2, C, c0, g_a0);
    }
    V8 g_v0 = zero8();
    if (p.has_vres) {
        const V8 v0 = ld8f(p.v0 + c0);
        struct Row2 { u32x4_t v, vf, vl, dv2, dv2b, dvf; };
        auto fetch2 = [&](long n) {
            const long o = n * C + c0;
            Row2 r;
            r.v = ld8raw(p.v + o); r.vf = ld8raw(p.vfirst + o); r.vl = ld8raw(p.vl + o); r.dv2 = ld8raw(p.dv2 + o);
            if (p.dv2b) r.dv2b = ld8raw(p.dv2b + o);
            if (p.dvf_in) r.dvf = ld8raw(p.dvf_in + o);
            return r;
        };
        Row2 nxt{};
        if (PF && lo < hi) nxt = fetch2(lo);
        for (long n = lo; n < hi; n += step) {
            const long o = n * C + c0;
            Row2 cur;
            if (PF) { cur = nxt; if (n + step < hi) nxt = fetch2(n + step); } else cur = fetch2(n);
            const V8 v = cvt8(cur.v), vf = cvt8(cur.vf), vl = cvt8(cur.vl);
            V8 dv2 = cvt8(cur.dv2);
            if (p.dv2b) {
                const V8 t = cvt8(cur.dv2b);
#pragma unroll
                for (int e = 0; e < 8; ++e) dv2.f[e] += t.f[e];
            }
            V8 dv, dvf, dvl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = sigmoidf_(v0.f[e] + vl.f[e]);
                dv.f[e] = dv2.f[e] * (1.f - sv);
                dvf.f[e] = dv2.f[e] * sv;
                dvl.f[e] = dv2.f[e] * (vf.f[e] - v.f[e]) * sv * (1.f - sv);
                g_v0.f[e] += dvl.f[e];
            }
            if (p.dvf_in) {                                      // running sum over the layers (autograd would add 3 x 172 MB per layer)
                const V8 t = cvt8(cur.dvf);
#pragma unroll
                for (int e = 0; e < 8; ++e) dvf.f[e] += t.f[e];
            }
            st8f(p.dv + o, dv); st8f(p.dvfirst + o, dvf); st8f(p.dvl + o, dvl);
        }
    }
    put_partial(p.part, 4, 3, C, c0, g_v0);
}

// ---------------------------------------------------------------------------------------------- F4: GroupNorm + bonus + gate
struct PostFwd {
    long ntok; int C; float eps;
    const uint16_t *y, *r, *k, *v, *g, *ln_w, *ln_b, *r_k;
    uint16_t* out;
};
__global__ void post_fwd_kernel(PostFwd p) {
    const int c0 = threadIdx.x * 8, C = p.C;
    const V8 lw = ld8f(p.ln_w + c0), lb = ld8f(p.ln_b + c0), rk = ld8f(p.r_k + c0);
    for (long n = own_lo(p.ntok), n_end = own_hi(p.ntok); n < n_end; ++n) {
        const long o = n * C + c0;
        const V8 y = ld8f(p.y + o), r = ld8f(p.r + o), k = ld8f(p.k + o), v = ld8f(p.v + o), g = ld8f(p.g + o);
        float s1 = 0.f, sb = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += y.f[e]; sb = fmaf(r.f[e] * k.f[e], rk.f[e], sb); }
        s1 = group_sum<3>(s1); sb = group_sum<3>(sb);
        const float mean = s1 * (1.f / 64.f);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = y.f[e] - mean; s2 = fmaf(d, d, s2); }
        s2 = group_sum<3>(s2);
        const float rstd = fast_rsqrt(s2 * (1.f / 64.f) + p.eps);
        V8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gn = (y.f[e] - mean) * rstd * lw.f[e] + lb.f[e];
            out.f[e] = (gn + sb * v.f[e]) * g.f[e];
        }
        st8f(p.out + o, out);
    }
}
// RWKV-6 (VisualRWKV-v6/v6.0/src/model.py:176-184, jit_func_2 and the silu of :166): out = GroupNorm(H, C, eps)(y) * silu(gg)
__global__ void gn_silu_fwd_kernel(long ntok, int C, float eps, const uint16_t* __restrict__ yp, const uint16_t* __restrict__ ggp,
                                   const uint16_t* __restrict__ ln_w, const uint16_t* __restrict__ ln_b, uint16_t* __restrict__ outp) {
    const int c0 = threadIdx.x * 8;
    const V8 lw = ld8f(ln_w + c0), lb = ld8f(ln_b + c0);
    for (long n = own_lo(ntok), n_end = own_hi(ntok); n < n_end; ++n) {
        const long o = n * C + c0;
        const V8 y = ld8f(yp + o), gg = ld8f(ggp + o);
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += y.f[e];
        const float mean = group_sum<3>(s1) * (1.f / 64.f);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = y.f[e] - mean; s2 = fmaf(d, d, s2); }
        const float rstd = fast_rsqrt(group_sum<3>(s2) * (1.f / 64.f) + eps);
        V8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) out.f[e] = ((y.f[e] - mean) * rstd * lw.f[e] + lb.f[e]) * (gg.f[e] * sigmoidf_(gg.f[e]));
        st8f(outp + o, out);
    }
}
__global__ void gn_silu_bwd_kernel(long ntok, int C, float eps, const uint16_t* __restrict__ yp, const uint16_t* __restrict__ ggp,
                                   const uint16_t* __restrict__ ln_w, const uint16_t* __restrict__ ln_b, const uint16_t* __restrict__ doutp,
                                   uint16_t* __restrict__ dyp, uint16_t* __restrict__ dggp, float* __restrict__ part) {
    const int c0 = threadIdx.x * 8;
    const V8 lw = ld8f(ln_w + c0), lb = ld8f(ln_b + c0);
    V8 g_w = zero8(), g_b = zero8();
    for (long n = range_lo(ntok), hi = range_hi(ntok); n < hi; ++n) {
        const long o = n * C + c0;
        const V8 y = ld8f(yp + o), gg = ld8f(ggp + o), d = ld8f(doutp + o);
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += y.f[e];
        const float mean = group_sum<3>(s1) * (1.f / 64.f);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dd = y.f[e] - mean; s2 = fmaf(dd, dd, s2); }
        const float rstd = fast_rsqrt(group_sum<3>(s2) * (1.f / 64.f) + eps);
        V8 yn, dyn, dgg;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            yn.f[e] = (y.f[e] - mean) * rstd;
            const float gn = yn.f[e] * lw.f[e] + lb.f[e];
            const float sg = sigmoidf_(gg.f[e]), silu = gg.f[e] * sg;
            dgg.f[e] = d.f[e] * gn * (sg * (1.f + gg.f[e] * (1.f - sg)));      // d silu / d gg
            const float dt = d.f[e] * silu;
            g_w.f[e] = fmaf(dt, yn.f[e], g_w.f[e]);
            g_b.f[e] += dt;
            dyn.f[e] = dt * lw.f[e];
            m1 += dyn.f[e];
            m2 = fmaf(dyn.f[e], yn.f[e], m2);
        }
        m1 = group_sum<3>(m1) * (1.f / 64.f); m2 = group_sum<3>(m2) * (1.f / 64.f);
        V8 dy;
#pragma unroll
        for (int e = 0; e < 8; ++e) dy.f[e] = rstd * (dyn.f[e] - m1 - yn.f[e] * m2);
        st8f(dyp + o, dy); st8f(dggp + o, dgg);
    }
    put_partial(part, 2, 0, C, c0, g_w); put_partial(part, 2, 1, C, c0, g_b);
}

struct PostBwd {
    long ntok; int C; float eps;
    const uint16_t *y, *r, *k, *v, *g, *ln_w, *ln_b, *r_k, *dout;
    uint16_t *dy, *dr, *dk, *dv, *dg;
    float* part;                                       // [grid][3][C] partials of dln_w dln_b dr_k
};
template <int LB>
__global__ __launch_bounds__(LB) void post_bwd_kernel(PostBwd p) {
    constexpr bool PF = VRWKV_BWD_PF && LB <= 256;
    const int c0 = threadIdx.x * 8, C = p.C;
    const V8 lw = ld8f(p.ln_w + c0), lb = ld8f(p.ln_b + c0), rk = ld8f(p.r_k + c0);
    V8 g_w = zero8(), g_b = zero8(), g_rk = zero8();
    struct Row { u32x4_t y, r, k, v, g, d; };
    auto fetch = [&](long n) {
        const long o = n * C + c0;
        Row q;
        q.y = ld8raw(p.y + o); q.r = ld8raw(p.r + o); q.k = ld8raw(p.k + o); q.v = ld8raw(p.v + o); q.g = ld8raw(p.g + o); q.d = ld8raw(p.dout + o);
        return q;
    };
    const long lo = tok_first(p.ntok), hi = tok_end(p.ntok), step = tok_step();
    Row nxt{};
    if (PF && lo < hi) nxt = fetch(lo);
    for (long n = lo; n < hi; n += step) {
        const long o = n * C + c0;
        Row cur;
        if (PF) { cur = nxt; if (n + step < hi) nxt = fetch(n + step); } else cur = fetch(n);
        const V8 y = cvt8(cur.y), r = cvt8(cur.r), k = cvt8(cur.k), v = cvt8(cur.v), g = cvt8(cur.g);
        const V8 d = cvt8(cur.d);
        float s1 = 0.f, sb = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += y.f[e]; sb = fmaf(r.f[e] * k.f[e], rk.f[e], sb); }
        s1 = group_sum<3>(s1); sb = group_sum<3>(sb);
        const float mean = s1 * (1.f / 64.f);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dd = y.f[e] - mean; s2 = fmaf(dd, dd, s2); }
        s2 = group_sum<3>(s2);
        const float rstd = fast_rsqrt(s2 * (1.f / 64.f) + p.eps);
        V8 yn, dt, dyn, dg;
        float m1 = 0.f, m2 = 0.f, ds = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            yn.f[e] = (y.f[e] - mean) * rstd;
            const float gn = yn.f[e] * lw.f[e] + lb.f[e];
            dg.f[e] = d.f[e] * (gn + sb * v.f[e]);
            dt.f[e] = d.f[e] * g.f[e];
            g_w.f[e] = fmaf(dt.f[e], yn.f[e], g_w.f[e]);
            g_b.f[e] += dt.f[e];
            dyn.f[e] = dt.f[e] * lw.f[e];
            m1 += dyn.f[e];
            m2 = fmaf(dyn.f[e], yn.f[e], m2);
            ds = fmaf(dt.f[e], v.f[e], ds);
        }
        m1 = group_sum<3>(m1) * (1.f / 64.f); m2 = group_sum<3>(m2) * (1.f / 64.f); ds = group_sum<3>(ds);
        V8 dy, dr, dk, dv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dy.f[e] = rstd * (dyn.f[e] - m1 - yn.f[e] * m2);
            dr.f[e] = ds * k.f[e] * rk.f[e];
            dk.f[e] = ds * r.f[e] * rk.f[e];
            dv.f[e] = dt.f[e] * sb;
            g_rk.f[e] = fmaf(ds, r.f[e] * k.f[e], g_rk.f[e]);
        }
        st8f(p.dy + o, dy); st8f(p.dr + o, dr); st8f(p.dk + o, dk); st8f(p.dv + o, dv); st8f(p.dg + o, dg);
    }
    put_partial(p.part, 3, 0, C, c0, g_w); put_partial(p.part, 3, 1, C, c0, g_b); put_partial(p.part, 3, 2, C, c0, g_rk);
}

// ---------------------------------------------------------------------------------------------- F6: relu^2
// One 16-byte vector per thread, no loop, non-temporal accesses: the same arithmetic as a grid-stride loop over <= 4096 workgroups
// ran at 4.8-5.1 TB/s, this form at 6.5 (benchmarks/eltwise_probe.hip, profiles/r4_eltwise_probe.jsonl) -- a wave that issues
// its loads, its store and ends keeps more requests in flight per CU than a resident wave that waits for its own store every
// iteration.
__global__ __launch_bounds__(256) void relusq_fwd_kernel(long n8, const uint16_t* __restrict__ h, uint16_t* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    V8 v = ld8f(h + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float r = fmaxf(v.f[e], 0.f); v.f[e] = r * r; }
    st8f(y + i * 8, v);
}
__global__ __launch_bounds__(256) void relusq_bwd_kernel(long n8, const uint16_t* __restrict__ h, const uint16_t* __restrict__ dy,
                                                         uint16_t* __restrict__ dh) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const V8 v = ld8f(h + i * 8), d = ld8f(dy + i * 8);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.f[e] = 2.f * fmaxf(v.f[e], 0.f) * d.f[e];
    st8f(dh + i * 8, o);
}

inline int ok_c(int C) { return C > 0 && C % 64 == 0 && C / 8 <= 1024; }
inline dim3 tok_grid(long ntok, int tpb) { return dim3((unsigned)((ntok + tpb - 1) / tpb)); }
#ifndef VRWKV_BWD_GRID
#define VRWKV_BWD_GRID 1024
#endif
constexpr int BWD_GRID = VRWKV_BWD_GRID;         // workgroups (= partial rows) of the backward kernels: 4 per CU
inline int bwd_grid(long ntok) { long g = (ntok + TPB - 1) / TPB; return (int)(g < BWD_GRID ? g : BWD_GRID); }
// kva_bwd / post_bwd with the prefetched row: 138-145 registers, three workgroups of 256 threads per CU -> one round of 768
#ifndef VRWKV_BWD_GRID_PF
#define VRWKV_BWD_GRID_PF 768
#endif
inline int bwd_grid_pf(long ntok, int C) {
    const int g = bwd_grid(ntok), cap = (VRWKV_BWD_PF && C / 8 <= 256) ? VRWKV_BWD_GRID_PF : BWD_GRID;
    return g < cap ? g : cap;
}
inline void colsum(int G, long width, const float* part, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)(width / 16)), dim3(256), 0, st, G, width, part, out);
}
inline int done() { hipError_t e = hipGetLastError(); return e == hipSuccess ? VRWKV_OK : (int)e; }

}  // namespace

extern "C" {

int vrwkv_mix_fwd_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, void* const* out, void* stream) {
    return vrwkv_mix_fwd_prev_bf16(ntok, T, C, M, x, nullptr, mu, out, stream);
}

int vrwkv_mix_fwd_prev_bf16(long ntok, int T, int C, int M, const void* x, const void* x_prev, const void* const* mu,
                            void* const* out, void* stream) {
    if (ntok <= 0 || T <= 0 || !x || !mu || !out || (M != 1 && M != 2 && M != 6)) return VRWKV_EINVAL;
    if (!ok_c(C) || ntok % T != 0) return VRWKV_ESHAPE;
    Ptrs6 m{}; MPtrs6 o{};
    for (int i = 0; i < M; ++i) { m.p[i] = (const uint16_t*)mu[i]; o.p[i] = (uint16_t*)out[i]; if (!m.p[i] || !o.p[i]) return VRWKV_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (M == 6) hipLaunchKernelGGL(mix_fwd_kernel<6>, tok_grid(ntok, TPB_MIX), dim3(C / 8), 0, st, ntok, T, C, (const uint16_t*)x, (const uint16_t*)x_prev, m, o, Ptrs6{});
    else if (M == 2) hipLaunchKernelGGL(mix_fwd_kernel<2>, tok_grid(ntok, TPB_MIX), dim3(C / 8), 0, st, ntok, T, C, (const uint16_t*)x, (const uint16_t*)x_prev, m, o, Ptrs6{});
    else hipLaunchKernelGGL(mix_fwd_kernel<1>, tok_grid(ntok, TPB_MIX), dim3(C / 8), 0, st, ntok, T, C, (const uint16_t*)x, (const uint16_t*)x_prev, m, o, Ptrs6{});
    return done();
}

long vrwkv_param_grad_ws_floats(long ntok, int C, int nvec) { return (long)bwd_grid(ntok) * nvec * C; }

int vrwkv_mix_bwd_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, const void* const* dout,
                       void* dx, float* dmu, float* ws, void* stream) {
    return vrwkv_mix_bwd2_bf16(ntok, T, C, M, x, mu, dout, nullptr, dx, dmu, ws, stream);
}

int vrwkv_mix_bwd2_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, const void* const* dout,
                        const void* dout3_second, void* dx, float* dmu, float* ws, void* stream) {
    if (ntok <= 0 || T <= 0 || !x || !mu || !dout || !dx || !dmu || !ws || (M != 1 && M != 2 && M != 6)) return VRWKV_EINVAL;
    if (dout3_second && M != 6) return VRWKV_EINVAL;
    if (!ok_c(C) || ntok % T != 0) return VRWKV_ESHAPE;
    Ptrs6 m{}, d{};
    for (int i = 0; i < M; ++i) { m.p[i] = (const uint16_t*)mu[i]; d.p[i] = (const uint16_t*)dout[i]; if (!m.p[i] || !d.p[i]) return VRWKV_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int G = bwd_grid(ntok);
    const int threads = C / 4 < 512 ? C / 4 : 512;
    const uint16_t* d2 = (const uint16_t*)dout3_second;
    if (M == 6 && d2) hipLaunchKernelGGL((mix_bwd_kernel<6, true>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)x, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{});
    else if (M == 6) hipLaunchKernelGGL((mix_bwd_kernel<6, false>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)x, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{});
    else if (M == 2) hipLaunchKernelGGL((mix_bwd_kernel<2, false>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)x, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{});
    else hipLaunchKernelGGL((mix_bwd_kernel<1, false>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)x, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{});
    colsum(G, (long)M * C, ws, dmu, st);
    return done();
}

// mix backward when x is the (unstored) LayerNorm output of vrwkv_ln_mix_fwd_bf16: recomputed from xn, mean, rstd, ln_w, ln_b
int vrwkv_mix_bwd_ln_bf16(long ntok, int T, int C, int M, const void* xn, const float* mean, const float* rstd, const void* ln_w,
                          const void* ln_b, const void* const* mu, const void* const* dout, const void* dout3_second, void* dx,
                          float* dmu, float* ws, void* stream) {
    if (ntok <= 0 || T <= 0 || !xn || !mean || !rstd || !ln_w || !ln_b || !mu || !dout || !dx || !dmu || !ws || M != 6) return VRWKV_EINVAL;
    if (!ok_c(C) || ntok % T != 0) return VRWKV_ESHAPE;
    Ptrs6 m{}, d{};
    for (int i = 0; i < M; ++i) { m.p[i] = (const uint16_t*)mu[i]; d.p[i] = (const uint16_t*)dout[i]; if (!m.p[i] || !d.p[i]) return VRWKV_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int G = bwd_grid(ntok);
    const int threads = C / 4 < 512 ? C / 4 : 512;
    const uint16_t* d2 = (const uint16_t*)dout3_second;
    const LnX ln{mean, rstd, (const uint16_t*)ln_w, (const uint16_t*)ln_b};
    if (d2) hipLaunchKernelGGL((mix_bwd_kernel<6, true, false, true>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)xn, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{}, ln);
    else hipLaunchKernelGGL((mix_bwd_kernel<6, false, false, true>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)xn, m, d, d2, (uint16_t*)dx, ws, Ptrs6{}, MPtrs6{}, ln);
    colsum(G, (long)M * C, ws, dmu, st);
    return done();
}

// RWKV-6 data-dependent token shift (VisualRWKV-v6/v6.0/src/model.py:150-160): out_j = x + (x[t-1] - x) (mu_j + mm_j), j < 5.
int vrwkv_ddmix_fwd_bf16(long ntok, int T, int C, const void* x, const void* const* mu, const void* const* mm, void* const* out, void* stream) {
    if (ntok <= 0 || T <= 0 || !x || !mu || !mm || !out) return VRWKV_EINVAL;
    if (!ok_c(C) || ntok % T != 0) return VRWKV_ESHAPE;
    Ptrs6 m{}, t{}; MPtrs6 o{};
    for (int i = 0; i < 5; ++i) {
        m.p[i] = (const uint16_t*)mu[i]; t.p[i] = (const uint16_t*)mm[i]; o.p[i] = (uint16_t*)out[i];
        if (!m.p[i] || !t.p[i] || !o.p[i]) return VRWKV_EINVAL;
    }
    hipLaunchKernelGGL((mix_fwd_kernel<5, true>), tok_grid(ntok, TPB_MIX), dim3(C / 8), 0, (hipStream_t)stream, ntok, T, C, (const uint16_t*)x,
                       (const uint16_t*)nullptr, m, o, t);
    return done();
}
// backward: dx, dmm_j (ntok, C) bf16 each, dmu = 5*C floats; ws: vrwkv_param_grad_ws_floats(ntok, C, 5)
int vrwkv_ddmix_bwd_bf16(long ntok, int T, int C, const void* x, const void* const* mu, const void* const* mm, const void* const* dout,
                         void* dx, void* const* dmm, float* dmu, float* ws, void* stream) {
    if (ntok <= 0 || T <= 0 || !x || !mu || !mm || !dout || !dx || !dmm || !dmu || !ws) return VRWKV_EINVAL;
    if (!ok_c(C) || ntok % T != 0) return VRWKV_ESHAPE;
    Ptrs6 m{}, t{}, d{}; MPtrs6 o{};
    for (int i = 0; i < 5; ++i) {
        m.p[i] = (const uint16_t*)mu[i]; t.p[i] = (const uint16_t*)mm[i]; d.p[i] = (const uint16_t*)dout[i]; o.p[i] = (uint16_t*)dmm[i];
        if (!m.p[i] || !t.p[i] || !d.p[i] || !o.p[i]) return VRWKV_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int G = bwd_grid(ntok);
    const int threads = C / 4 < 512 ? C / 4 : 512;
    hipLaunchKernelGGL((mix_bwd_kernel<5, false, true>), dim3(G), dim3(threads), 0, st, ntok, T, C, (const uint16_t*)x, m, d, (const uint16_t*)nullptr,
                       (uint16_t*)dx, ws, t, o);
    colsum(G, 5L * C, ws, dmu, st);
    return done();
}

// RWKV-6 output stage: out = GroupNorm(C/64 groups, eps)(y) * silu(gg)   (model.py:166,176-184)
int vrwkv_gn_silu_fwd_bf16(long ntok, int C, float eps, const void* y, const void* gg, const void* ln_w, const void* ln_b, void* out, void* stream) {
    if (ntok <= 0 || !y || !gg || !ln_w || !ln_b || !out) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(gn_silu_fwd_kernel, tok_grid(ntok, TPB_GN), dim3(C / 8), 0, (hipStream_t)stream, ntok, C, eps, (const uint16_t*)y, (const uint16_t*)gg,
                       (const uint16_t*)ln_w, (const uint16_t*)ln_b, (uint16_t*)out);
    return done();
}
// dparams = [dln_w | dln_b] (2 C floats); ws: vrwkv_param_grad_ws_floats(ntok, C, 2)
int vrwkv_gn_silu_bwd_bf16(long ntok, int C, float eps, const void* y, const void* gg, const void* ln_w, const void* ln_b, const void* dout,
                           void* dy, void* dgg, float* dparams, float* ws, void* stream) {
    if (ntok <= 0 || !y || !gg || !ln_w || !ln_b || !dout || !dy || !dgg || !dparams || !ws) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    const int G = bwd_grid(ntok);
    hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, ntok, C, eps, (const uint16_t*)y, (const uint16_t*)gg,
                       (const uint16_t*)ln_w, (const uint16_t*)ln_b, (const uint16_t*)dout, (uint16_t*)dy, (uint16_t*)dgg, ws);
    colsum(G, 2L * C, ws, dparams, (hipStream_t)stream);
    return done();
}

int vrwkv_decay_fwd_bf16(long ntok, int C, const void* h, const void* w0, void* w, void* stream) {
    if (ntok <= 0 || !h || !w0 || !w) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(decay_fwd_kernel, tok_grid(ntok, TPB_DECAY), dim3(C / 8), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)h, (const uint16_t*)w0, (uint16_t*)w);
    return done();
}
int vrwkv_decay_bwd_bf16(long ntok, int C, const void* h, const void* w0, const void* dw, void* dh, float* dw0, float* ws, void* stream) {
    if (ntok <= 0 || !h || !w0 || !dw || !dh || !dw0 || !ws) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    const int G = bwd_grid(ntok);
    hipLaunchKernelGGL(decay_bwd_kernel, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, ntok, C, (const uint16_t*)h, (const uint16_t*)w0,
                       (const uint16_t*)dw, (uint16_t*)dh, ws);
    colsum(G, C, ws, dw0, (hipStream_t)stream);
    return done();
}

int vrwkv_kva_fwd_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl, const void* al,
                       const void* k_k, const void* k_a, const void* a0, const void* v0,
                       void* k2, void* v2, void* z, void* b, void* stream) {
    if (ntok <= 0 || !k || !al || !k_k || !k_a || !a0 || !k2 || !z || !b) return VRWKV_EINVAL;
    if (has_vres && (!v || !vfirst || !vl || !v0 || !v2)) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    KvaFwd p{ntok, C, has_vres, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)vfirst, (const uint16_t*)vl, (const uint16_t*)al,
             (const uint16_t*)k_k, (const uint16_t*)k_a, (const uint16_t*)a0, (const uint16_t*)v0,
             (uint16_t*)k2, (uint16_t*)v2, (uint16_t*)z, (uint16_t*)b};
    hipLaunchKernelGGL(kva_fwd_kernel, tok_grid(ntok, TPB_KVA), dim3(C / 8), 0, (hipStream_t)stream, p);
    return done();
}
int vrwkv_kva_bwd_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl, const void* al,
                       const void* k_k, const void* k_a, const void* a0, const void* v0,
                       const void* dk2, const void* dv2, const void* dz, const void* db,
                       void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                       float* dparams, float* ws, void* stream) {
    return vrwkv_kva_bwd2_bf16(ntok, C, has_vres, k, v, vfirst, vl, al, k_k, k_a, a0, v0, dk2, dv2, dz, db, nullptr, nullptr,
                               dk, dv, dvfirst, dvl, dal, dparams, ws, stream);
}
int vrwkv_kva_bwd2_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl, const void* al,
                        const void* k_k, const void* k_a, const void* a0, const void* v0,
                        const void* dk2, const void* dv2, const void* dz, const void* db, const void* dk2_second, const void* dv2_second,
                        void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                        float* dparams, float* ws, void* stream) {
    return vrwkv_kva_bwd3_bf16(ntok, C, has_vres, k, v, vfirst, vl, al, k_k, k_a, a0, v0, dk2, dv2, dz, db, dk2_second, dv2_second, nullptr,
                               dk, dv, dvfirst, dvl, dal, dparams, ws, stream);
}
// dvfirst_in (optional, has_vres only): added to dvfirst -- the layers hand the gradient of v_first down a chain instead of each
// returning its own term for autograd to sum
int vrwkv_kva_bwd3_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl, const void* al,
                        const void* k_k, const void* k_a, const void* a0, const void* v0,
                        const void* dk2, const void* dv2, const void* dz, const void* db, const void* dk2_second, const void* dv2_second,
                        const void* dvfirst_in, void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                        float* dparams, float* ws, void* stream) {
    if ((dv2_second || dvfirst_in) && !has_vres) return VRWKV_EINVAL;
    if (ntok <= 0 || !k || !al || !k_k || !k_a || !a0 || !dk2 || !dz || !db || !dk || !dal || !dparams || !ws) return VRWKV_EINVAL;
    if (has_vres && (!v || !vfirst || !vl || !v0 || !dv2 || !dv || !dvfirst || !dvl)) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    KvaBwd p{ntok, C, has_vres, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)vfirst, (const uint16_t*)vl, (const uint16_t*)al,
             (const uint16_t*)k_k, (const uint16_t*)k_a, (const uint16_t*)a0, (const uint16_t*)v0,
             (const uint16_t*)dk2, (const uint16_t*)dv2, (const uint16_t*)dz, (const uint16_t*)db,
             (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dvfirst, (uint16_t*)dvl, (uint16_t*)dal, ws,
             (const uint16_t*)dk2_second, (const uint16_t*)dv2_second, (const uint16_t*)dvfirst_in};
    const int G = bwd_grid_pf(ntok, C);
    if (C / 8 <= 256) hipLaunchKernelGGL(kva_bwd_kernel<256>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    else if (C / 8 <= 512) hipLaunchKernelGGL(kva_bwd_kernel<512>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(kva_bwd_kernel<1024>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    colsum(G, 4L * C, ws, dparams, (hipStream_t)stream);       // dparams = [dk_k | dk_a | da0 | dv0], C floats each
    return done();
}

int vrwkv_post_fwd_bf16(long ntok, int C, float eps, const void* y, const void* r, const void* k, const void* v, const void* g,
                        const void* ln_w, const void* ln_b, const void* r_k, void* out, void* stream) {
    if (ntok <= 0 || !y || !r || !k || !v || !g || !ln_w || !ln_b || !r_k || !out) return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    PostFwd p{ntok, C, eps, (const uint16_t*)y, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)g,
              (const uint16_t*)ln_w, (const uint16_t*)ln_b, (const uint16_t*)r_k, (uint16_t*)out};
    hipLaunchKernelGGL(post_fwd_kernel, tok_grid(ntok, TPB_POST), dim3(C / 8), 0, (hipStream_t)stream, p);
    return done();
}
int vrwkv_post_bwd_bf16(long ntok, int C, float eps, const void* y, const void* r, const void* k, const void* v, const void* g,
                        const void* ln_w, const void* ln_b, const void* r_k, const void* dout,
                        void* dy, void* dr, void* dk, void* dv, void* dg, float* dparams, float* ws, void* stream) {
    if (ntok <= 0 || !y || !r || !k || !v || !g || !ln_w || !ln_b || !r_k || !dout || !dy || !dr || !dk || !dv || !dg || !dparams || !ws)
        return VRWKV_EINVAL;
    if (!ok_c(C)) return VRWKV_ESHAPE;
    PostBwd p{ntok, C, eps, (const uint16_t*)y, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)g,
              (const uint16_t*)ln_w, (const uint16_t*)ln_b, (const uint16_t*)r_k, (const uint16_t*)dout,
              (uint16_t*)dy, (uint16_t*)dr, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dg, ws};
    const int G = bwd_grid_pf(ntok, C);
    if (C / 8 <= 256) hipLaunchKernelGGL(post_bwd_kernel<256>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    else if (C / 8 <= 512) hipLaunchKernelGGL(post_bwd_kernel<512>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(post_bwd_kernel<1024>, dim3(G), dim3(C / 8), 0, (hipStream_t)stream, p);
    colsum(G, 3L * C, ws, dparams, (hipStream_t)stream);       // dparams = [dln_w | dln_b | dr_k]
    return done();
}

int vrwkv_relusq_fwd_bf16(long n, const void* h, void* y, void* stream) {
    if (n <= 0 || !h || !y) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    const long n8 = n / 8;
    const long blocks = (n8 + 255) / 256;
    if (blocks > 0x7fffffffL) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(relusq_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n8, (const uint16_t*)h, (uint16_t*)y);
    return done();
}
int vrwkv_relusq_bwd_bf16(long n, const void* h, const void* dy, void* dh, void* stream) {
    if (n <= 0 || !h || !dy || !dh) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    const long n8 = n / 8;
    const long blocks = (n8 + 255) / 256;
    if (blocks > 0x7fffffffL) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(relusq_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n8, (const uint16_t*)h, (const uint16_t*)dy, (uint16_t*)dh);
    return done();
}

}  // extern "C"
