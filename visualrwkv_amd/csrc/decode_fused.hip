// Fused per-token kernels of the decode step (stateful generation).  At T = 1 a layer of
// RWKV_Tmix_x070 / RWKV_CMix_x070 (VisualRWKV-v7/v7.00/src/model.py:166-194,221-227,247-254) is launch-bound: the
// captured step spends ~5-8 us per kernel whatever the kernel does.
//
//   tmix_head   : per (b, head): second LoRA stage of w / a / g / v-gate, decay soft-clamp, k/v/a glue, the WKV7
//                 state step, GroupNorm + bonus + gate (was: gemv, decay_fwd, kva_fwd, wkv7_step, post_fwd), plus the
//                 replacement of the carried time-mix row as a side job.
//   ln_mix_prev : LayerNorm of the residual row, token shift against the carried previous row, M lerps, and the
//                 update of the carried row (was: layer_norm, mix_fwd_prev, copy_).  Used for widths the LayerNorm
//                 fold of gemv_decode.hip does not cover (C < 512 or C > 4096); otherwise that fold does this work
//                 inside the consuming GEMV launch.
//
// Intermediate tensors of the unfused path are bf16; the same roundings are kept here (rb()), so the fused step
// tracks the unfused one to reduction-order differences.
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

constexpr int LM_MAXM = 6;
constexpr int LM_THREADS = 256;
constexpr int LM_MAXCH = 4;                  // 8-channel chunks per thread: C <= 8192

DEVFN float rb(float x) { return __uint_as_float(f32_to_bf16_bits(x) << 16); }     // round to bf16, keep as fp32
DEVFN float sigmoidf_(float x) { return 1.f / (1.f + fast_exp(-x)); }

struct V8 { float f[8]; };
DEVFN V8 ld8f(const uint16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    V8 r;
    r.f[0] = bf16_lo(u.x); r.f[1] = bf16_hi(u.x); r.f[2] = bf16_lo(u.y); r.f[3] = bf16_hi(u.y);
    r.f[4] = bf16_lo(u.z); r.f[5] = bf16_hi(u.z); r.f[6] = bf16_lo(u.w); r.f[7] = bf16_hi(u.w);
    return r;
}
DEVFN void st8f(uint16_t* p, const V8& v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v.f[0], v.f[1]), pack_bf16x2(v.f[2], v.f[3]),
                                              pack_bf16x2(v.f[4], v.f[5]), pack_bf16x2(v.f[6], v.f[7]));
}

// sum over the workgroup (4 waves); every thread gets the result
DEVFN float wg_sum(float v, float* red) {
    v = group_sum<6>(v);
    __syncthreads();                                     // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

struct LnMixArgs {
    const uint16_t *x, *ln_w, *ln_b;
    uint16_t* x_prev;                        // (B,C) carried row: read, then overwritten with LN(x)
    const uint16_t* mu[LM_MAXM];
    uint16_t* out[LM_MAXM];
    int C, M;
    float eps;
};

DEVFN V8 unpack(const uint4& u) {
    V8 r;
    r.f[0] = bf16_lo(u.x); r.f[1] = bf16_hi(u.x); r.f[2] = bf16_lo(u.y); r.f[3] = bf16_hi(u.y);
    r.f[4] = bf16_lo(u.z); r.f[5] = bf16_hi(u.z); r.f[6] = bf16_lo(u.w); r.f[7] = bf16_hi(u.w);
    return r;
}

// One workgroup per row.  Everything the row needs (x, LayerNorm parameters, the carried row, the lerp coefficients)
// is loaded before the first reduction: one memory round trip per launch.
__global__ __launch_bounds__(LM_THREADS) void ln_mix_prev_kernel(LnMixArgs a) {
    __shared__ float red[4];
    const int C = a.C, nch = C / 8;
    const size_t row = (size_t)blockIdx.x * C;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 xr[LM_MAXCH], lwr[LM_MAXCH], lbr[LM_MAXCH], xpr[LM_MAXCH], mur[LM_MAXCH][LM_MAXM];
#pragma unroll
    for (int i = 0; i < LM_MAXCH; ++i) {
        const int ch = threadIdx.x + i * LM_THREADS;
        const bool live = ch < nch;
        xr[i] = live ? *reinterpret_cast<const uint4*>(a.x + row + ch * 8) : zero;
        lwr[i] = live ? *reinterpret_cast<const uint4*>(a.ln_w + ch * 8) : zero;
        lbr[i] = live ? *reinterpret_cast<const uint4*>(a.ln_b + ch * 8) : zero;
        xpr[i] = live ? *reinterpret_cast<const uint4*>(a.x_prev + row + ch * 8) : zero;
#pragma unroll
        for (int j = 0; j < LM_MAXM; ++j) mur[i][j] = (live && j < a.M) ? *reinterpret_cast<const uint4*>(a.mu[j] + ch * 8) : zero;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LM_MAXCH; ++i) {
        const V8 xv = unpack(xr[i]);                     // dead chunks hold zeros
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xv.f[e];
    }
    const float mean = wg_sum(s, red) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LM_MAXCH; ++i) {
        if (threadIdx.x + i * LM_THREADS < nch) {
            const V8 xv = unpack(xr[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xv.f[e] - mean; s2 = fmaf(d, d, s2); }
        }
    }
    const float rstd = 1.f / sqrtf(wg_sum(s2, red) / (float)C + a.eps);
#pragma unroll
    for (int i = 0; i < LM_MAXCH; ++i) {
        const int ch = threadIdx.x + i * LM_THREADS;
        if (ch < nch) {
            const int c0 = ch * 8;
            const V8 xv = unpack(xr[i]), lw = unpack(lwr[i]), lb = unpack(lbr[i]), xp = unpack(xpr[i]);
            V8 h, xx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h.f[e] = rb((xv.f[e] - mean) * rstd * lw.f[e] + lb.f[e]);
                xx.f[e] = xp.f[e] - h.f[e];
            }
            st8f(a.x_prev + row + c0, h);
#pragma unroll
            for (int j = 0; j < LM_MAXM; ++j) {
                if (j < a.M) {
                    const V8 m = unpack(mur[i][j]);
                    V8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.f[e] = fmaf(xx.f[e], m.f[e], h.f[e]);
                    st8f(a.out[j] + row + c0, o);
                }
            }
        }
    }
}

constexpr int TH_THREADS = 256;

struct HeadArgs {
    int B, H, C;
    const uint16_t *r, *k, *v, *v_first;     // (B,C); v_first null on layer 0 (no value residual)
    const uint16_t* hid[4];                  // LoRA hidden vectors (B,D_i): w (tanh applied), a, g (sigmoid applied), v-gate
    const uint16_t* W2t[4];                  // second factors TRANSPOSED, (C, D_i) row-major: w2^T, a2^T, g2^T, v2^T
    int D[4];
    const uint16_t *w0, *a0, *v0, *k_k, *k_a, *r_k, *ln_w, *ln_b;     // (C)
    float eps;
    float* state;                            // (B,H,64,64) fp32, in place
    uint16_t* out;                           // (B,C)
    const uint16_t* carry_src;               // optional side job: carry_dst[b, 64h + c] = carry_src[b, 64h + c]
    uint16_t* carry_dst;
};

DEVFN float dot8(const uint4& a, const uint4& b) {
    return bf16_lo(a.x) * bf16_lo(b.x) + bf16_hi(a.x) * bf16_hi(b.x) + bf16_lo(a.y) * bf16_lo(b.y) + bf16_hi(a.y) * bf16_hi(b.y)
         + bf16_lo(a.z) * bf16_lo(b.z) + bf16_hi(a.z) * bf16_hi(b.z) + bf16_lo(a.w) * bf16_lo(b.w) + bf16_hi(a.w) * bf16_hi(b.w);
}

// Thread (c, qd): channel c of the head, quarter qd of every reduction (LoRA widths, state columns).  Every global
// load is issued before the first barrier -- the launch costs one memory round trip, the arithmetic is ~1 us.
__global__ __launch_bounds__(TH_THREADS) void tmix_head_kernel(HeadArgs p) {
    __shared__ float S[64][65];
    __shared__ float part[4][4][64];          // [kind][quarter][channel] partial sums (LoRA, then sa / y of the step)
    __shared__ float vec[6][64];              // decay, q, k, z, a of the step; v2
    const int tid = threadIdx.x, c = tid & 63, qd = uniform_i32(tid >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t cb = (size_t)b * p.C + h * 64;          // this head's 64 channels of batch row b
    const int hc = h * 64 + c;
    float* sp = p.state + (size_t)blockIdx.x * 4096;
    float4 st[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = reinterpret_cast<const float4*>(sp)[tid + i * TH_THREADS];
    if (p.carry_src && qd == 1) p.carry_dst[cb + c] = p.carry_src[cb + c];
    const bool vres = p.v_first != nullptr;
    const int nk = vres ? 4 : 3;
    // per-channel operands (used by wave 0 only; loaded by every wave to keep the code uniform)
    const float kx = bf16_to_f32(p.k[cb + c]), vx = bf16_to_f32(p.v[cb + c]), rr = bf16_to_f32(p.r[cb + c]);
    const float vf = vres ? bf16_to_f32(p.v_first[cb + c]) : 0.f, v0 = vres ? bf16_to_f32(p.v0[hc]) : 0.f;
    const float w0 = bf16_to_f32(p.w0[hc]), a0 = bf16_to_f32(p.a0[hc]), k_k = bf16_to_f32(p.k_k[hc]), k_a = bf16_to_f32(p.k_a[hc]);
    const float r_k = bf16_to_f32(p.r_k[hc]), ln_w = bf16_to_f32(p.ln_w[hc]), ln_b = bf16_to_f32(p.ln_b[hc]);
    // second LoRA stage: a quarter of row hc of W2^T (16-byte loads) against the same quarter of the hidden vector
    // (wave-uniform address: scalar loads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = 0.f;
        if (i < nk) {
            const int D = p.D[i], Dq = D / 4;
            const uint4* W = reinterpret_cast<const uint4*>(p.W2t[i] + (size_t)hc * D + qd * Dq);
            const uint4* hv = reinterpret_cast<const uint4*>(p.hid[i] + (size_t)b * D + qd * Dq);
#pragma unroll 4
            for (int ch = 0; ch < Dq / 8; ++ch) acc += dot8(W[ch], hv[ch]);
        }
        part[i][qd][c] = acc;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * TH_THREADS, r = e >> 4, cc = (e & 15) * 4;
        S[r][cc] = st[i].x; S[r][cc + 1] = st[i].y; S[r][cc + 2] = st[i].z; S[r][cc + 3] = st[i].w;
    }
    __syncthreads();
    float gate = 0.f, k2 = 0.f, v2 = 0.f;
    if (qd == 0) {
        float lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = rb(part[i][0][c] + part[i][1][c] + part[i][2][c] + part[i][3][c]);
        gate = lo[2];
        // decay soft-clamp (tmix_fused.hip decay_fwd_kernel), then the step's w = exp(-exp(w_raw))
        const float u = lo[0] + w0;
        const float w_raw = rb(-(fmaxf(-u, 0.f) + fast_log(1.f + fast_exp(-fabsf(u)))) - 0.5f);
        // k / v / a glue (kva_fwd_kernel)
        const float a = sigmoidf_(a0 + lo[1]);
        float kk = kx * k_k;
        const float ss = group_sum<6>(kk * kk);
        kk *= 1.f / fmaxf(sqrtf(ss), 1e-12f);
        k2 = rb(kx * (1.f + (a - 1.f) * k_a));
        v2 = vres ? rb(vx + (vf - vx) * sigmoidf_(v0 + lo[3])) : vx;
        vec[0][c] = fast_exp(-fast_exp(w_raw));
        vec[1][c] = rr;
        vec[2][c] = k2;
        vec[3][c] = rb(-kk);
        vec[4][c] = rb(kk * a);
        vec[5][c] = v2;
    }
    __syncthreads();
    // wkv7_step_kernel: thread (c, qd) owns columns [16 qd, 16 qd + 16) of state row c
    const int j0 = qd * 16;
    float sa = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sa = fmaf(S[c][j0 + j], vec[3][j0 + j], sa);
    part[0][qd][c] = sa;
    __syncthreads();
    sa = part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
    const float vi = vec[5][c];
    float y = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float s = fmaf(S[c][j0 + j], vec[0][j0 + j], fmaf(sa, vec[4][j0 + j], vec[2][j0 + j] * vi));
        S[c][j0 + j] = s;
        y = fmaf(s, vec[1][j0 + j], y);
    }
    part[1][qd][c] = y;
    __syncthreads();
    if (qd == 0) {
        y = rb(part[1][0][c] + part[1][1][c] + part[1][2][c] + part[1][3][c]);
        // GroupNorm over the head + bonus + gate (post_fwd_kernel)
        const float mean = group_sum<6>(y) * (1.f / 64.f);
        const float dlt = y - mean;
        const float rstd = fast_rsqrt(group_sum<6>(dlt * dlt) * (1.f / 64.f) + p.eps);
        const float sb = group_sum<6>(rr * k2 * r_k);
        p.out[cb + c] = (uint16_t)f32_to_bf16_bits((dlt * rstd * ln_w + ln_b + sb * v2) * gate);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * TH_THREADS, r = e >> 4, cc = (e & 15) * 4;
        reinterpret_cast<float4*>(sp)[e] = make_float4(S[r][cc], S[r][cc + 1], S[r][cc + 2], S[r][cc + 3]);
    }
}

}  // namespace

extern "C" int vrwkv_decode_ln_mix_bf16(int B, int C, int M, const void* x, const void* ln_w, const void* ln_b, float eps,
                                        void* x_prev, const void* const* mu, void* const* out, void* stream) {
    if (B <= 0 || M <= 0 || M > LM_MAXM || !x || !ln_w || !ln_b || !x_prev || !mu || !out) return VRWKV_EINVAL;
    if (C <= 0 || C % 8 != 0 || C > 8 * LM_THREADS * LM_MAXCH) return VRWKV_ESHAPE;
    LnMixArgs a{};
    a.x = (const uint16_t*)x; a.ln_w = (const uint16_t*)ln_w; a.ln_b = (const uint16_t*)ln_b; a.x_prev = (uint16_t*)x_prev;
    a.C = C; a.M = M; a.eps = eps;
    uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ln_w) | reinterpret_cast<uintptr_t>(ln_b) |
                   reinterpret_cast<uintptr_t>(x_prev);
    for (int j = 0; j < M; ++j) {
        if (!mu[j] || !out[j]) return VRWKV_EINVAL;
        a.mu[j] = (const uint16_t*)mu[j]; a.out[j] = (uint16_t*)out[j];
        al |= reinterpret_cast<uintptr_t>(mu[j]) | reinterpret_cast<uintptr_t>(out[j]);
    }
    if (al & 15u) return VRWKV_EALIGN;
    hipLaunchKernelGGL(ln_mix_prev_kernel, dim3((unsigned)B), dim3(LM_THREADS), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

extern "C" int vrwkv_decode_tmix_head_bf16(int B, int H, const void* r, const void* k, const void* v, const void* v_first,
                                           const void* const* hid, const void* const* W2t, const int* D,
                                           const void* w0, const void* a0, const void* v0, const void* k_k, const void* k_a,
                                           const void* r_k, const void* ln_w, const void* ln_b, float eps,
                                           float* state, void* out, const void* carry_src, void* carry_dst,
                                           void* stream) {
    if ((carry_src == nullptr) != (carry_dst == nullptr)) return VRWKV_EINVAL;
    if (B <= 0 || H <= 0 || !r || !k || !v || !hid || !W2t || !D || !w0 || !a0 || !k_k || !k_a || !r_k || !ln_w || !ln_b ||
        !state || !out) return VRWKV_EINVAL;
    if (v_first && !v0) return VRWKV_EINVAL;
    if (reinterpret_cast<uintptr_t>(state) & 15u) return VRWKV_EALIGN;
    HeadArgs p{};
    p.B = B; p.H = H; p.C = H * 64;
    p.r = (const uint16_t*)r; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.v_first = (const uint16_t*)v_first;
    const int nk = v_first ? 4 : 3;
    for (int i = 0; i < nk; ++i) {
        if (!hid[i] || !W2t[i]) return VRWKV_EINVAL;
        if (D[i] <= 0 || D[i] % 32 != 0) return VRWKV_ESHAPE;
        if ((reinterpret_cast<uintptr_t>(hid[i]) | reinterpret_cast<uintptr_t>(W2t[i])) & 15u) return VRWKV_EALIGN;
        p.hid[i] = (const uint16_t*)hid[i]; p.W2t[i] = (const uint16_t*)W2t[i]; p.D[i] = D[i];
    }
    p.w0 = (const uint16_t*)w0; p.a0 = (const uint16_t*)a0; p.v0 = (const uint16_t*)v0; p.k_k = (const uint16_t*)k_k;
    p.k_a = (const uint16_t*)k_a; p.r_k = (const uint16_t*)r_k; p.ln_w = (const uint16_t*)ln_w; p.ln_b = (const uint16_t*)ln_b;
    p.eps = eps; p.state = state; p.out = (uint16_t*)out;
    p.carry_src = (const uint16_t*)carry_src; p.carry_dst = (uint16_t*)carry_dst;
    hipLaunchKernelGGL(tmix_head_kernel, dim3((unsigned)(B * H)), dim3(TH_THREADS), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
