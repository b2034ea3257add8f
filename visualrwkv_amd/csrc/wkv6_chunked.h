// WKV6 (RWKV-6 "Finch" time-mix recurrence, BASELINE config 4) forward and backward for gfx950 -- chunked matmul form
// on the MFMA units, same building blocks as the WKV7 kernels (wkv7_chunked.h).
//
// Replaces kernel_forward / kernel_backward_111 / kernel_backward_222 of
// VisualRWKV-v6/v6.0/cuda/wkv6_cuda.cu:7-227.  Per head (N = 64), state S[i][j] (i = value, j = key), log decay
// ew_t = -exp(w_raw_t) (computed by the caller, src/model.py:62), d_t = exp(ew_t):
//     y_t[i] = sum_j r_t[j] (u[j] k_t[j] v_t[i] + S[i][j])           S[i][j] <- S[i][j] d_t[j] + k_t[j] v_t[i]
// The reference walks T token by token (forward) and five times (backward, with a per-thread array of T floats).
// Here a 16-token chunk is one set of 16x16 MFMA tiles.  With x_t = sum_{s<=t} ew_s inside the chunk and
// m = x at the chunk's midpoint (keeps both exponentials in fp32 range for per-token log decays down to about -11;
// beyond that the exponent is clamped at 80):
//     Re = r e^{x_{t-1}}     Rt = r e^{x_{t-1}-m}     Kh = k e^{m-x_t}     Kb = k e^{x_L-x_t}     c_L = e^{x_L}
//     A  = tril_strict(Rt Kh^T) + diag(sum_j r u k)
//     Y  = Re S0^T + A V                      S_L = S0 diag(c_L) + V^T Kb
// Precision: fp32 operands are split hi/lo ("bf16x3", see wkv7_chunked.h); v, dy are exact bf16.
// oracle/wkv6_oracle.py::wkv6_chunked is the CPU statement of the same algebra (1e-15 vs fp64 autograd).
//
// One workgroup = 4 waves per (b,h); lane (t = lane & 15, g = lane >> 4) of wave w prepares token t, key columns
// 16w+4g..+3 of a chunk into a double-buffered LDS image (one LDS-only barrier per chunk); then wave w owns value
// columns i in [16w, 16w+16) of S^T (4 accumulator tiles) and produces y for them.  T need not be a multiple of 16
// (the reference has no such requirement): tokens past T are prepared as r = k = v = 0, ew = 0 and never stored.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_kernels.h>
#include <wkv7_chunked.h>

namespace wkv6c {

using wkv7c::JT;
using wkv7c::L;
using wkv7c::N;
using wkv7c::TJ;
using wkv7c::ld8;
using wkv7c::ld_nat;
using wkv7c::ld_perm;
using wkv7c::mk8;
using wkv7c::split4;
using wkv7c::st8;
using wkv7c::unpack4;
using wkv7c::zero4;

constexpr float EXP_CLAMP = 80.f;

struct Fwd6Args {
    int T, H;
    const uint16_t *r, *k, *v;      // bf16 (B,T,H*N)
    const float* ew;                // f32  (B,T,H*N)   log decay = -exp(w_raw)
    const uint16_t* u;              // bf16 (H*N)
    uint16_t* y;                    // bf16 (B,T,H*N)
    float* s;                       // optional f32 (B*H, ceil(T/16), N, N): S^T at the START of every chunk (for the backward)
};

struct Buf6F {
    uint16_t re[2][L][TJ];          // Re hi,lo   [t][j]
    uint16_t rt[2][L][TJ];          // Rt hi,lo   [t][j]
    uint16_t kh[2][L][TJ];          // Kh hi,lo   [t][j]
    uint16_t kbT[2][N][JT];         // Kb hi,lo   [j][t]
    uint16_t vT[N][JT];             // v          [i][t]
    float cl[N];                    // c_L[j]
    float dpart[4][L];              // per producer wave: sum over its 16 key columns of r u k
};
struct Lds6F { Buf6F b[2]; };

struct Raw6 { uint2 r, k, v; float4 ew; };

DEVFN void st_b16x4_T(uint16_t (*M)[JT], int row0, int col, uint2 v) {   // 4 values -> M[row0+e][col]
    M[row0 + 0][col] = (uint16_t)v.x; M[row0 + 1][col] = (uint16_t)(v.x >> 16);
    M[row0 + 2][col] = (uint16_t)v.y; M[row0 + 3][col] = (uint16_t)(v.y >> 16);
}

// decay factors of one lane's 4 columns: inclusive scan of ew over the 16 tokens of the row
struct Decay6 { float e_r[4], e_re[4], e_h[4], e_b[4], c_l[4]; };
DEVFN Decay6 decay_factors(const float* ew, int lane) {
    Decay6 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = ew[e];
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        const float xl = lane_bcast(x, (lane & 48) | 15), m = lane_bcast(x, (lane & 48) | 7);
        const float xp = x - ew[e];
        d.e_re[e] = fast_exp(xp);
        d.e_r[e] = fast_exp(fminf(xp - m, EXP_CLAMP));
        d.e_h[e] = fast_exp(fminf(m - x, EXP_CLAMP));
        d.e_b[e] = fast_exp(xl - x);
        d.c_l[e] = fast_exp(xl);
    }
    return d;
}

// D[x_row][y_row] = sum_j X[x_row][j] Y[y_row][j] over 64 natural-k columns (bf16x3): reg r <-> x_row 4g+r, lane c16 <-> y_row
DEVFN f32x4 score6(const uint16_t (*Xh)[TJ], const uint16_t (*Xl)[TJ], const uint16_t (*Yh)[TJ], const uint16_t (*Yl)[TJ],
                   int c16, int g) {
    f32x4 acc = zero4();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 xh = ld_nat(Xh, c16, kb, g), xl = ld_nat(Xl, c16, kb, g);
        const bf16x8 yh = ld_nat(Yh, c16, kb, g), yl = ld_nat(Yl, c16, kb, g);
        acc = mfma_16x16x32_bf16(xh, yh, acc);
        acc = mfma_16x16x32_bf16(xh, yl, acc);
        acc = mfma_16x16x32_bf16(xl, yh, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void fwd6_kernel(Fwd6Args p) {
    __shared__ __attribute__((aligned(16))) Lds6F lds;
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, g = lane >> 4;
    const size_t ts = (size_t)H * N;                                   // token stride
    const size_t head_base = ((size_t)(blockIdx.x / H) * T * H + (blockIdx.x % H)) * N;
    const int nchunk = (T + L - 1) / L;
    const int j0 = 16 * wave + 4 * g;
    float uu[4];
    unpack4(*reinterpret_cast<const uint2*>(p.u + (size_t)(blockIdx.x % H) * N + j0), uu);

    auto fetch = [&](Raw6& rc, int c) {
        const int tt = c * L + c16;
        if (tt < T) {
            const size_t o = head_base + (size_t)tt * ts + j0;
            rc.r = *reinterpret_cast<const uint2*>(p.r + o); rc.k = *reinterpret_cast<const uint2*>(p.k + o);
            rc.v = *reinterpret_cast<const uint2*>(p.v + o); rc.ew = *reinterpret_cast<const float4*>(p.ew + o);
        } else {
            rc.r = make_uint2(0, 0); rc.k = make_uint2(0, 0); rc.v = make_uint2(0, 0);
            rc.ew = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    f32x4 S[4];                           // S^T tiles: S[jb][r] = S^T[j = 16jb+4g+r][i = 16*wave + c16]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) S[jb] = zero4();
    float* sdst = p.s ? p.s + ((size_t)blockIdx.x * nchunk * N) * N + 16 * wave + c16 : nullptr;

    Raw6 rc;
    fetch(rc, 0);
    for (int c = 0; c < nchunk; ++c) {
        Buf6F& B = lds.b[c & 1];
        // ------------------------------------------------------------ prepare chunk c (key columns j0..j0+3, token c16)
        {
            float r[4], k[4];
            unpack4(rc.r, r); unpack4(rc.k, k);
            const float ew[4] = {rc.ew.x, rc.ew.y, rc.ew.z, rc.ew.w};
            const uint2 vraw = rc.v;
            if (c + 1 < nchunk) fetch(rc, c + 1);
            const Decay6 d = decay_factors(ew, lane);
            float re[4], rt[4], kh[4], kb[4], dp = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                re[e] = r[e] * d.e_re[e]; rt[e] = r[e] * d.e_r[e]; kh[e] = k[e] * d.e_h[e]; kb[e] = k[e] * d.e_b[e];
                dp = fmaf(r[e] * uu[e], k[e], dp);
            }
            dp += lane_xor16(dp);
            dp += lane_xor32(dp);
            if (g == 0) B.dpart[wave][c16] = dp;
            uint2 h, l;
            split4(re, h, l); st8(&B.re[0][c16][j0], h); st8(&B.re[1][c16][j0], l);
            split4(rt, h, l); st8(&B.rt[0][c16][j0], h); st8(&B.rt[1][c16][j0], l);
            split4(kh, h, l); st8(&B.kh[0][c16][j0], h); st8(&B.kh[1][c16][j0], l);
            split4(kb, h, l); st_b16x4_T(B.kbT[0], j0, c16, h); st_b16x4_T(B.kbT[1], j0, c16, l);
            st_b16x4_T(B.vT, j0, c16, vraw);
            if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(d.c_l[0], d.c_l[1], d.c_l[2], d.c_l[3]);
        }
        block_sync_lds();
        // ------------------------------------------------------------ consume chunk c (value columns 16*wave + c16)
        // A^T = Kh Rt^T: lane (g, c16 = t), reg r <-> s = 4g+r, i.e. the A-operand image of A (strictly lower + diagonal)
        f32x4 at = score6(B.kh[0], B.kh[1], B.rt[0], B.rt[1], c16, g);
        const float dt = B.dpart[0][c16] + B.dpart[1][c16] + B.dpart[2][c16] + B.dpart[3][c16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * g + r;
            at[r] = s < c16 ? at[r] : (s == c16 ? dt : 0.f);
        }
        uint2 ah, al;
        split4(at, ah, al);
        const uint2 vv = ld8(&B.vT[16 * wave + c16][4 * g]);
        const bf16x8 bvv = mk8(vv, vv);
        uint2 sh[4], sl[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) split4(S[jb], sh[jb], sl[jb]);
        const bf16x8 bsh[2] = {mk8(sh[0], sh[1]), mk8(sh[2], sh[3])};
        const bf16x8 bsl[2] = {mk8(sl[0], sl[1]), mk8(sl[2], sl[3])};
        f32x4 Y = mfma_16x16x32_bf16(mk8(ah, al), bvv, zero4());                 // (A_hi + A_lo) V
        f32x4 Y1 = zero4();
        {
            const bf16x8 xh = ld_perm(B.re[0], c16, 0, g), xl = ld_perm(B.re[1], c16, 0, g);
            Y = mfma_16x16x32_bf16(xh, bsh[0], Y); Y = mfma_16x16x32_bf16(xh, bsl[0], Y); Y = mfma_16x16x32_bf16(xl, bsh[0], Y);
        }
        {
            const bf16x8 xh = ld_perm(B.re[0], c16, 1, g), xl = ld_perm(B.re[1], c16, 1, g);
            Y1 = mfma_16x16x32_bf16(xh, bsh[1], Y1); Y1 = mfma_16x16x32_bf16(xh, bsl[1], Y1); Y1 = mfma_16x16x32_bf16(xl, bsh[1], Y1);
        }
        {
            const uint32_t y01 = cvt_pk_bf16(Y[0] + Y1[0], Y[1] + Y1[1]), y23 = cvt_pk_bf16(Y[2] + Y1[2], Y[3] + Y1[3]);
            uint16_t* yc = p.y + head_base + (size_t)(c * L + 4 * g) * ts + 16 * wave + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c * L + 4 * g + r < T) yc[r * ts] = (uint16_t)((r < 2 ? y01 : y23) >> (16 * (r & 1)));
        }
        // S_L^T = diag(c_L) S0^T + Kb^T V ; checkpoint = state at the START of the chunk
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            if (sdst) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sdst[((size_t)c * N + 16 * jb + 4 * g + r) * N] = S[jb][r];
            }
            const float4 cl = *reinterpret_cast<const float4*>(&B.cl[16 * jb + 4 * g]);
            f32x4 acc = S[jb];
            acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
            const int j = 16 * jb + c16;
            S[jb] = mfma_16x16x32_bf16(mk8(ld8(&B.kbT[0][j][4 * g]), ld8(&B.kbT[1][j][4 * g])), bvv, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Closed-form chunk backward from the chunk-start state S0 (checkpointed by the forward) and dS = dL/dS_L carried
// from the later chunks.  With dA = dY V^T (dd = its diagonal, dAl = its strictly lower part):
//     dV   = A^T dY + Kb dS^T                 dS0  = dS diag(c_L) + dY^T Re
//     dRe  = dY S0        dRa = dAl Kh        dKh  = dAl^T Rt        dKb = V dS
//     gr   = dRe e^{x_{t-1}} + dRa e^{x_{t-1}-m} + dd u k            gk = dKh e^{m-x_t} + dKb e^{x_L-x_t} + dd u r
//     gu  += dd r k
//     dL/dx_t = -dKh Kh - dKb Kb + [dRe Re + dRa Rt]_{t+1} + [t = L-1] (sum_t dKb Kb + sum_i dS S0 c_L)
//               + [t = mid] sum_t (dKh Kh - dRa Rt)          g_ew_s = sum_{t >= s} dL/dx_t,   gw = g_ew ew
// (gw is the gradient with respect to the raw w: d ew / d w_raw = ew; kernel_backward_222 ends with the same factor,
// wkv6_cuda.cu:203,225.)  Work split: products contracting over the key index run on value columns i (wave w holds
// dS^T[:, 16w..] tiles), products contracting over i or t run on key columns j (a second copy of dS as [i][j] tiles) --
// no cross-wave reduction.  A wave prepares and finishes the same 16 key columns, so the raw r, k, ew and the decay
// factors stay in registers between the two phases.  Two LDS-only barriers per chunk, 57 KB of LDS.
struct Bwd6Args {
    int T, H;
    const uint16_t *r, *k, *v;
    const float* ew;
    const uint16_t *u, *gy;
    const float* s;                 // chunk-start checkpoints written by the forward
    uint16_t *gr, *gk, *gv, *gw;    // bf16 (B,T,H*N)
    uint16_t* gu;                   // bf16 (B,H*N): per-sample, summed over the batch by the caller (src/model.py:84)
};

constexpr int RS6 = 68;             // row stride (floats) of the tail bounce strips (conflict-free for the float4 column reads)
struct Lds6B {
    uint16_t rt[2][L][TJ], kh[2][L][TJ], kb[2][L][TJ];        // [t][j] hi,lo
    uint16_t dy[L][TJ], v[L][TJ];                             // [t][i] exact bf16
    uint16_t reT[2][N][JT], rtT[2][N][JT], khT[2][N][JT];     // [j][t] hi,lo
    uint16_t dyT[N][JT];                                      // [i][t]
    float cl[N], glast[N];
    float dpart[4][L];
    float res[4][L][RS6];                                     // dRe dRa dKh dKb: C layout -> token-per-lane
};

// sum_i X[x_row][i] Y[y_row][i] for exact-bf16 images
DEVFN f32x4 score6_exact(const uint16_t (*X)[TJ], const uint16_t (*Y)[TJ], int c16, int g) {
    f32x4 acc = zero4();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) acc = mfma_16x16x32_bf16(ld_nat(X, c16, kb, g), ld_nat(Y, c16, kb, g), acc);
    return acc;
}
// tiles (4 x f32x4, C layout) -> permuted-k B operands, hi and lo, for the two 32-wide k blocks
DEVFN void tiles_to_b6(const f32x4* tl, bf16x8* bh, bf16x8* bl) {
    uint2 h[4], l[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) split4(tl[b], h[b], l[b]);
    bh[0] = mk8(h[0], h[1]); bh[1] = mk8(h[2], h[3]);
    bl[0] = mk8(l[0], l[1]); bl[1] = mk8(l[2], l[3]);
}
// acc += P Q with P^T given in C layout (registers) and Q[k][n] read as the hi/lo [n][k] image rows
DEVFN f32x4 mm_reg_img(f32x4 acc, f32x4 pt, const uint16_t (*Qh)[JT], const uint16_t (*Ql)[JT], int col, int g) {
    uint2 ph, pl;
    split4(pt, ph, pl);
    const uint2 qh = ld8(&Qh[col][4 * g]), ql = ld8(&Ql[col][4 * g]);
    acc = mfma_16x16x32_bf16(mk8(ph, ph), mk8(qh, ql), acc);
    return mfma_16x16x32_bf16(mk8(pl.x, pl.y, 0u, 0u), mk8(qh.x, qh.y, 0u, 0u), acc);
}

__global__ __launch_bounds__(256) void bwd6_kernel(Bwd6Args p) {
    __shared__ __attribute__((aligned(16))) Lds6B lds;
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, g = lane >> 4;
    const size_t ts = (size_t)H * N;
    const int hh = blockIdx.x % H;
    const size_t head_base = ((size_t)(blockIdx.x / H) * T * H + hh) * N;
    const int nchunk = (T + L - 1) / L;
    const int j0 = 16 * wave + 4 * g;               // prepare / tail: token c16, columns j0..j0+3
    const int jc = 16 * wave + c16;                 // C-layout column of this lane in the j-split products
    float uu[4];
    unpack4(*reinterpret_cast<const uint2*>(p.u + (size_t)hh * N + j0), uu);
    const float* sbase = p.s + (size_t)blockIdx.x * nchunk * N * N;

    f32x4 dS1[4], dS2[4];                           // dS^T[j = 16jb+4g+r][i = 16w+c16] ;  dS[i = 16ib+4g+r][j = 16w+c16]
#pragma unroll
    for (int x = 0; x < 4; ++x) { dS1[x] = zero4(); dS2[x] = zero4(); }
    float gu_acc[4] = {0.f, 0.f, 0.f, 0.f};

    for (int c = nchunk - 1; c >= 0; --c) {
        // chunk-start state as [i][j] tiles: S0[ib][r] = S[16ib+4g+r][jc] = s[c][jc][16ib+4g+r]
        f32x4 S0[4];
        {
            const float* sp = sbase + ((size_t)c * N + jc) * N + 4 * g;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float4 x = *reinterpret_cast<const float4*>(sp + 16 * ib);
                S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
            }
        }
        // ------------------------------------------------------------ prepare (token c16, columns j0..j0+3)
        const int tt = c * L + c16;
        float r[4], k[4], ew[4];
        Decay6 d;
        {
            uint2 rr = make_uint2(0, 0), kk = rr, vv = rr, gg = rr;
            float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tt < T) {
                const size_t o = head_base + (size_t)tt * ts + j0;
                rr = *reinterpret_cast<const uint2*>(p.r + o); kk = *reinterpret_cast<const uint2*>(p.k + o);
                vv = *reinterpret_cast<const uint2*>(p.v + o); gg = *reinterpret_cast<const uint2*>(p.gy + o);
                e4 = *reinterpret_cast<const float4*>(p.ew + o);
            }
            unpack4(rr, r); unpack4(kk, k);
            ew[0] = e4.x; ew[1] = e4.y; ew[2] = e4.z; ew[3] = e4.w;
            d = decay_factors(ew, lane);
            float re[4], rt[4], kh[4], kb[4], dp = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                re[e] = r[e] * d.e_re[e]; rt[e] = r[e] * d.e_r[e]; kh[e] = k[e] * d.e_h[e]; kb[e] = k[e] * d.e_b[e];
                dp = fmaf(r[e] * uu[e], k[e], dp);
            }
            dp += lane_xor16(dp);
            dp += lane_xor32(dp);
            if (g == 0) lds.dpart[wave][c16] = dp;
            uint2 h, l;
            split4(rt, h, l); st8(&lds.rt[0][c16][j0], h); st8(&lds.rt[1][c16][j0], l);
            st_b16x4_T(lds.rtT[0], j0, c16, h); st_b16x4_T(lds.rtT[1], j0, c16, l);
            split4(kh, h, l); st8(&lds.kh[0][c16][j0], h); st8(&lds.kh[1][c16][j0], l);
            st_b16x4_T(lds.khT[0], j0, c16, h); st_b16x4_T(lds.khT[1], j0, c16, l);
            split4(kb, h, l); st8(&lds.kb[0][c16][j0], h); st8(&lds.kb[1][c16][j0], l);
            split4(re, h, l); st_b16x4_T(lds.reT[0], j0, c16, h); st_b16x4_T(lds.reT[1], j0, c16, l);
            st8(&lds.v[c16][j0], vv);
            st8(&lds.dy[c16][j0], gg);
            st_b16x4_T(lds.dyT, j0, c16, gg);
            if (c16 == 15) *reinterpret_cast<float4*>(&lds.cl[j0]) = make_float4(d.c_l[0], d.c_l[1], d.c_l[2], d.c_l[3]);
        }
        block_sync_lds();
        // ------------------------------------------------------------ scores (every wave, in registers)
        f32x4 ac = score6(lds.rt[0], lds.rt[1], lds.kh[0], lds.kh[1], c16, g);       // A[t = 4g+r][s = c16]
        f32x4 da = score6_exact(lds.dy, lds.v, c16, g);                             // dA[t = 4g+r][s = c16]
        f32x4 dat = score6_exact(lds.v, lds.dy, c16, g);                            // dA[t = c16][s = 4g+r]
        const float dt_row = lds.dpart[0][c16] + lds.dpart[1][c16] + lds.dpart[2][c16] + lds.dpart[3][c16];
        float dd = 0.f;                              // dd_t for t = c16 (all four lane groups)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int t = 4 * g + r4;
            dd += (t == c16) ? da[r4] : 0.f;
            // the diagonal of A is d_t at t == s: lanes whose column c16 equals their row 4g+r4 need d_{c16}
            ac[r4] = c16 < t ? ac[r4] : (c16 == t ? dt_row : 0.f);
            da[r4] = c16 < t ? da[r4] : 0.f;
            dat[r4] = t < c16 ? dat[r4] : 0.f;
        }
        dd += lane_xor16(dd);
        dd += lane_xor32(dd);
        // ------------------------------------------------------------ i-split: dV, dS^T   (value column 16w + c16)
        const uint2 dyv = ld8(&lds.dyT[16 * wave + c16][4 * g]);
        const bf16x8 bdy = mk8(dyv, dyv);
        {
            uint2 ah, al;
            split4(ac, ah, al);
            bf16x8 b1h[2], b1l[2];
            tiles_to_b6(dS1, b1h, b1l);
            f32x4 dV = mfma_16x16x32_bf16(mk8(ah, al), bdy, zero4());                       // A^T dY
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2) {
                const bf16x8 xh = ld_perm(lds.kb[0], c16, kb2, g), xl = ld_perm(lds.kb[1], c16, kb2, g);
                dV = mfma_16x16x32_bf16(xh, b1h[kb2], dV);
                dV = mfma_16x16x32_bf16(xh, b1l[kb2], dV);
                dV = mfma_16x16x32_bf16(xl, b1h[kb2], dV);
            }
            const uint32_t v01 = cvt_pk_bf16(dV[0], dV[1]), v23 = cvt_pk_bf16(dV[2], dV[3]);
            uint16_t* gvc = p.gv + head_base + (size_t)(c * L + 4 * g) * ts + 16 * wave + c16;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (c * L + 4 * g + r4 < T) gvc[r4 * ts] = (uint16_t)((r4 < 2 ? v01 : v23) >> (16 * (r4 & 1)));
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float4 cl = *reinterpret_cast<const float4*>(&lds.cl[16 * jb + 4 * g]);
                f32x4 acc = dS1[jb];
                acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
                const int j = 16 * jb + c16;
                dS1[jb] = mfma_16x16x32_bf16(mk8(ld8(&lds.reT[0][j][4 * g]), ld8(&lds.reT[1][j][4 * g])), bdy, acc);
            }
        }
        // ------------------------------------------------------------ j-split (key column jc)
        f32x4 dRe = zero4(), dKb = zero4();
        {
            bf16x8 s0h[2], s0l[2], d2h[2], d2l[2];
            tiles_to_b6(S0, s0h, s0l);
            tiles_to_b6(dS2, d2h, d2l);
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2) {
                const bf16x8 ydy = ld_perm(lds.dy, c16, kb2, g), yv = ld_perm(lds.v, c16, kb2, g);
                dRe = mfma_16x16x32_bf16(ydy, s0h[kb2], dRe);
                dRe = mfma_16x16x32_bf16(ydy, s0l[kb2], dRe);
                dKb = mfma_16x16x32_bf16(yv, d2h[kb2], dKb);
                dKb = mfma_16x16x32_bf16(yv, d2l[kb2], dKb);
            }
        }
        const f32x4 dRa = mm_reg_img(zero4(), dat, lds.khT[0], lds.khT[1], jc, g);          // dAl Kh
        const f32x4 dKh = mm_reg_img(zero4(), da, lds.rtT[0], lds.rtT[1], jc, g);           // dAl^T Rt
        {
            const float clj = lds.cl[jc];
            float gl = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) gl = fmaf(dS2[ib][r4], S0[ib][r4], gl);
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            if (g == 0) lds.glast[jc] = gl * clj;
            const bf16x8 bre = mk8(ld8(&lds.reT[0][jc][4 * g]), ld8(&lds.reT[1][jc][4 * g]));
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                f32x4 acc = dS2[ib];
                acc[0] *= clj; acc[1] *= clj; acc[2] *= clj; acc[3] *= clj;
                const uint2 dyi = ld8(&lds.dyT[16 * ib + c16][4 * g]);
                dS2[ib] = mfma_16x16x32_bf16(mk8(dyi, dyi), bre, acc);
            }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            lds.res[0][4 * g + r4][jc] = dRe[r4];
            lds.res[1][4 * g + r4][jc] = dRa[r4];
            lds.res[2][4 * g + r4][jc] = dKh[r4];
            lds.res[3][4 * g + r4][jc] = dKb[r4];
        }
        wave_lds_fence();               // strips [16w,16w+16) and glast[16w..] are written and read by this wave only
        // ------------------------------------------------------------ element-wise tail (token c16, columns j0..j0+3)
        {
            const float4 q0 = *reinterpret_cast<const float4*>(&lds.res[0][c16][j0]);
            const float4 q1 = *reinterpret_cast<const float4*>(&lds.res[1][c16][j0]);
            const float4 q2 = *reinterpret_cast<const float4*>(&lds.res[2][c16][j0]);
            const float4 q3 = *reinterpret_cast<const float4*>(&lds.res[3][c16][j0]);
            const float4 g4 = *reinterpret_cast<const float4*>(&lds.glast[j0]);
            const float vre[4] = {q0.x, q0.y, q0.z, q0.w}, vra[4] = {q1.x, q1.y, q1.z, q1.w};
            const float vkh[4] = {q2.x, q2.y, q2.z, q2.w}, vkb[4] = {q3.x, q3.y, q3.z, q3.w};
            const float gls[4] = {g4.x, g4.y, g4.z, g4.w};
            float gr[4], gk[4], gw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ddu = dd * uu[e];
                gr[e] = vre[e] * d.e_re[e] + vra[e] * d.e_r[e] + ddu * k[e];
                gk[e] = vkh[e] * d.e_h[e] + vkb[e] * d.e_b[e] + ddu * r[e];
                gu_acc[e] = fmaf(dd * r[e], k[e], gu_acc[e]);
                const float pa = vra[e] * (r[e] * d.e_r[e]);                      // dRa Rt
                const float pr = vre[e] * (r[e] * d.e_re[e]) + pa;                // dRe Re + dRa Rt
                const float ph = vkh[e] * (k[e] * d.e_h[e]);
                const float pb = vkb[e] * (k[e] * d.e_b[e]);
                float gx = dpp_shl<1>(pr) - ph - pb;
                const float sum_b = group_sum<4>(pb), sum_m = group_sum<4>(ph - pa);
                if (c16 == 15) gx += sum_b + gls[e];
                if (c16 == 7) gx += sum_m;
                gx += dpp_shl<1>(gx); gx += dpp_shl<2>(gx); gx += dpp_shl<4>(gx); gx += dpp_shl<8>(gx);   // suffix sum over t
                gw[e] = gx * ew[e];
            }
            if (tt < T) {
                const size_t o = head_base + (size_t)tt * ts + j0;
                *reinterpret_cast<uint2*>(p.gr + o) = make_uint2(cvt_pk_bf16(gr[0], gr[1]), cvt_pk_bf16(gr[2], gr[3]));
                *reinterpret_cast<uint2*>(p.gk + o) = make_uint2(cvt_pk_bf16(gk[0], gk[1]), cvt_pk_bf16(gk[2], gk[3]));
                *reinterpret_cast<uint2*>(p.gw + o) = make_uint2(cvt_pk_bf16(gw[0], gw[1]), cvt_pk_bf16(gw[2], gw[3]));
            }
        }
        block_sync_lds();               // the next chunk's images overwrite this one's
    }
    // gu[b, h, j] = sum over the tokens of this sample
#pragma unroll
    for (int e = 0; e < 4; ++e) gu_acc[e] = group_sum<4>(gu_acc[e]);
    if (c16 == 0)
        *reinterpret_cast<uint2*>(p.gu + (size_t)blockIdx.x * N + j0) =
            make_uint2(cvt_pk_bf16(gu_acc[0], gu_acc[1]), cvt_pk_bf16(gu_acc[2], gu_acc[3]));
}

}  // namespace wkv6c
