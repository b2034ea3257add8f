// WKV7 backward in chunked (matmul) form on bf16 MFMA with split operands -- gfx950.
//
// Replaces the reference's backward_kernel (VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130): instead of
// un-stepping the state token by token with a division by the decay, each 16-token chunk is
// differentiated in closed form from the chunk-start state S0 = s[c-1] and the saved sa (both written
// by the forward, as in the reference) -- oracle/wkv7_chunked.py::backward is the CPU statement of the
// algorithm (validated to 1e-15 against fp64 autograd).  With the forward quantities of
// wkv7_chunked.h and dS = dL/dS_L entering the chunk:
//     dSA = Ab dS^T + M_qa^T dY          Ab = a c_L/c_t, Kb = k c_L/c_t
//     dR  = (I - M_za)^-T dSA
//     dV  = Kb dS^T + M_qk^T dY + M_zk^T dR
//     dM_za = tril_(dR SA^T)  dM_zk = tril_(dR V^T)  dM_qa = tril(dY SA^T)  dM_qk = tril(dY V^T)
//     dZt = dR S0 + dM_za Ah + dM_zk Kh      dQt = dY S0 + dM_qa Ah + dM_qk Kh
//     dAh = SA dU + dM_za^T Zt + dM_qa^T Qt  dKh = V dU + dM_zk^T Zt + dM_qk^T Qt     (dU = dS diag(c_L))
//     dS0 = dU + dY^T Qt + dR^T Zt
//     dz = dZt c_{t-1}  dq = dQt c_t  da = dAh/c_t  dk = dKh/c_t
//     g_t = dq q - da a - dk k + dz_{t+1} z_{t+1} (+ sum_i dS (.) S_L at t = L);  dw_raw_t = log(w_t) sum_{r>=t} g_r
//
// Work split (one workgroup of 4 waves per (b,h)): products that contract over the key index j are
// split over the value index i (wave w owns i in [16w,16w+16), dS kept as S^T accumulator tiles);
// products that contract over i are split over j (wave w owns j in [16w,16w+16), a second copy of dS
// kept as S tiles), so no cross-wave reduction is ever needed -- the same idea as the reference's
// dstate/dstateT pair, but at tile granularity.  Accumulators are fed back as MFMA B operands with
// permuted k-slots (see wkv7_chunked.h).
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>

namespace wkv7c {

using wkv7::BwdArgs;

struct LdsB {
    uint16_t opnd[12][L][TJ];   // [t][j]: Zt Qt Ah Kh Ab Kb, each hi,lo
    uint16_t trn[8][N][JT];     // [j][t]: ZtT QtT AhT KhT, each hi,lo
    uint16_t ti[6][L][TJ];      // [t][i]: 0 V  1 dY  2 SA_hi 3 SA_lo  4 dR_hi 5 dR_lo
    uint16_t it[3][N][JT];      // [i][t]: 0 dY^T  1 dR^T_hi  2 dR^T_lo
    uint16_t sc[4][2][L][SS];   // A-operand images [row][col], hi/lo: 0 M_qa^T  1 M_qk^T  2 M_zk^T  3 T^T
    uint16_t dsc[8][2][L][SS];  // 0 dM_za 1 dM_za^T 2 dM_zk 3 dM_zk^T 4 dM_qa 5 dM_qa^T 6 dM_qk 7 dM_qk^T
    float cl[N];                // c_L[j]
    float glast[N];             // sum_i dS[i][j] S_L[i][j]
    float res[4][L][N];         // dZt dQt dAh dKh, back in [t][j] form for the element-wise tail
};

DEVFN void st_b16x4_T(uint16_t (*M)[JT], int row0, int col, uint2 v) {   // 4 values -> M[row0+e][col]
    M[row0 + 0][col] = (uint16_t)v.x; M[row0 + 1][col] = (uint16_t)(v.x >> 16);
    M[row0 + 2][col] = (uint16_t)v.y; M[row0 + 3][col] = (uint16_t)(v.y >> 16);
}
DEVFN void st_b16x4_col(uint16_t (*M)[TJ], int row0, int col, uint2 v) {  // 4 values -> M[row0+e][col]
    M[row0 + 0][col] = (uint16_t)v.x; M[row0 + 1][col] = (uint16_t)(v.x >> 16);
    M[row0 + 2][col] = (uint16_t)v.y; M[row0 + 3][col] = (uint16_t)(v.y >> 16);
}

// sum_k X[row c16][k] Y[row c16][k] over 64 natural-k columns; xl / yl < 0 means that operand is exact bf16
template <bool XLO, bool YLO>
DEVFN f32x4 dot64(const uint16_t (*Xh)[TJ], const uint16_t (*Xl)[TJ], const uint16_t (*Yh)[TJ], const uint16_t (*Yl)[TJ],
                  int c16, int g) {
    f32x4 acc = zero4();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 xh = ld_nat(Xh, c16, kb, g), yh = ld_nat(Yh, c16, kb, g);
        acc = mfma_16x16x32_bf16(xh, yh, acc);
        if (YLO) acc = mfma_16x16x32_bf16(xh, ld_nat(Yl, c16, kb, g), acc);
        if (XLO) acc = mfma_16x16x32_bf16(ld_nat(Xl, c16, kb, g), yh, acc);
    }
    return acc;
}

// tiles (4 x f32x4 in C layout) -> two k-blocks of permuted-k B operands, hi and lo
DEVFN void tiles_to_b(const f32x4* tl, float scale, bf16x8* bh, bf16x8* bl) {
    uint2 h[4], l[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x4 x = tl[b];
        x[0] *= scale; x[1] *= scale; x[2] *= scale; x[3] *= scale;
        split4(x, h[b], l[b]);
    }
    bh[0] = mk8(h[0], h[1]); bh[1] = mk8(h[2], h[3]);
    bl[0] = mk8(l[0], l[1]); bl[1] = mk8(l[2], l[3]);
}

// acc += A[row][perm k] * B(regs) with A hi/lo in LDS ([row][64] images), B hi/lo from tiles
template <bool ALO>
DEVFN f32x4 mm_perm(f32x4 acc, const uint16_t (*Ah_)[TJ], const uint16_t (*Al_)[TJ], int row, int g,
                    const bf16x8* bh, const bf16x8* bl) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 ah = ld_perm(Ah_, row, kb, g);
        acc = mfma_16x16x32_bf16(ah, bh[kb], acc);
        acc = mfma_16x16x32_bf16(ah, bl[kb], acc);
        if (ALO) acc = mfma_16x16x32_bf16(ld_perm(Al_, row, kb, g), bh[kb], acc);
    }
    return acc;
}

// acc += M[row][s] * B[s][col] for a 16x16 score-type A (hi/lo images in LDS) and a B given as (hi, lo) dwords
DEVFN f32x4 mm_small(f32x4 acc, const uint16_t (*Mh)[SS], const uint16_t (*Ml)[SS], int row, int g, uint2 bh, uint2 bl) {
    const uint2 mh = ld8(&Mh[row][4 * g]), ml = ld8(&Ml[row][4 * g]);
    acc = mfma_16x16x32_bf16(mk8(mh, mh), mk8(bh, bl), acc);
    return mfma_16x16x32_bf16(mk8(ml.x, ml.y, 0u, 0u), mk8(bh.x, bh.y, 0u, 0u), acc);
}
// same with an exact-bf16 B:  [M_hi | M_lo] x [B ; B]
DEVFN f32x4 mm_small_exact(f32x4 acc, const uint16_t (*Mh)[SS], const uint16_t (*Ml)[SS], int row, int g, uint2 b) {
    return mfma_16x16x32_bf16(mk8(ld8(&Mh[row][4 * g]), ld8(&Ml[row][4 * g])), mk8(b, b), acc);
}

DEVFN void store_dscore(LdsB& lds, int slot, f32x4 d, int c16, int g, int mode) {
    // mode 0: keep c16 <  4g+r   1: keep 4g+r <  c16   2: keep c16 <= 4g+r   3: keep 4g+r <= c16
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int x = 4 * g + r;
        const bool keep = mode == 0 ? (c16 < x) : mode == 1 ? (x < c16) : mode == 2 ? (c16 <= x) : (x <= c16);
        d[r] = keep ? d[r] : 0.f;
    }
    uint2 h, l;
    split4(d, h, l);
    st8(&lds.dsc[slot][0][c16][4 * g], h);
    st8(&lds.dsc[slot][1][c16][4 * g], l);
}

template <bool PROF>
__global__ __launch_bounds__(256) void bwd_kernel_t(BwdArgs p) {
    LdsB& lds = *reinterpret_cast<LdsB*>(dyn_lds());
    WKV_STAMP_DECL
    const int T = p.T, H = p.H;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c16 = lane & 15, g = lane >> 4;
    const size_t head_base = ((size_t)b * T * H + h) * N;
    const size_t tstride = (size_t)H * N;
    const int nchunk = T / L;
    const int c0 = 16 * wave + 4 * g;                       // phase-1 column group of this lane
    const size_t prep_off = head_base + (size_t)c16 * tstride + c0;
    const float* sbase = p.s + (size_t)blockIdx.x * nchunk * N * N;

    f32x4 dS1[4], dS2[4], SL[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { dS1[x] = zero4(); dS2[x] = zero4(); }
    // S_L of the last chunk, as S[i][j] tiles: SL[ib][r] = S[i = 16ib+4g+r][j = 16*wave + c16]
    {
        const float* sp = sbase + (size_t)(nchunk - 1) * N * N + (size_t)(16 * wave + c16) * N + 4 * g;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float4 x = *reinterpret_cast<const float4*>(sp + 16 * ib);
            SL[ib][0] = x.x; SL[ib][1] = x.y; SL[ib][2] = x.z; SL[ib][3] = x.w;
        }
    }

    // inputs of the chunk being processed are fetched one chunk ahead (registers)
    struct Raw { uint2 w, q, k, z, a, v, dy; float4 sa; } raw;
    f32x4 S0n[4];
    auto fetch = [&](int c) {
        const size_t off = prep_off + (size_t)c * L * tstride;
        raw.w = *reinterpret_cast<const uint2*>(p.w + off); raw.q = *reinterpret_cast<const uint2*>(p.q + off);
        raw.k = *reinterpret_cast<const uint2*>(p.k + off); raw.z = *reinterpret_cast<const uint2*>(p.z + off);
        raw.a = *reinterpret_cast<const uint2*>(p.a + off); raw.v = *reinterpret_cast<const uint2*>(p.v + off);
        raw.dy = *reinterpret_cast<const uint2*>(p.dy + off); raw.sa = *reinterpret_cast<const float4*>(p.sa + off);
        if (c > 0) {          // chunk-start state S0 = s[c-1] as S[i][j] tiles (zero for the first chunk)
            const float* sp = sbase + (size_t)(c - 1) * N * N + (size_t)(16 * wave + c16) * N + 4 * g;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float4 x = *reinterpret_cast<const float4*>(sp + 16 * ib);
                S0n[ib][0] = x.x; S0n[ib][1] = x.y; S0n[ib][2] = x.z; S0n[ib][3] = x.w;
            }
        } else {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) S0n[ib] = zero4();
        }
    };
    fetch(nchunk - 1);

    for (int c = nchunk - 1; c >= 0; --c) {
        const size_t coff = prep_off + (size_t)c * L * tstride;
        // ------------------------------------------------------------ phase 1: operands
        float q[4], k[4], z[4], a[4], lw[4], cc[4], cp[4], ic[4];
        f32x4 S0[4];
        {
            float wr[4];
            unpack4(raw.w, wr); unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
            const uint2 rv = raw.v;
            const uint2 rdy = raw.dy;
            const float4 rsa = raw.sa;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) S0[ib] = S0n[ib];
            float zt[4], qt[4], ah[4], kh[4], ab[4], kb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lw[e] = -fast_exp(wr[e]);
                float x = lw[e];
                x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
                const float tot = lane_bcast(x, (lane & 48) | 15);
                cc[e] = fast_exp(x); cp[e] = fast_exp(x - lw[e]); ic[e] = fast_exp(-x);
                const float cb = fast_exp(tot - x);
                zt[e] = z[e] * cp[e]; qt[e] = q[e] * cc[e]; ah[e] = a[e] * ic[e]; kh[e] = k[e] * ic[e];
                ab[e] = a[e] * cb; kb[e] = k[e] * cb;
            }
            uint2 hh, ll;
            split4(zt, hh, ll); st8(&lds.opnd[0][c16][c0], hh); st8(&lds.opnd[1][c16][c0], ll);
            st_b16x4_T(lds.trn[0], c0, c16, hh); st_b16x4_T(lds.trn[1], c0, c16, ll);
            split4(qt, hh, ll); st8(&lds.opnd[2][c16][c0], hh); st8(&lds.opnd[3][c16][c0], ll);
            st_b16x4_T(lds.trn[2], c0, c16, hh); st_b16x4_T(lds.trn[3], c0, c16, ll);
            split4(ah, hh, ll); st8(&lds.opnd[4][c16][c0], hh); st8(&lds.opnd[5][c16][c0], ll);
            st_b16x4_T(lds.trn[4], c0, c16, hh); st_b16x4_T(lds.trn[5], c0, c16, ll);
            split4(kh, hh, ll); st8(&lds.opnd[6][c16][c0], hh); st8(&lds.opnd[7][c16][c0], ll);
            st_b16x4_T(lds.trn[6], c0, c16, hh); st_b16x4_T(lds.trn[7], c0, c16, ll);
            split4(ab, hh, ll); st8(&lds.opnd[8][c16][c0], hh); st8(&lds.opnd[9][c16][c0], ll);
            split4(kb, hh, ll); st8(&lds.opnd[10][c16][c0], hh); st8(&lds.opnd[11][c16][c0], ll);
            st8(&lds.ti[0][c16][c0], rv);
            st8(&lds.ti[1][c16][c0], rdy);
            st_b16x4_T(lds.it[0], c0, c16, rdy);
            const float sav[4] = {rsa.x, rsa.y, rsa.z, rsa.w};
            split4(sav, hh, ll); st8(&lds.ti[2][c16][c0], hh); st8(&lds.ti[3][c16][c0], ll);
            if (c16 == 15) *reinterpret_cast<float4*>(&lds.cl[c0]) = make_float4(cc[0], cc[1], cc[2], cc[3]);
        }
        if (c > 0) fetch(c - 1);            // prefetch: consumed in the next iteration
        WKV_STAMP(0)
        block_sync_lds();
        WKV_STAMP(1)

        // ------------------------------------------------------------ phase 2: scores (one matrix per wave)
        if (wave == 1) {          // (Qt Ah^T)[t][s] held as lane c16 = s, r <-> t: A image of M_qa^T
            f32x4 d = dot64<true, true>(lds.opnd[2], lds.opnd[3], lds.opnd[4], lds.opnd[5], c16, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = (c16 <= 4 * g + r) ? d[r] : 0.f;
            uint2 hh, ll; split4(d, hh, ll);
            st8(&lds.sc[0][0][c16][4 * g], hh); st8(&lds.sc[0][1][c16][4 * g], ll);
        } else if (wave == 2) {   // M_qk^T
            f32x4 d = dot64<true, true>(lds.opnd[2], lds.opnd[3], lds.opnd[6], lds.opnd[7], c16, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = (c16 <= 4 * g + r) ? d[r] : 0.f;
            uint2 hh, ll; split4(d, hh, ll);
            st8(&lds.sc[1][0][c16][4 * g], hh); st8(&lds.sc[1][1][c16][4 * g], ll);
        } else if (wave == 3) {   // M_zk^T (strict)
            f32x4 d = dot64<true, true>(lds.opnd[0], lds.opnd[1], lds.opnd[6], lds.opnd[7], c16, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = (c16 < 4 * g + r) ? d[r] : 0.f;
            uint2 hh, ll; split4(d, hh, ll);
            st8(&lds.sc[2][0][c16][4 * g], hh); st8(&lds.sc[2][1][c16][4 * g], ll);
        } else {                  // T = (I - M_za)^-1 in C layout = A image of T^T
            f32x4 X = dot64<true, true>(lds.opnd[0], lds.opnd[1], lds.opnd[4], lds.opnd[5], c16, g);   // [t][s]
            f32x4 XT = dot64<true, true>(lds.opnd[4], lds.opnd[5], lds.opnd[0], lds.opnd[1], c16, g);  // [s][t]
            f32x4 Tc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[r] = (c16 < 4 * g + r) ? X[r] : 0.f;
                XT[r] = (4 * g + r < c16) ? XT[r] : 0.f;
                Tc[r] = X[r] + ((4 * g + r == c16) ? 1.f : 0.f);
            }
#pragma unroll
            for (int level = 0; level < 3; ++level) {
                const f32x4 XTn = regmm(X, XT);                 // (X^T)^2
                f32x4 Xn = X;
                if (level < 2) Xn = regmm(XT, X);               // X^2
                const f32x4 D = regmm(XTn, Tc);                 // X_k T
#pragma unroll
                for (int r = 0; r < 4; ++r) Tc[r] += D[r];
                X = Xn; XT = XTn;
            }
            uint2 hh, ll; split4(Tc, hh, ll);                   // Tc[r] = T[4g+r][c16] = T^T[c16][4g+r]
            st8(&lds.sc[3][0][c16][4 * g], hh); st8(&lds.sc[3][1][c16][4 * g], ll);
        }
        WKV_STAMP(2)
        block_sync_lds();
        WKV_STAMP(3)

        // ------------------------------------------------------------ phase 3: i-split products (i = 16*wave + c16)
        {
            bf16x8 bh[2], bl[2];
            tiles_to_b(dS1, 1.f, bh, bl);
            const uint2 dy = ld8(&lds.it[0][16 * wave + c16][4 * g]);
            f32x4 dSA = mm_small_exact(zero4(), lds.sc[0][0], lds.sc[0][1], c16, g, dy);          // M_qa^T dY
            dSA = mm_perm<true>(dSA, lds.opnd[8], lds.opnd[9], c16, g, bh, bl);                   // Ab dS^T
            uint2 xh, xl;
            split4(dSA, xh, xl);
            f32x4 dR = mm_small(zero4(), lds.sc[3][0], lds.sc[3][1], c16, g, xh, xl);             // T^T dSA
            uint2 rh, rl;
            split4(dR, rh, rl);
            f32x4 dV = mm_small_exact(zero4(), lds.sc[1][0], lds.sc[1][1], c16, g, dy);           // M_qk^T dY
            dV = mm_perm<true>(dV, lds.opnd[10], lds.opnd[11], c16, g, bh, bl);                   // Kb dS^T
            dV = mm_small(dV, lds.sc[2][0], lds.sc[2][1], c16, g, rh, rl);                        // M_zk^T dR
            const size_t o = head_base + (size_t)(c * L + 4 * g) * tstride + 16 * wave + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) p.dv[o + r * tstride] = (uint16_t)f32_to_bf16_bits(dV[r]);
            // dR for the j-split phase: [t][i] and [i][t] images
            st_b16x4_col(lds.ti[4], 4 * g, 16 * wave + c16, rh);
            st_b16x4_col(lds.ti[5], 4 * g, 16 * wave + c16, rl);
            st8(&lds.it[1][16 * wave + c16][4 * g], rh);
            st8(&lds.it[2][16 * wave + c16][4 * g], rl);
            // dS^T <- diag(c_L) dS^T + [Qt^T | Zt^T] [dY ; dR]
            const bf16x8 b1 = mk8(dy, rh), b2 = mk8(0u, 0u, rl.x, rl.y);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float4 cl = *reinterpret_cast<const float4*>(&lds.cl[16 * jb + 4 * g]);
                f32x4 acc = dS1[jb];
                acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
                const int j = 16 * jb + c16;
                const bf16x8 ah = mk8(ld8(&lds.trn[2][j][4 * g]), ld8(&lds.trn[0][j][4 * g]));
                const bf16x8 al = mk8(ld8(&lds.trn[3][j][4 * g]), ld8(&lds.trn[1][j][4 * g]));
                acc = mfma_16x16x32_bf16(ah, b1, acc);
                acc = mfma_16x16x32_bf16(ah, b2, acc);
                acc = mfma_16x16x32_bf16(al, b1, acc);
                dS1[jb] = acc;
            }
        }
        WKV_STAMP(4)
        block_sync_lds();
        WKV_STAMP(5)

        // ------------------------------------------------------------ phase 4: score gradients (one matrix per wave)
        if (wave == 0) {          // dM_za = tril_(dR SA^T)
            store_dscore(lds, 1, dot64<true, true>(lds.ti[4], lds.ti[5], lds.ti[2], lds.ti[3], c16, g), c16, g, 0);
            store_dscore(lds, 0, dot64<true, true>(lds.ti[2], lds.ti[3], lds.ti[4], lds.ti[5], c16, g), c16, g, 1);
        } else if (wave == 1) {   // dM_zk = tril_(dR V^T)
            store_dscore(lds, 3, dot64<true, false>(lds.ti[4], lds.ti[5], lds.ti[0], lds.ti[0], c16, g), c16, g, 0);
            store_dscore(lds, 2, dot64<false, true>(lds.ti[0], lds.ti[0], lds.ti[4], lds.ti[5], c16, g), c16, g, 1);
        } else if (wave == 2) {   // dM_qa = tril(dY SA^T)
            store_dscore(lds, 5, dot64<false, true>(lds.ti[1], lds.ti[1], lds.ti[2], lds.ti[3], c16, g), c16, g, 2);
            store_dscore(lds, 4, dot64<true, false>(lds.ti[2], lds.ti[3], lds.ti[1], lds.ti[1], c16, g), c16, g, 3);
        } else {                  // dM_qk = tril(dY V^T)
            store_dscore(lds, 7, dot64<false, false>(lds.ti[1], lds.ti[1], lds.ti[0], lds.ti[0], c16, g), c16, g, 2);
            store_dscore(lds, 6, dot64<false, false>(lds.ti[0], lds.ti[0], lds.ti[1], lds.ti[1], c16, g), c16, g, 3);
        }
        WKV_STAMP(6)
        block_sync_lds();
        WKV_STAMP(7)

        // ------------------------------------------------------------ phase 5: j-split products (j = 16*wave + c16)
        {
            const int j = 16 * wave + c16;
            const float clj = lds.cl[j];
            bf16x8 s0h[2], s0l[2], duh[2], dul[2];
            tiles_to_b(S0, 1.f, s0h, s0l);
            tiles_to_b(dS2, clj, duh, dul);
            const uint2 zth = ld8(&lds.trn[0][j][4 * g]), ztl = ld8(&lds.trn[1][j][4 * g]);
            const uint2 qth = ld8(&lds.trn[2][j][4 * g]), qtl = ld8(&lds.trn[3][j][4 * g]);
            const uint2 ahh = ld8(&lds.trn[4][j][4 * g]), ahl = ld8(&lds.trn[5][j][4 * g]);
            const uint2 khh = ld8(&lds.trn[6][j][4 * g]), khl = ld8(&lds.trn[7][j][4 * g]);

            f32x4 dZt = mm_perm<true>(zero4(), lds.ti[4], lds.ti[5], c16, g, s0h, s0l);            // dR S0
            dZt = mm_small(dZt, lds.dsc[0][0], lds.dsc[0][1], c16, g, ahh, ahl);                    // dM_za Ah
            dZt = mm_small(dZt, lds.dsc[2][0], lds.dsc[2][1], c16, g, khh, khl);                    // dM_zk Kh
            f32x4 dQt = mm_perm<false>(zero4(), lds.ti[1], lds.ti[1], c16, g, s0h, s0l);           // dY S0
            dQt = mm_small(dQt, lds.dsc[4][0], lds.dsc[4][1], c16, g, ahh, ahl);                    // dM_qa Ah
            dQt = mm_small(dQt, lds.dsc[6][0], lds.dsc[6][1], c16, g, khh, khl);                    // dM_qk Kh
            f32x4 dAh = mm_perm<true>(zero4(), lds.ti[2], lds.ti[3], c16, g, duh, dul);            // SA dU
            dAh = mm_small(dAh, lds.dsc[1][0], lds.dsc[1][1], c16, g, zth, ztl);                    // dM_za^T Zt
            dAh = mm_small(dAh, lds.dsc[5][0], lds.dsc[5][1], c16, g, qth, qtl);                    // dM_qa^T Qt
            f32x4 dKh = mm_perm<false>(zero4(), lds.ti[0], lds.ti[0], c16, g, duh, dul);           // V dU
            dKh = mm_small(dKh, lds.dsc[3][0], lds.dsc[3][1], c16, g, zth, ztl);                    // dM_zk^T Zt
            dKh = mm_small(dKh, lds.dsc[7][0], lds.dsc[7][1], c16, g, qth, qtl);                    // dM_qk^T Qt

            // sum_i dS_L[i][j] S_L[i][j]
            float gl = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) gl = fmaf(dS2[ib][r], SL[ib][r], gl);
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            if (g == 0) lds.glast[j] = gl;

            // dS <- dS diag(c_L) + [dY^T | dR^T] [Qt ; Zt]
            const bf16x8 bqh = mk8(qth, zth), bql = mk8(qtl, ztl);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                f32x4 acc = dS2[ib];
                acc[0] *= clj; acc[1] *= clj; acc[2] *= clj; acc[3] *= clj;
                const int i = 16 * ib + c16;
                const bf16x8 ah = mk8(ld8(&lds.it[0][i][4 * g]), ld8(&lds.it[1][i][4 * g]));
                const uint2 rl = ld8(&lds.it[2][i][4 * g]);
                acc = mfma_16x16x32_bf16(ah, bqh, acc);
                acc = mfma_16x16x32_bf16(ah, bql, acc);
                acc = mfma_16x16x32_bf16(mk8(0u, 0u, rl.x, rl.y), bqh, acc);
                dS2[ib] = acc;
                SL[ib] = S0[ib];                              // S_L of the next (earlier) chunk
            }
            // results back to [t][j] form (rows 4g+r, column j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lds.res[0][4 * g + r][j] = dZt[r];
                lds.res[1][4 * g + r][j] = dQt[r];
                lds.res[2][4 * g + r][j] = dAh[r];
                lds.res[3][4 * g + r][j] = dKh[r];
            }
        }
        WKV_STAMP(8)
        wave_lds_fence();
        // ------------------------------------------------------------ element-wise tail (lane: token c16, columns c0..c0+3)
        {
            const float4 rz = *reinterpret_cast<const float4*>(&lds.res[0][c16][c0]);
            const float4 rq = *reinterpret_cast<const float4*>(&lds.res[1][c16][c0]);
            const float4 ra = *reinterpret_cast<const float4*>(&lds.res[2][c16][c0]);
            const float4 rk = *reinterpret_cast<const float4*>(&lds.res[3][c16][c0]);
            const float4 gl4 = *reinterpret_cast<const float4*>(&lds.glast[c0]);
            const float dzt[4] = {rz.x, rz.y, rz.z, rz.w}, dqt[4] = {rq.x, rq.y, rq.z, rq.w};
            const float dah[4] = {ra.x, ra.y, ra.z, ra.w}, dkh[4] = {rk.x, rk.y, rk.z, rk.w};
            const float glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
            float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dz[e] = dzt[e] * cp[e]; dq[e] = dqt[e] * cc[e]; da[e] = dah[e] * ic[e]; dk[e] = dkh[e] * ic[e];
                float gt = dq[e] * q[e] - da[e] * a[e] - dk[e] * k[e] + dpp_shl<1>(dz[e] * z[e]);
                if (c16 == 15) gt += glv[e];
                gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
                dw[e] = gt * lw[e];
            }
            *reinterpret_cast<uint2*>(p.dw + coff) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
            *reinterpret_cast<uint2*>(p.dq + coff) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
            *reinterpret_cast<uint2*>(p.dk + coff) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
            *reinterpret_cast<uint2*>(p.dz + coff) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
            *reinterpret_cast<uint2*>(p.da + coff) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
        }
        WKV_STAMP(9)
        block_sync_lds();
        WKV_STAMP(10)
    }
    WKV_STAMP_FLUSH(0, 0, 11)
}

}  // namespace wkv7c
