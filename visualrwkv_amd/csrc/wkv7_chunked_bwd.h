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


}  // namespace wkv7c
