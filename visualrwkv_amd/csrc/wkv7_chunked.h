// WKV7 forward in chunked (matmul) form on bf16 MFMA with split operands -- gfx950.
//
// Same operator as wkv7_kernels.h (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-52) but the
// 16 tokens of a checkpoint chunk are processed together so that all work is dense 16x16 tiles
// (oracle/wkv7_chunked.py is the CPU statement of this algorithm and its validation):
//     c_t = prod_{r<=t} w_r ;  Zt = z*c_{t-1}  Qt = q*c_t  Ah = a/c_t  Kh = k/c_t
//     M_zk = tril_(Zt Kh^T)  M_qa = tril(Qt Ah^T)  M_qk = tril(Qt Kh^T)  T = (I - tril_(Zt Ah^T))^-1
//     SA = T (Zt S0^T + M_zk V) ;  Y = Qt S0^T + M_qa SA + M_qk V
//     S_L = S0 diag(c_L) + SA^T (Ah c_L) + V^T (Kh c_L)
// Precision: every fp32 operand x is split into bf16 hi = rn(x), lo = rn(x - hi) and a product is
// hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation ("bf16x3", ~2^-16 relative);
// plain bf16 products miss the 1e-3 parity bar, bf16x3 lands at ~1.5e-4 (tests/test_chunked_numerics.py).
//
// Work split: one workgroup (4 waves) per (b,h).
//   phase 1  wave w prepares key-columns j in [16w,16w+16): decay scan over t (DPP row scan), scaled
//            operands, hi/lo split, written to LDS in [t][j] and (for the state update) [j][t] form;
//   phase 2  wave 1..3 build one masked score matrix each, wave 0 builds T by nilpotent doubling
//            (T = (I+M)(I+M^2)(I+M^4)(I+M^8)), all register-resident;
//   phase 3  wave w owns value-columns i in [16w,16w+16) of S^T, held as 4 MFMA accumulator tiles.
//            The accumulators are fed straight back as B operands: the contraction index of an MFMA
//            is only a label, so the k-slots are permuted to match the C/D register map (no LDS
//            round trip, no cross-lane traffic on the S -> R -> SA -> Y -> S chain).
#pragma once
#include <gfx950_prims.h>
#include <wkv7_kernels.h>

namespace wkv7c {

using wkv7::FwdArgs;
constexpr int N = 64, L = 16;
constexpr int TJ = 68;   // [t][j] row stride in bf16 elements (136 B: conflict-free 8-byte column reads).  72 (144 B) looks
                         // better on paper for the 16-byte reads but measured 4-12 % slower in all five kernels.
constexpr int JT = 24;   // [j][t] row stride (48 B)
constexpr int SS = 24;   // [t][s] row stride (48 B)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
DEVFN bf16x8 mk8(uint2 lo, uint2 hi) { return mk8(lo.x, lo.y, hi.x, hi.y); }
DEVFN f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

struct Lds {
    uint16_t opnd[8][L][TJ];    // 0 Zt_hi 1 Zt_lo 2 Qt_hi 3 Qt_lo 4 Ah_hi 5 Ah_lo 6 Kh_hi 7 Kh_lo   [t][j]
    uint16_t trn[4][N][JT];     // 0 Ab_hi 1 Ab_lo 2 Kb_hi 3 Kb_lo  (Ab = a*c_L/c_t, Kb = k*c_L/c_t)  [j][t]
    uint16_t vt[N][JT];         // v (exact bf16)                                                      [i][t]
    uint16_t sc[4][2][L][SS];   // 0 M_zk 1 M_qa 2 M_qk 3 T ; [hi,lo][t][s]
    float cl[N];                // c_L[j]
};

struct RawChunk { uint2 w, q, k, z, a, v; };

DEVFN uint2 ld8(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
DEVFN void st8(uint16_t* p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }

// split 4 consecutive values -> packed hi (2 dwords) and lo (2 dwords)
DEVFN void split4(const float* x, uint2& hi, uint2& lo) {
    split_pk(x[0], x[1], hi.x, lo.x);
    split_pk(x[2], x[3], hi.y, lo.y);
}
DEVFN void split4(f32x4 x, uint2& hi, uint2& lo) {
    split_pk(x[0], x[1], hi.x, lo.x);
    split_pk(x[2], x[3], hi.y, lo.y);
}

DEVFN void load_chunk(RawChunk& rc, const FwdArgs& p, size_t off) {
    rc.w = *reinterpret_cast<const uint2*>(p.w + off);
    rc.q = *reinterpret_cast<const uint2*>(p.q + off);
    rc.k = *reinterpret_cast<const uint2*>(p.k + off);
    rc.z = *reinterpret_cast<const uint2*>(p.z + off);
    rc.a = *reinterpret_cast<const uint2*>(p.a + off);
    rc.v = *reinterpret_cast<const uint2*>(p.v + off);
}

DEVFN void unpack4(uint2 u, float* f) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
}

// phase 1: lane (t = lane&15, g = lane>>4) of wave w handles token t, columns j0..j0+3, j0 = 16w+4g
DEVFN void prep(Lds& lds, const RawChunk& rc, int wave, int lane) {
    const int t = lane & 15, g = lane >> 4, j0 = 16 * wave + 4 * g;
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(rc.w, wr); unpack4(rc.q, q); unpack4(rc.k, k); unpack4(rc.z, z); unpack4(rc.a, a);
    float zt[4], qt[4], ah[4], kh[4], ab[4], kb[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp(wr[e]);               // log w_t          (wkv7_cuda.cu:21)
        float x = lw;                                    // inclusive scan over t inside the 16-lane row
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        const float tot = lane_bcast(x, (lane & 48) | 15);   // log c_L
        const float c = fast_exp(x), cp = fast_exp(x - lw), ic = fast_exp(-x), cb = fast_exp(tot - x);
        zt[e] = z[e] * cp; qt[e] = q[e] * c; ah[e] = a[e] * ic; kh[e] = k[e] * ic;
        ab[e] = a[e] * cb; kb[e] = k[e] * cb; cend[e] = c;
    }
    uint2 h, l;
    split4(zt, h, l); st8(&lds.opnd[0][t][j0], h); st8(&lds.opnd[1][t][j0], l);
    split4(qt, h, l); st8(&lds.opnd[2][t][j0], h); st8(&lds.opnd[3][t][j0], l);
    split4(ah, h, l); st8(&lds.opnd[4][t][j0], h); st8(&lds.opnd[5][t][j0], l);
    split4(kh, h, l); st8(&lds.opnd[6][t][j0], h); st8(&lds.opnd[7][t][j0], l);
    split4(ab, h, l);
    lds.trn[0][j0 + 0][t] = (uint16_t)h.x; lds.trn[0][j0 + 1][t] = (uint16_t)(h.x >> 16);
    lds.trn[0][j0 + 2][t] = (uint16_t)h.y; lds.trn[0][j0 + 3][t] = (uint16_t)(h.y >> 16);
    lds.trn[1][j0 + 0][t] = (uint16_t)l.x; lds.trn[1][j0 + 1][t] = (uint16_t)(l.x >> 16);
    lds.trn[1][j0 + 2][t] = (uint16_t)l.y; lds.trn[1][j0 + 3][t] = (uint16_t)(l.y >> 16);
    split4(kb, h, l);
    lds.trn[2][j0 + 0][t] = (uint16_t)h.x; lds.trn[2][j0 + 1][t] = (uint16_t)(h.x >> 16);
    lds.trn[2][j0 + 2][t] = (uint16_t)h.y; lds.trn[2][j0 + 3][t] = (uint16_t)(h.y >> 16);
    lds.trn[3][j0 + 0][t] = (uint16_t)l.x; lds.trn[3][j0 + 1][t] = (uint16_t)(l.x >> 16);
    lds.trn[3][j0 + 2][t] = (uint16_t)l.y; lds.trn[3][j0 + 3][t] = (uint16_t)(l.y >> 16);
    lds.vt[j0 + 0][t] = (uint16_t)rc.v.x; lds.vt[j0 + 1][t] = (uint16_t)(rc.v.x >> 16);
    lds.vt[j0 + 2][t] = (uint16_t)rc.v.y; lds.vt[j0 + 3][t] = (uint16_t)(rc.v.y >> 16);
    if (t == 15) *reinterpret_cast<float4*>(&lds.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}

// natural-k operand: row `row`, elements j = 32kb + 8g + e
DEVFN bf16x8 ld_nat(const uint16_t (*M)[TJ], int row, int kb, int g) {
    const uint16_t* p = &M[row][32 * kb + 8 * g];
    return mk8(ld8(p), ld8(p + 4));
}
// D[x_row][y_row] = sum_j X[x_row][j] Y[y_row][j]   (bf16x3); lane supplies row c16 of both operands
DEVFN f32x4 score(const Lds& lds, int mx, int my, int c16, int g) {
    f32x4 acc = zero4();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 xh = ld_nat(lds.opnd[mx], c16, kb, g), xl = ld_nat(lds.opnd[mx + 1], c16, kb, g);
        const bf16x8 yh = ld_nat(lds.opnd[my], c16, kb, g), yl = ld_nat(lds.opnd[my + 1], c16, kb, g);
        acc = mfma_16x16x32_bf16(xh, yh, acc);
        acc = mfma_16x16x32_bf16(xh, yl, acc);
        acc = mfma_16x16x32_bf16(xl, yh, acc);
    }
    return acc;
}
// D = P*Q for 16x16 register matrices: pt = P^T in C layout (acts as the A operand), qc = Q in C layout
DEVFN f32x4 regmm(f32x4 pt, f32x4 qc) {
    uint2 ph, pl, qh, ql;
    split4(pt, ph, pl); split4(qc, qh, ql);
    f32x4 acc = mfma_16x16x32_bf16(mk8(ph, ph), mk8(qh, ql), zero4());
    return mfma_16x16x32_bf16(mk8(pl.x, pl.y, 0u, 0u), mk8(qh.x, qh.y, 0u, 0u), acc);
}
// masked transposed score -> A-operand layout in LDS: D[s][t] held as lane (g, c16 = t), reg r <-> s = 4g+r
DEVFN void store_score(Lds& lds, int slot, f32x4 d, int c16, int g, bool inclusive) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s = 4 * g + r;
        const bool keep = inclusive ? (s <= c16) : (s < c16);
        d[r] = keep ? d[r] : 0.f;
    }
    uint2 h, l;
    split4(d, h, l);
    st8(&lds.sc[slot][0][c16][4 * g], h);
    st8(&lds.sc[slot][1][c16][4 * g], l);
}

DEVFN void scores(Lds& lds, int wave, int lane) {
    const int c16 = lane & 15, g = lane >> 4;
    if (wave == 1) {
        store_score(lds, 0, score(lds, 6, 0, c16, g), c16, g, false);     // (Kh Zt^T)[s][t] = M_zk[t][s], s <  t
    } else if (wave == 2) {
        store_score(lds, 1, score(lds, 4, 2, c16, g), c16, g, true);      // (Ah Qt^T)[s][t] = M_qa[t][s], s <= t
    } else if (wave == 3) {
        store_score(lds, 2, score(lds, 6, 2, c16, g), c16, g, true);      // (Kh Qt^T)[s][t] = M_qk[t][s], s <= t
    } else {
        // X = M_za (lane (g, c16 = s), reg r <-> t = 4g+r), XT = M_za^T (lane (g, c16 = t), reg r <-> s = 4g+r)
        f32x4 X = score(lds, 0, 4, c16, g), XT = score(lds, 4, 0, c16, g), TT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r] = (c16 < 4 * g + r) ? X[r] : 0.f;
            XT[r] = (4 * g + r < c16) ? XT[r] : 0.f;
            TT[r] = XT[r] + ((4 * g + r == c16) ? 1.f : 0.f);            // (I + M)^T
        }
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const f32x4 X2 = regmm(XT, X);                               // X*X
            f32x4 XT2 = XT;
            if (level < 2) XT2 = regmm(X, XT);                           // X^T*X^T
            const f32x4 D = regmm(X2, TT);                               // (T X2)^T = X2^T T^T
#pragma unroll
            for (int r = 0; r < 4; ++r) TT[r] += D[r];
            X = X2; XT = XT2;
        }
        uint2 h, l;                                                      // TT[r] = T[c16][4g+r]
        split4(TT, h, l);
        st8(&lds.sc[3][0][c16][4 * g], h);
        st8(&lds.sc[3][1][c16][4 * g], l);
    }
}

// permuted-k operand for products against the S accumulators:
// slot (g, e<4) <-> j = 32kb + 4g + e ; slot (g, e>=4) <-> j = 32kb + 16 + 4g + (e-4)
DEVFN bf16x8 ld_perm(const uint16_t (*M)[TJ], int row, int kb, int g) {
    const uint16_t* p = &M[row][32 * kb + 4 * g];
    return mk8(ld8(p), ld8(p + 16));
}

// PROF: per-phase shader-clock deltas are accumulated in registers (s_memtime is scalar) and written to
// p.dbg once at kernel end by WKV_STAMP_FLUSH -- a stamp that touched memory would bill its own latency to the
// next phase.
#define WKV_STAMP_DECL unsigned long long tprev_ = PROF ? clock64_() : 0ull, tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WKV_STAMP(slot)                                                        \
    if (PROF) {                                                                \
        const unsigned long long now_ = clock64_();                            \
        tacc_[slot] += now_ - tprev_;                                          \
        tprev_ = now_;                                                         \
    }
#define WKV_STAMP_FLUSH(who, base, n)                                          \
    if (PROF && blockIdx.x == 0 && tid == (who)) {                             \
        for (int i_ = 0; i_ < (n); ++i_) p.dbg[(base) + i_] = tacc_[i_];       \
    }



}  // namespace wkv7c
