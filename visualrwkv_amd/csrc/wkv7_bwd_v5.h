// WKV7 backward, chunked MFMA form, second-generation schedule -- gfx950.
//
// Closed-form differentiation of a 16-token chunk from S0 = s[c-1] and the saved sa (oracle/wkv7_chunked.py::backward is
// the CPU statement, validated to 1e-15 against fp64 autograd; replaces the token-by-token un-stepping of the reference,
// VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130).  With the forward quantities of wkv7_chunked.h and dS = dL/dS_L entering:
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
// with a producer/consumer split
// (8 waves per (b,h), three barrier-delimited segments per chunk).  Its layout choices were driven by the round-1 counters of
// its predecessor (profiles/r1_wkv7_pmc_b16.txt: VALU:MFMA = 10.5:1, a third of the VALU instructions register moves that
// assemble MFMA operands from two LDS reads, LDS pipe 45 % busy with 22 % bank conflicts, 100 two-byte LDS scatter
// stores per chunk for transposed operand copies) and by its phase stamps (j-split 3.0k of 9k cycles per chunk, half of
// it a vmcnt wait: a wave that both prefetches S0 and stores gradients has to drain its stores before it may use the
// load, because loads and stores share one counter on gfx9 and return out of order with respect to each other):
//
//  * mfma(X, Y) computes D[m][n] = sum_k X(m,k) Y(n,k) with BOTH operands supplied "one row per lane, eight k per
//    lane", so swapping the operands transposes the result.  Every product whose result feeds the element-wise tail
//    (dZt dQt dAh dKh) or a global store (dV) is issued with the operands swapped: the accumulators come out as
//    "token = lane, 4 consecutive channels = registers", which IS the tail / store layout.  The LDS bounce of v3
//    (20 ds_write_b32 + 6 ds_read_b128 per wave and chunk) and the four 2-byte global stores of dV are gone.
//  * Operand images are [16 tokens][64 channels] bf16, 128-byte rows, no padding, XOR-swizzled in 16-byte slots
//    (slot ^= row & 7): a row read of 8 consecutive k is ONE conflict-free ds_read_b128.  State tiles keep their rows
//    in the interleaved order tix() so that two tiles side by side are 8 consecutive channels -- the permuted-k reads
//    (ds_read2_b64, half the LDS rate) and their register shuffles disappear.
//  * Operands whose k index is the token are read with ds_read_b64_tr_b16 from the same row-major images: no transposed
//    copies, no 2-byte scatter stores.
//  * Half-depth (16-token) products never need zero-filled or duplicated registers: the zero / duplicate halves live in
//    the LDS images ("DZ" images [h|h],[l|0] for T and M_zk; pair images [za|zk],[qa|qk] for the score gradients, which
//    fuse two 16-deep products into one K=32 MFMA) or in one operand built once per chunk.  The legacy K=16 bf16 MFMA is
//    NOT used: mixed into an accumulation chain with K=32 MFMAs it returned wrong sums on gfx950 with ROCm 7.2 (missing
//    wait states between the two instruction classes; found with the register dump against the host emulator).
//  * The consumers issue no global loads inside the loop: the producers bring S0 = s[c-1] into LDS with LDS-DMA
//    (global_load_lds_dwordx4, source pre-swizzled), so the consumers' vmcnt only ever counts stores and is never waited.
//  * The decay factors of the tail come from the producers' log2-domain scan through LDS (two fp32 images) instead of
//    being recomputed from a second global read of w; S_L is no longer carried (the decay-gradient term sum_i dS.S_L is
//    formed one chunk early against S0).
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>     // regmm_f32

namespace wkv7v5 {

using wkv7::BwdArgs;
using namespace wkv7c;       // N, L, mk8, split4, unpack4, ld8, st8, zero4, WKV_STAMP*

constexpr int IMG = L * N;           // elements of one [16][64] image
constexpr int HLI = L * 32;          // elements of one [16][16] pair-interleaved image ([a4 b4] per 16 bytes)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// element offset of (row, col) in a swizzled [16][64] bf16 image (8-element = 16-byte slots)
DEVFN int img_off(int row, int col) { return row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7)); }
// element offset of (row, col) in a swizzled [R][64] fp32 image (4-element = 16-byte slots, 16 per row)
DEVFN int f32_off(int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); }
// element offset of the 16-byte group (row, column group cg) of a pair-interleaved [16][16] image
DEVFN int hl_off(int row, int cg) { return row * 32 + ((cg ^ ((0 - (row >> 2)) & 3)) << 3); }
// channel held by row rho of state tile tb: tiles 2kb and 2kb+1 side by side are 8 consecutive channels per 4 rows
DEVFN int tix(int tb, int rho) { return 32 * (tb >> 1) + 8 * (rho >> 2) + 4 * (tb & 1) + (rho & 3); }

struct BufV5 {                       // produced per chunk, double buffered
    uint16_t opnd[8][IMG];           // Zt_h Zt_l Qt_h Qt_l Ah_h Ah_l Kh_h Kh_l      [t][j]
    uint16_t ab[4][IMG];             // Ab_h Ab_l Kb_h Kb_l                          [t][j]
    uint16_t ti[4][IMG];             // V  dY  SA_h  SA_l                            [t][i]
    uint16_t raw[4][IMG];            // q k z a as loaded (decay-gradient integrand) [t][j]
    uint16_t dz[2][IMG];             // "DZ" images of M_zk and T^T: columns 8g.. = [h h], columns 32+8g.. = [l 0]   [t][.]
    float dec[2][IMG];               // log2 c_t (inclusive), log2 w_t               [t][j] fp32
    float s0[N * N];                 // S0 = s[c-1] as stored ([j][i] fp32), 16-byte slots swizzled with j & 15
    uint16_t sc[2][HLI];             // M_qa, M_qk   image[t][s], [hi4 lo4] per 16 bytes
    float cl[N];                     // c_L[j]
    float glast[N];                  // sum_i dS_L[i][j] S_L[i][j] of this chunk (written one iteration early)
};
struct LdsV5 {
    BufV5 b[2];
    uint16_t dr[2][IMG];             // dR hi, lo  [t][i]  (consumers -> producers' dM and consumers' j-split)
    uint16_t dsc[4][HLI];            // score gradients image[t][s]: ZH = [za_h zk_h], ZL = [za_l zk_l], QH = [qa_h qk_h], QL
};
static_assert(sizeof(LdsV5) <= 160 * 1024, "LDS budget");

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
DEVFN bf16x8 ld16(const uint16_t* p) { VRWKV_LDS_TRACE(2, p) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(p)); }
DEVFN void st16(uint16_t* p, uint2 a, uint2 b) {
    u32x4v v = {a.x, a.y, b.x, b.y};
    VRWKV_LDS_TRACE(6, p)
    *reinterpret_cast<u32x4v*>(p) = v;
}
DEVFN f32x4 mfma32(bf16x8 x, bf16x8 y, f32x4 acc) { return mfma_16x16x32_bf16(x, y, acc); }

struct RawB { uint2 w, q, k, z, a, v, dy; float4 sa; };

// Per-lane LDS addressing (element offsets inside an image), computed once.
struct LaneAddr {
    int row[2];      // row read: 8 consecutive k of row c16, k block 0 / 1                       (ds_read_b128)
    int trc;         // transposing read, rows 4g.., the wave's own 16 columns                       (tr_b64)
    int tri[2];      // transposing read, rows 4g.., the columns of tile pair 0 / 1 (+4 for the odd tile of the pair)
    int own;         // 8-byte piece (row c16, columns 16 w + 4g..) -- producer stores, tail loads
    int hl;          // pair image: group (row c16, column group g)                                   (ds_read_b128)
    int hlt;         // pair image, transposing read: rows 4g.., column c16 (+4 for the second of the pair) (tr_b64)
    int f32;         // fp32 image: (row c16, columns 16 w + 4g..)                                    (ds_read_b128)
};
DEVFN LaneAddr lane_addr(int c16, int g, int w) {
    LaneAddr a;
    a.row[0] = img_off(c16, 8 * g);
    a.row[1] = img_off(c16, 32 + 8 * g);
    const int tr = 4 * g + (c16 >> 2), q = c16 & 3;
    a.trc = img_off(tr, 16 * w + 4 * q);
    a.tri[0] = img_off(tr, 8 * q);
    a.tri[1] = img_off(tr, 32 + 8 * q);
    a.own = img_off(c16, 16 * w + 4 * g);
    a.hl = hl_off(c16, g);
    a.hlt = hl_off(tr, q);
    a.f32 = f32_off(c16, 16 * w + 4 * g);
    return a;
}

// sum_k X(m,k) Y(n,k) over the 64 channels of two image rows (hi/lo pairs); result lane (c16 = n), registers m = 4g+r
template <bool XLO, bool YLO>
DEVFN f32x4 dot64(const uint16_t* Xh, const uint16_t* Xl, const uint16_t* Yh, const uint16_t* Yl, const LaneAddr& la) {
    f32x4 acc;
    {
        const bf16x8 xh = ld16(Xh + la.row[0]), yh = ld16(Yh + la.row[0]);
        acc = mfma32(xh, yh, zero4());
        if (YLO) acc = mfma32(xh, ld16(Yl + la.row[0]), acc);
        if (XLO) acc = mfma32(ld16(Xl + la.row[0]), yh, acc);
    }
    {
        const bf16x8 xh = ld16(Xh + la.row[1]), yh = ld16(Yh + la.row[1]);
        acc = mfma32(xh, yh, acc);
        if (YLO) acc = mfma32(xh, ld16(Yl + la.row[1]), acc);
        if (XLO) acc = mfma32(ld16(Xl + la.row[1]), yh, acc);
    }
    return acc;
}
// keep D[m = 4g+r][n = c16] where m (>|>=|<|<=) n, split
template <bool INCLUSIVE, bool UPPER>
DEVFN void mask_split(f32x4 d, int c16, int g, uint2& h, uint2& l) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        const bool keep = UPPER ? (INCLUSIVE ? m >= c16 : m > c16) : (INCLUSIVE ? m <= c16 : m < c16);
        d[r] = keep ? d[r] : 0.f;
    }
    split4(d, h, l);
}

// ------------------------------------------------------------------------------------------ producers
struct KeepB { float ah[4], kh[4], ab[4], kb[4]; };
DEVFN void prep_a(BufV5& B, const RawB& raw, int c16, int j0, const LaneAddr& la, KeepB& keep) {
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(raw.w, wr); unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
    float zt[4], qt[4], x2[4], l2[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp2(wr[e] * LOG2E) * LOG2E;          // log2 w_t   (w_t = exp(-exp(w_raw)), wkv7_cuda.cu:21)
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        // c_t = 2^x ; c_{t-1} is the previous lane's c_t (1 for the first token) ; c_L / c_t from the row's last lane
        const float cc = fast_exp2(x), ic = fast_exp2(-x);
        const float cp = dpp_shr1_fill(cc, 1.f), cb = dpp_row_last(cc) * ic;
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; keep.ah[e] = a[e] * ic; keep.kh[e] = k[e] * ic;
        keep.ab[e] = a[e] * cb; keep.kb[e] = k[e] * cb; cend[e] = cc; x2[e] = x; l2[e] = lw;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&B.opnd[0][la.own], hh); st8(&B.opnd[1][la.own], ll);
    split4(qt, hh, ll); st8(&B.opnd[2][la.own], hh); st8(&B.opnd[3][la.own], ll);
    *reinterpret_cast<float4*>(&B.dec[0][la.f32]) = make_float4(x2[0], x2[1], x2[2], x2[3]);
    *reinterpret_cast<float4*>(&B.dec[1][la.f32]) = make_float4(l2[0], l2[1], l2[2], l2[3]);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}
DEVFN void prep_b(BufV5& B, const RawB& raw, const LaneAddr& la, const KeepB& keep) {
    uint2 hh, ll;
    split4(keep.ah, hh, ll); st8(&B.opnd[4][la.own], hh); st8(&B.opnd[5][la.own], ll);
    split4(keep.kh, hh, ll); st8(&B.opnd[6][la.own], hh); st8(&B.opnd[7][la.own], ll);
    split4(keep.ab, hh, ll); st8(&B.ab[0][la.own], hh); st8(&B.ab[1][la.own], ll);
    split4(keep.kb, hh, ll); st8(&B.ab[2][la.own], hh); st8(&B.ab[3][la.own], ll);
    st8(&B.ti[0][la.own], raw.v);
    st8(&B.ti[1][la.own], raw.dy);
    const float sav[4] = {raw.sa.x, raw.sa.y, raw.sa.z, raw.sa.w};
    split4(sav, hh, ll); st8(&B.ti[2][la.own], hh); st8(&B.ti[3][la.own], ll);
    st8(&B.raw[0][la.own], raw.q); st8(&B.raw[1][la.own], raw.k);
    st8(&B.raw[2][la.own], raw.z); st8(&B.raw[3][la.own], raw.a);
}
// S0 image of a chunk = s[cidx] ([j][i] fp32, 16 KB) by LDS-DMA, four rows (1 KB) per instruction; this wave brings the
// row groups [k0, k1) of 16.  LDS position (row, slot') <- global (row, slot' ^ (row & 15)): the swizzle goes on the SOURCE
// address.  s_chunk == nullptr (chunk 0 of the sequence: S0 = 0): the image is zero-filled instead.
DEVFN void dma_state(float* img, const float* s_chunk, int k0, int k1, int lane) {
    if (s_chunk) {
        for (int k = k0; k < k1; ++k) {
            const int row = 4 * k + (lane >> 4);
            lds_dma16(s_chunk + row * N + (((lane & 15) ^ (row & 15)) << 2), img + 4 * k * N);
        }
    } else {
        for (int k = k0; k < k1; ++k) *reinterpret_cast<float4*>(img + 4 * k * N + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// U^T V on the f32 matrix core with two independent accumulation chains (dependent f32 MFMAs cost 40 cycles each)
DEVFN f32x4 regmm_f32x2(f32x4 u, f32x4 v) {
    f32x4 a = mfma_16x16x4_f32(u[0], v[0], zero4()), b = mfma_16x16x4_f32(u[1], v[1], zero4());
    a = mfma_16x16x4_f32(u[2], v[2], a);
    b = mfma_16x16x4_f32(u[3], v[3], b);
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    return a;
}
// U^T V for 16x16 register matrices in C layout (U as the X operand, V as the Y operand), split precision:
// X = [u_h | u_l], Y = [v_h ; v_h]  +  X = [u_h | 0], Y = [v_l ; 0]
DEVFN f32x4 regmm_x3(f32x4 u, f32x4 v) {
    uint2 uh, ul, vh, vl;
    split4(u, uh, ul);
    split4(v, vh, vl);
    const f32x4 acc = mfma32(mk8(uh, ul), mk8(vh, vh), zero4());
    return mfma32(mk8(uh.x, uh.y, 0u, 0u), mk8(vl.x, vl.y, 0u, 0u), acc);
}

template <bool DBL_BF16>
DEVFN void scores(BufV5& B, int pw, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (pw == 1) {            // image[t][s] = M_qa[s][t] = Qt_s . Ah_t , s >= t
        mask_split<true, true>(dot64<true, true>(B.opnd[2], B.opnd[3], B.opnd[4], B.opnd[5], la), c16, g, hh, ll);
        st16(B.sc[0] + la.hl, hh, ll);
    } else if (pw == 2) {     // M_qk[s][t] = Qt_s . Kh_t , s >= t
        mask_split<true, true>(dot64<true, true>(B.opnd[2], B.opnd[3], B.opnd[6], B.opnd[7], la), c16, g, hh, ll);
        st16(B.sc[1] + la.hl, hh, ll);
    } else if (pw == 3) {     // M_zk[s][t] = Zt_s . Kh_t , s > t      (DZ image)
        mask_split<false, true>(dot64<true, true>(B.opnd[0], B.opnd[1], B.opnd[6], B.opnd[7], la), c16, g, hh, ll);
        st16(B.dz[0] + la.row[0], hh, hh);
        st16(B.dz[0] + la.row[1], ll, make_uint2(0u, 0u));
    } else {                  // T = (I - M_za)^-1 by nilpotent doubling, register resident     (DZ image of T^T)
        // X[r] = M_za[4g+r][c16] and its transpose from the same eight row reads (operands swapped)
        f32x4 X, XT, Tc;
        {
            const bf16x8 zh0 = ld16(B.opnd[0] + la.row[0]), zl0 = ld16(B.opnd[1] + la.row[0]);
            const bf16x8 ah0 = ld16(B.opnd[4] + la.row[0]), al0 = ld16(B.opnd[5] + la.row[0]);
            const bf16x8 zh1 = ld16(B.opnd[0] + la.row[1]), zl1 = ld16(B.opnd[1] + la.row[1]);
            const bf16x8 ah1 = ld16(B.opnd[4] + la.row[1]), al1 = ld16(B.opnd[5] + la.row[1]);
            X = mfma32(zh0, ah0, zero4()); XT = mfma32(ah0, zh0, zero4());
            X = mfma32(zh0, al0, X);       XT = mfma32(al0, zh0, XT);
            X = mfma32(zl0, ah0, X);       XT = mfma32(ah0, zl0, XT);
            X = mfma32(zh1, ah1, X);       XT = mfma32(ah1, zh1, XT);
            X = mfma32(zh1, al1, X);       XT = mfma32(al1, zh1, XT);
            X = mfma32(zl1, ah1, X);       XT = mfma32(ah1, zl1, XT);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r] = (c16 < 4 * g + r) ? X[r] : 0.f;
            XT[r] = (4 * g + r < c16) ? XT[r] : 0.f;
            Tc[r] = X[r] + ((4 * g + r == c16) ? 1.f : 0.f);
        }
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const f32x4 XTn = DBL_BF16 ? regmm_x3(X, XT) : regmm_f32x2(X, XT);          // (X^T)^2
            f32x4 Xn = X;
            if (level < 2) Xn = DBL_BF16 ? regmm_x3(XT, X) : regmm_f32x2(XT, X);         // X^2
            const f32x4 D = DBL_BF16 ? regmm_x3(XTn, Tc) : regmm_f32x2(XTn, Tc);         // X_k T
#pragma unroll
            for (int r = 0; r < 4; ++r) Tc[r] += D[r];
            X = Xn; XT = XTn;
        }
        split4(Tc, hh, ll);                                      // Tc[r] = T[4g+r][c16] -> image[c16][4g+r]
        st16(B.dz[1] + la.row[0], hh, hh);
        st16(B.dz[1] + la.row[1], ll, make_uint2(0u, 0u));
    }
}
// score gradients image[t][s] = dM[t][s]: D[m = s][n = t] = X_s . Y_t with X in {SA, V}, Y in {dR, dY}; pair images
DEVFN void dscores(LdsV5& lds, const BufV5& B, int pw, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (pw == 0) mask_split<false, false>(dot64<true, true>(B.ti[2], B.ti[3], lds.dr[0], lds.dr[1], la), c16, g, hh, ll);          // dM_za = tril_(dR SA^T)
    else if (pw == 1) mask_split<false, false>(dot64<false, true>(B.ti[0], B.ti[0], lds.dr[0], lds.dr[1], la), c16, g, hh, ll);    // dM_zk = tril_(dR V^T)
    else if (pw == 2) mask_split<true, false>(dot64<true, false>(B.ti[2], B.ti[3], B.ti[1], B.ti[1], la), c16, g, hh, ll);        // dM_qa = tril(dY SA^T)
    else mask_split<true, false>(dot64<false, false>(B.ti[0], B.ti[0], B.ti[1], B.ti[1], la), c16, g, hh, ll);                    // dM_qk = tril(dY V^T)
    const int o = la.hl + 4 * (pw & 1);                      // za / qa first, zk / qk second of the pair
    st8(&lds.dsc[pw & 2][o], hh);
    st8(&lds.dsc[(pw & 2) + 1][o], ll);
}

// tiles (4 x f32x4 in C layout) -> hi/lo operands for the two k blocks (tiles 2kb, 2kb+1 side by side)
DEVFN void tiles_op(const f32x4* tl, bf16x8* oh, bf16x8* ol) {
    uint2 h[4], l[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) split4(tl[b], h[b], l[b]);
    oh[0] = mk8(h[0], h[1]); oh[1] = mk8(h[2], h[3]);
    ol[0] = mk8(l[0], l[1]); ol[1] = mk8(l[2], l[3]);
}

// ------------------------------------------------------------------------------------------ kernel
// MODE bit 1 (2): T doubling on the bf16 matrix core (split operands) instead of the f32 one.
// MODE bit 2 (4): producers at wave priority 2 instead of 1.
// MODE bit 7 (128): priorities by segment -- in segment 1 the producers drop to 0 and the consumers rise to 1 (the producers
// have ~1.2k cycles of slack per chunk there and the consumers none); everywhere else the producers stay above the consumers.
// TPAR: sequence-parallel launch: blockIdx.x = (b*H + h) * nseg + seg, chunks [c_lo, c_hi),
// dL/dS enters as ds_in[b,h,seg] and leaves as ds_out[b,h,seg] (both [i][j] fp32).  A launch that only asks for ds_out
// (ds_in == null: the first pass of the sequence-parallel backward, whose gradients are discarded) runs LITE: only what
// propagates dL/dS is computed -- no S0 images, no score gradients, no dV, no j-split output products, no tail, no stores.
template <bool PROF, int MODE = 0, bool TPAR = false>
__global__ __launch_bounds__(512) void bwd_kernel_v5(BwdArgs p) {
    LdsV5& lds = *reinterpret_cast<LdsV5*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const int nseg = TPAR ? p.nseg : 1;
    const unsigned bh = TPAR ? blockIdx.x / (unsigned)nseg : blockIdx.x;
    const int seg = TPAR ? (int)(blockIdx.x % (unsigned)nseg) : 0;
    const int c_lo = TPAR ? (int)((long)nchunk * seg / nseg) : 0, c_hi = TPAR ? (int)((long)nchunk * (seg + 1) / nseg) : nchunk;
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    const float* sbase = p.s + (size_t)bh * nchunk * N * N;
    const bool lite = TPAR && p.ds_out != nullptr && p.ds_in == nullptr;        // wave-uniform; constant false without TPAR
    WKV_STAMP_DECL
    const unsigned long long rt0_ = PROF ? realtime64_() : 0ull;       // constant-rate (100 MHz) counter: cycles / time = shader clock

    if (wave >= 4) {
        // ================================================================== producers (one chunk ahead; loads only)
        const int pw = wave - 4;
        // static priority for the producers (the younger half of the workgroup loses VALU arbitration otherwise): -7 % kernel time
        wave_priority<(MODE & 4) ? 2 : 1>();
        const LaneAddr la = lane_addr(c16, g, pw);
        const unsigned lane_off = (unsigned)c16 * ts + 16u * pw + 4u * g;
        auto fetch = [&](RawB& r, int c) {
            const size_t o = head_base + (size_t)c * L * ts + lane_off;
            r.w = *reinterpret_cast<const uint2*>(p.w + o); r.q = *reinterpret_cast<const uint2*>(p.q + o);
            r.k = *reinterpret_cast<const uint2*>(p.k + o); r.z = *reinterpret_cast<const uint2*>(p.z + o);
            r.a = *reinterpret_cast<const uint2*>(p.a + o); r.v = *reinterpret_cast<const uint2*>(p.v + o);
            r.dy = *reinterpret_cast<const uint2*>(p.dy + o); r.sa = *reinterpret_cast<const float4*>(p.sa + o);
        };
        RawB raw;
        fetch(raw, c_hi - 1);
        {   // prologue: the last chunk, completely
            KeepB keep;
            BufV5& B = lds.b[(c_hi - 1) & 1];
            if (!lite) dma_state(B.s0, c_hi - 1 > 0 ? sbase + (size_t)(c_hi - 2) * N * N : nullptr, 4 * pw, 4 * pw + 4, lane);   // S0 of chunk c is s[c-1]
            prep_a(B, raw, c16, 16 * pw + 4 * g, la, keep);
            prep_b(B, raw, la, keep);
            if (c_hi - 1 > c_lo) fetch(raw, c_hi - 2);
            block_sync_lds();
            if (!lite || pw < 2) scores<(MODE & 2) != 0>(B, pw, c16, g, la);       // lite: only T and M_qa are used
            if (!lite && pw > 0 && c_hi - 1 > c_lo)
                dma_state(lds.b[(c_hi - 2) & 1].s0, c_hi - 2 > 0 ? sbase + (size_t)(c_hi - 3) * N * N : nullptr, pw == 1 ? 0 : pw == 2 ? 5 : 10, pw == 1 ? 5 : pw == 2 ? 10 : 16, lane);
            block_sync_lds();
        }
        // iteration: consumers process chunk c, producers build chunk c-1 into the other buffer.  Global traffic of a
        // producer wave, in issue order: [S0 image of chunk c-2 by LDS-DMA, end of segment 3 (waves 1..3: wave 0 runs the
        // T chain)] [raw inputs of chunk c-2, segment 2]; only loads, so the counted waits below are exact.
        for (int c = c_hi - 1; c >= c_lo; --c) {
            const bool more = c > c_lo;
            KeepB keep;
            BufV5& Bn = lds.b[(c - 1) & 1];
            WKV_STAMP(0)
            if (MODE & 128) wave_priority<0>();         // segment 1: the producers have slack, the consumers do not
            if (more) prep_a(Bn, raw, c16, 16 * pw + 4 * g, la, keep);
            vmem_drain();                               // the S0 image of chunk c-1 (issued a segment ago) has landed
            WKV_STAMP(1)
            block_sync_lds();                           // X: dR(c) is in LDS
            if (MODE & 128) wave_priority<(MODE & 4) ? 2 : 1>();
            WKV_STAMP(2)
            if (more) {
                prep_b(Bn, raw, la, keep);
                if (c - 1 > c_lo) fetch(raw, c - 2);    // consumed in the next iteration's first segment
            }
            WKV_STAMP(3)
            if (!lite) dscores(lds, lds.b[c & 1], pw, c16, g, la);
            WKV_STAMP(4)
            block_sync_lds();                           // Y: dM(c) ready, all images of c-1 written
            WKV_STAMP(5)
            if (more && (!lite || pw < 2)) scores<(MODE & 2) != 0>(Bn, pw, c16, g, la);
            if (!lite && pw > 0 && c - 1 > c_lo)                 // S0 of chunk c-2 = s[c-3] into the buffer the consumers have just left
                dma_state(lds.b[c & 1].s0, c - 2 > 0 ? sbase + (size_t)(c - 3) * N * N : nullptr, pw == 1 ? 0 : pw == 2 ? 5 : 10, pw == 1 ? 5 : pw == 2 ? 10 : 16, lane);
            WKV_STAMP(6)
            block_sync_lds();                           // Z
            WKV_STAMP(7)
        }
        WKV_STAMP_FLUSH(256, 8, 8)
        return;
    }

    // ====================================================================== consumers (stores only inside the loop)
    const LaneAddr la = lane_addr(c16, g, wave);
    const int j = 16 * wave + c16;                      // key column of the j-split tiles
    f32x4 dS1[4], dS2[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { dS1[x] = zero4(); dS2[x] = zero4(); }
    float gl0 = 0.f;                                    // sum_i dS[i][j] S_L[i][j] of the first chunk processed
    if (TPAR && p.ds_in) {       // dS1[jb][r] = dS[16w+c16][tix(jb,4g+r)] ; dS2[ib][r] = dS[tix(ib,4g+r)][16w+c16]
        const float* di = p.ds_in + (size_t)blockIdx.x * N * N;
        const float* sp = sbase + (size_t)(c_hi - 1) * N * N + (size_t)j * N;       // S_L = s[c_hi-1]: [j][i]
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float4 t = *reinterpret_cast<const float4*>(di + (size_t)(16 * wave + c16) * N + tix(x, 4 * g));
            dS1[x][0] = t.x; dS1[x][1] = t.y; dS1[x][2] = t.z; dS1[x][3] = t.w;
            const float4 sl = *reinterpret_cast<const float4*>(sp + tix(x, 4 * g));
            const float slv[4] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dS2[x][r] = di[(size_t)tix(x, 4 * g + r) * N + j];
                gl0 = fmaf(dS2[x][r], slv[r], gl0);
            }
        }
        gl0 += lane_xor16(gl0);
        gl0 += lane_xor32(gl0);
    }
    if (g == 0) lds.b[(c_hi - 1) & 1].glast[j] = gl0;
    const unsigned out_off = (unsigned)c16 * ts + 16u * wave + 4u * g;      // token c16, channels 16w+4g..+3

    block_sync_lds(); block_sync_lds();                 // prologue barriers of the producers
    for (int c = c_hi - 1; c >= c_lo; --c) {
        const BufV5& B = lds.b[c & 1];
        const size_t cbase = head_base + (size_t)c * L * ts;
        WKV_STAMP(0)
        if (MODE & 128) wave_priority<1>();
        // ---------------------------------------------------------------- segment 1: i-split (i = 16w + c16)
        {
            bf16x8 sh[2], sl[2];
            tiles_op(dS1, sh, sl);
            const uint2 dyv = lds_read_tr16(&B.ti[1][la.trc]);               // dY[4g+e][i]
            const bf16x8 dyd = mk8(dyv, dyv);
            // dSA[t][i] = sum_s M_qa[s][t] dY[s][i] + sum_j Ab[t][j] dS[i][j]
            f32x4 dSA = mfma32(ld16(&B.sc[0][la.hl]), dyd, zero4());
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 ah = ld16(&B.ab[0][la.row[kb]]);
                dSA = mfma32(ah, sh[kb], dSA);
                dSA = mfma32(ah, sl[kb], dSA);
                dSA = mfma32(ld16(&B.ab[1][la.row[kb]]), sh[kb], dSA);
            }
            uint2 xh, xl, rh, rl;
            split4(dSA, xh, xl);
            const bf16x8 xhl = mk8(xh, xl);
            // dR = T^T dSA in both orientations: [t][i] stays in registers, [i][t] (token per lane) goes to LDS
            const bf16x8 t1 = ld16(&B.dz[1][la.row[0]]), t2 = ld16(&B.dz[1][la.row[1]]);        // [T_h T_h], [T_l 0]
            f32x4 dR = mfma32(t1, xhl, zero4());
            dR = mfma32(t2, xhl, dR);
            f32x4 dRT = mfma32(xhl, t1, zero4());
            dRT = mfma32(xhl, t2, dRT);
            split4(dR, rh, rl);
            {
                uint2 th, tl;
                split4(dRT, th, tl);
                st8(&lds.dr[0][la.own], th);
                st8(&lds.dr[1][la.own], tl);
            }
            // dV^T[i][t] = sum_j dS[i][j] Kb[t][j] + sum_s dY[s][i] M_qk[s][t] + sum_s dR[s][i] M_zk[s][t]
            if (!lite) {
            f32x4 dV = mfma32(dyd, ld16(&B.sc[1][la.hl]), zero4());
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 kh = ld16(&B.ab[2][la.row[kb]]);
                dV = mfma32(sh[kb], kh, dV);
                dV = mfma32(sl[kb], kh, dV);
                dV = mfma32(sh[kb], ld16(&B.ab[3][la.row[kb]]), dV);
            }
            {
                const bf16x8 rhl = mk8(rh, rl);
                dV = mfma32(rhl, ld16(&B.dz[0][la.row[0]]), dV);                 // [M_zk_h M_zk_h]
                dV = mfma32(rhl, ld16(&B.dz[0][la.row[1]]), dV);                 // [M_zk_l 0]
            }
            *reinterpret_cast<uint2*>(p.dv + cbase + out_off) = make_uint2(cvt_pk_bf16(dV[0], dV[1]), cvt_pk_bf16(dV[2], dV[3]));
            }
            // dS^T <- diag(c_L) dS^T + [Qt^T | Zt^T] [dY ; dR]
            const bf16x8 y1 = mk8(dyv, rh), y2 = mk8(0u, 0u, rl.x, rl.y);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
                f32x4 acc = dS1[jb];
                acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
                const int o = la.tri[jb >> 1] + 4 * (jb & 1);
                const bf16x8 xh8 = mk8(lds_read_tr16(&B.opnd[2][o]), lds_read_tr16(&B.opnd[0][o]));
                const bf16x8 xl8 = mk8(lds_read_tr16(&B.opnd[3][o]), lds_read_tr16(&B.opnd[1][o]));
                acc = mfma32(xh8, y1, acc);
                acc = mfma32(xl8, y1, acc);
                acc = mfma32(xh8, y2, acc);
                dS1[jb] = acc;
            }
        }
        WKV_STAMP(1)
        block_sync_lds();                                   // X
        if (MODE & 128) wave_priority<0>();
        WKV_STAMP(2)
        // ---------------------------------------------------------------- segment 2: j-split (j = 16w + c16)
        f32x4 dZt, dQt, dAh, dKh;
        bf16x8 qzh, qzl;
        {
            const float clj = B.cl[j];
            f32x4 S0[4], dU[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                // [ib][r] = S0[i = tix(ib, 4g+r)][j]  <-  image row j (zeros for the first chunk of the sequence)
                if (!lite) {
                    const float4 x = *reinterpret_cast<const float4*>(&B.s0[f32_off(j, tix(ib, 4 * g))]);
                    S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
                } else S0[ib] = zero4();
                dU[ib] = dS2[ib];
                dU[ib][0] *= clj; dU[ib][1] *= clj; dU[ib][2] *= clj; dU[ib][3] *= clj;
            }
            if (!lite) {
            bf16x8 s0h[2], s0l[2], duh[2], dul[2];
            tiles_op(S0, s0h, s0l);
            tiles_op(dU, duh, dul);
            // transposed results: D[m = j][n = t]  (lane = token, registers = 4 consecutive channels of the wave's 16)
            {
                const bf16x8 drh = ld16(&lds.dr[0][la.row[0]]);
                dZt = mfma32(s0h[0], drh, zero4());                                  // dR S0
                dZt = mfma32(s0l[0], drh, dZt);
                dZt = mfma32(s0h[0], ld16(&lds.dr[1][la.row[0]]), dZt);
                const bf16x8 dyr = ld16(&B.ti[1][la.row[0]]);
                dQt = mfma32(s0h[0], dyr, zero4());                                  // dY S0
                dQt = mfma32(s0l[0], dyr, dQt);
                const bf16x8 sah = ld16(&B.ti[2][la.row[0]]);
                dAh = mfma32(duh[0], sah, zero4());                                  // SA dU
                dAh = mfma32(dul[0], sah, dAh);
                dAh = mfma32(duh[0], ld16(&B.ti[3][la.row[0]]), dAh);
                const bf16x8 vr = ld16(&B.ti[0][la.row[0]]);
                dKh = mfma32(duh[0], vr, zero4());                                   // V dU
                dKh = mfma32(dul[0], vr, dKh);
            }
            {
                const bf16x8 drh = ld16(&lds.dr[0][la.row[1]]);
                dZt = mfma32(s0h[1], drh, dZt);
                dZt = mfma32(s0l[1], drh, dZt);
                dZt = mfma32(s0h[1], ld16(&lds.dr[1][la.row[1]]), dZt);
                const bf16x8 dyr = ld16(&B.ti[1][la.row[1]]);
                dQt = mfma32(s0h[1], dyr, dQt);
                dQt = mfma32(s0l[1], dyr, dQt);
                const bf16x8 sah = ld16(&B.ti[2][la.row[1]]);
                dAh = mfma32(duh[1], sah, dAh);
                dAh = mfma32(dul[1], sah, dAh);
                dAh = mfma32(duh[1], ld16(&B.ti[3][la.row[1]]), dAh);
                const bf16x8 vr = ld16(&B.ti[0][la.row[1]]);
                dKh = mfma32(duh[1], vr, dKh);
                dKh = mfma32(dul[1], vr, dKh);
            }
            }
            // dS <- dU + [dY^T | dR^T] [Qt ; Zt]
            qzh = mk8(lds_read_tr16(&B.opnd[2][la.trc]), lds_read_tr16(&B.opnd[0][la.trc]));
            qzl = mk8(lds_read_tr16(&B.opnd[3][la.trc]), lds_read_tr16(&B.opnd[1][la.trc]));
            const u32x4v qz = __builtin_bit_cast(u32x4v, qzh);
            const bf16x8 zpad = mk8(qz[2], qz[3], 0u, 0u);                           // [Zt_h ; 0]
            float gl = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const int o = la.tri[ib >> 1] + 4 * (ib & 1);
                const bf16x8 x8 = mk8(lds_read_tr16(&B.ti[1][o]), lds_read_tr16(&lds.dr[0][o]));
                // [dR_l^T | finite filler]: the filler meets the zero half of zpad (another tile's dR_l: finite, not reused)
                const bf16x8 xl8 = mk8(lds_read_tr16(&lds.dr[1][o]), lds_read_tr16(&lds.dr[1][o ^ 4]));
                f32x4 acc = dU[ib];
                acc = mfma32(x8, qzh, acc);
                acc = mfma32(x8, qzl, acc);
                acc = mfma32(xl8, zpad, acc);
                dS2[ib] = acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) gl = fmaf(acc[r], S0[ib][r], gl);       // S0 of this chunk = S_L of the next one
            }
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            if (g == 0) lds.b[(c - 1) & 1].glast[j] = gl;
        }
        WKV_STAMP(3)
        block_sync_lds();                                   // Y
        WKV_STAMP(4)
        // ---------------------------------------------------------------- segment 3: dM products + element-wise tail
        if (!lite) {
        {
            // dZt += dM_za Ah + dM_zk Kh ; dQt += dM_qa Ah + dM_qk Kh : X = [Ah^T | Kh^T], Y = pair image rows
            const bf16x8 akh = mk8(lds_read_tr16(&B.opnd[4][la.trc]), lds_read_tr16(&B.opnd[6][la.trc]));
            const bf16x8 akl = mk8(lds_read_tr16(&B.opnd[5][la.trc]), lds_read_tr16(&B.opnd[7][la.trc]));
            {
                const bf16x8 zh = ld16(&lds.dsc[0][la.hl]), qh = ld16(&lds.dsc[2][la.hl]);
                dZt = mfma32(akh, zh, dZt);
                dZt = mfma32(akl, zh, dZt);
                dZt = mfma32(akh, ld16(&lds.dsc[1][la.hl]), dZt);
                dQt = mfma32(akh, qh, dQt);
                dQt = mfma32(akl, qh, dQt);
                dQt = mfma32(akh, ld16(&lds.dsc[3][la.hl]), dQt);
            }
            // dAh += dM_za^T Zt + dM_qa^T Qt ; dKh += dM_zk^T Zt + dM_qk^T Qt : X = [Qt^T | Zt^T], Y = [qX^T ; zX^T]
            {
                const bf16x8 yh = mk8(lds_read_tr16(&lds.dsc[2][la.hlt]), lds_read_tr16(&lds.dsc[0][la.hlt]));
                dAh = mfma32(qzh, yh, dAh);
                dAh = mfma32(qzl, yh, dAh);
                dAh = mfma32(qzh, mk8(lds_read_tr16(&lds.dsc[3][la.hlt]), lds_read_tr16(&lds.dsc[1][la.hlt])), dAh);
            }
            {
                const bf16x8 yh = mk8(lds_read_tr16(&lds.dsc[2][la.hlt + 4]), lds_read_tr16(&lds.dsc[0][la.hlt + 4]));
                dKh = mfma32(qzh, yh, dKh);
                dKh = mfma32(qzl, yh, dKh);
                dKh = mfma32(qzh, mk8(lds_read_tr16(&lds.dsc[3][la.hlt + 4]), lds_read_tr16(&lds.dsc[1][la.hlt + 4])), dKh);
            }
        }
        {
            // lane: token t = c16, channels 16w + 4g + e
            const float4 x4 = *reinterpret_cast<const float4*>(&B.dec[0][la.f32]);
            const float4 l4 = *reinterpret_cast<const float4*>(&B.dec[1][la.f32]);
            const float4 gl4 = *reinterpret_cast<const float4*>(&B.glast[16 * wave + 4 * g]);
            const float x2[4] = {x4.x, x4.y, x4.z, x4.w}, l2[4] = {l4.x, l4.y, l4.z, l4.w}, glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
            float q[4], k[4], z[4], a[4];
            unpack4(ld8(&B.raw[0][la.own]), q); unpack4(ld8(&B.raw[1][la.own]), k);
            unpack4(ld8(&B.raw[2][la.own]), z); unpack4(ld8(&B.raw[3][la.own]), a);
            float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float cc = fast_exp2(x2[e]), ic = fast_exp2(-x2[e]), cp = dpp_shr1_fill(cc, 1.f);
                dz[e] = dZt[e] * cp; dq[e] = dQt[e] * cc; da[e] = dAh[e] * ic; dk[e] = dKh[e] * ic;
                // decay-gradient integrand g_t = dq q - da a - dk k + (dz z)[t+1]  (+ sum_i dS.S_L at the last token)
                float gt = dq[e] * q[e] - da[e] * a[e] - dk[e] * k[e] + dpp_shl<1>(dz[e] * z[e]);
                if (c16 == 15) gt += glv[e];
                gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
                dw[e] = gt * (l2[e] * LN2);
            }
            const size_t o = cbase + out_off;
            *reinterpret_cast<uint2*>(p.dw + o) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
            *reinterpret_cast<uint2*>(p.dq + o) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
            *reinterpret_cast<uint2*>(p.dk + o) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
            *reinterpret_cast<uint2*>(p.dz + o) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
            *reinterpret_cast<uint2*>(p.da + o) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
        }
        }
        WKV_STAMP(5)
        block_sync_lds();                                   // Z
        WKV_STAMP(6)
    }
    if (TPAR && p.ds_out) {
        float* dout = p.ds_out + (size_t)blockIdx.x * N * N;
#pragma unroll
        for (int x = 0; x < 4; ++x)
            *reinterpret_cast<float4*>(dout + (size_t)(16 * wave + c16) * N + tix(x, 4 * g)) = make_float4(dS1[x][0], dS1[x][1], dS1[x][2], dS1[x][3]);
    }
    WKV_STAMP_FLUSH(0, 0, 7)
    if (PROF && blockIdx.x == 0 && tid == 0) p.dbg[7] = realtime64_() - rt0_;   // ticks of the 100 MHz counter over workgroup 0's life
}

}  // namespace wkv7v5
