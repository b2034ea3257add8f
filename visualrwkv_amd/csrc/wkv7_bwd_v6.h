// Shared building blocks of the WKV7 backward kernels -- gfx950 (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130): the log2-domain decay
// scan, the register-level 16 x 16 products of the doubling chain, and the score pieces (M_qa, M_qk, M_zk, T = (I - M_za)^-1) that
// wkv7_bwd_v8.h (the kernel) and wkv7_bwd_rows.h (its memory role) use.  Until round 6 this header was also the home of the round-3 kernel
// (three-stage wave pipeline with 8-byte register loads, the launcher's choice for tensors of 4 GiB and more); the launcher now cuts such a
// launch into batch slices for wkv7_bwd_v8.h, and that kernel lives on as an A/B partner in benchmarks/experiments/wkv7_bwd_v6_kernel.h.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#ifndef VRWKV_PDELAY
#define VRWKV_PDELAY 1
#endif
#include <wkv7_bwd_v5.h>     // image addressing, dot64, mask_split, regmm_x3, tiles_op, dma_state

namespace wkv7v6 {

using wkv7::BwdArgs;
using namespace wkv7c;
using namespace wkv7v5;      // IMG, HLI, img_off, f32_off, hl_off, tix, LaneAddr, lane_addr, RawB, ld16, st16, mfma32, ...

struct TailRaw { uint2 q, k, z, a; float x2[4]; };      // inputs kept for the tail: q k z a + log2 c_t (the prep's scan, not recomputed)
template <bool V> struct BoolTag { static constexpr bool value = V; };

// decay scan of one lane's 4 channels over the 16 tokens of the chunk (lane = token c16 inside each 16-lane row)
struct Decay { float x2[4], l2[4]; };
DEVFN Decay decay_scan(uint2 wraw) {
    float wr[4];
    unpack4(wraw, wr);
    Decay d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp2(wr[e] * LOG2E) * LOG2E;          // log2 w_t   (w_t = exp(-exp(w_raw)), wkv7_cuda.cu:21)
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        d.x2[e] = x; d.l2[e] = lw;
    }
    return d;
}

// U^T V for 16x16 register matrices in C layout with the operands already split (see wkv7v5::regmm_x3): a level of the
// doubling uses every matrix twice, so each is split once (4 splits per level instead of 6: -72 VALU on the T chain's wave)
struct Split16 { uint2 h, l; };
DEVFN Split16 split16(f32x4 x) { Split16 s; split4(x, s.h, s.l); return s; }
// FULLX: the second MFMA re-uses the first one's X operand [u_h | u_l] against [v_l ; v_l] (the full product, u_l v_l included)
// instead of [u_h | 0] x [v_l ; 0]: one 4-register operand less to assemble per product (the T chain is ~30 % register moves)
template <bool FULLX = false>
DEVFN f32x4 regmm_pre(const Split16& u, const Split16& v) {
    const bf16x8 x = mk8(u.h, u.l);
    const f32x4 acc = mfma32(x, mk8(v.h, v.h), zero4());
    if (FULLX) return mfma32(x, mk8(v.l, v.l), acc);
    return mfma32(mk8(u.h.x, u.h.y, 0u, 0u), mk8(v.l.x, v.l.y, 0u, 0u), acc);
}

// ------------------------------------------------------------------------------------------ I: scores, T, score gradients
// piece 0: T (DZ image of T^T)   1: M_qa   2: M_qk   3: M_zk (DZ image)
template <bool DBL_BF16, class LdsT, class ImgT, bool FULLX = false>       // LdsT: .sc, .dz   ImgT: .opnd   (wkv7_bwd_v7.h / v8.h reuse this with their own layouts)
DEVFN void scores6(LdsT& lds, const ImgT& B, int piece, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (piece == 1) {            // image[t][s] = M_qa[s][t] = Qt_s . Ah_t , s >= t
        mask_split<true, true>(dot64<true, true>(B.opnd[2], B.opnd[3], B.opnd[4], B.opnd[5], la), c16, g, hh, ll);
        st16(lds.sc[0] + la.hl, hh, ll);
    } else if (piece == 2) {     // M_qk[s][t] = Qt_s . Kh_t , s >= t
        mask_split<true, true>(dot64<true, true>(B.opnd[2], B.opnd[3], B.opnd[6], B.opnd[7], la), c16, g, hh, ll);
        st16(lds.sc[1] + la.hl, hh, ll);
    } else if (piece == 3) {     // M_zk[s][t] = Zt_s . Kh_t , s > t      (DZ image)
        mask_split<false, true>(dot64<true, true>(B.opnd[0], B.opnd[1], B.opnd[6], B.opnd[7], la), c16, g, hh, ll);
        st16(lds.dz[0] + la.row[0], hh, hh);
        st16(lds.dz[0] + la.row[1], ll, make_uint2(0u, 0u));
    } else {                     // T = (I - M_za)^-1 by nilpotent doubling, register resident     (DZ image of T^T)
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
        if (DBL_BF16) {
            Split16 sx = split16(X), sxt = split16(XT);
#pragma unroll
            for (int level = 0; level < 3; ++level) {
                const f32x4 XTn = regmm_pre<FULLX>(sx, sxt);         // (X^T)^2
                f32x4 Xn = X;
                if (level < 2) Xn = regmm_pre<FULLX>(sxt, sx);       // X^2
                const Split16 sxtn = split16(XTn);
                const f32x4 D = regmm_pre<FULLX>(sxtn, split16(Tc)); // X_k T
#pragma unroll
                for (int r = 0; r < 4; ++r) Tc[r] += D[r];
                X = Xn; XT = XTn;
                sxt = sxtn;
                if (level < 2) sx = split16(Xn);
            }
        } else {
#pragma unroll
            for (int level = 0; level < 3; ++level) {
                const f32x4 XTn = regmm_f32x2(X, XT);                // exact f32 matrix core: 4 MFMAs of 32 cycles, no VALU
                f32x4 Xn = X;
                if (level < 2) Xn = regmm_f32x2(XT, X);
                const f32x4 D = regmm_f32x2(XTn, Tc);
#pragma unroll
                for (int r = 0; r < 4; ++r) Tc[r] += D[r];
                X = Xn; XT = XTn;
            }
        }
        split4(Tc, hh, ll);                                      // Tc[r] = T[4g+r][c16] -> image[c16][4g+r]
        st16(lds.dz[1] + la.row[0], hh, hh);
        st16(lds.dz[1] + la.row[1], ll, make_uint2(0u, 0u));
    }
}

}  // namespace wkv7v6
