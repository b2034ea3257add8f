// WKV7 forward, chunked MFMA form, second-generation schedule -- gfx950.
//
// Same math as wkv7_chunked.h (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-52) and the producer/consumer split of
// wkv7_fwd_v3.h (8 waves per (b,h), producers one chunk ahead, one workgroup barrier per chunk), rebuilt with the operand
// layouts of wkv7_bwd_v5.h after its phase stamps on MI355X: the consumers spent 4.8k of their 7.0k cycles per chunk in
// the two store phases -- 24 narrow global stores per wave and chunk (y: 4 x 2 bytes, sa: 4 x 4 bytes, state checkpoint:
// 16 x 4 bytes per lane), store-issue bound.
//
//  * y and sa are produced TRANSPOSED by swapping the MFMA operands (mfma(X, Y) = sum_k X(m,k) Y(n,k), both operands
//    "one row per lane"): lane = token, registers = 4 consecutive value channels -> one 8-byte and one 16-byte store.
//  * the checkpoint S^T[j][i] is needed with i contiguous, while the state lives as S^T tiles with j in the registers (the
//    products contract over j).  Each new tile is transposed on the matrix core: its hi/lo split -- needed anyway as next
//    chunk's operand -- times the identity, [S_h | S_l] [I ; I], exact for hi + lo (2^-17 relative to S): 4 wide stores
//    instead of 16 narrow ones.
//  * operand images as in the backward: swizzled [16][64] bf16 rows read with one ds_read_b128, transposing reads
//    (ds_read_b64_tr_b16) instead of transposed copies built with 2-byte stores, hi/lo pair and "DZ" images for the
//    16-deep products, state tiles with interleaved rows (tix) so that two tiles are 8 consecutive channels.
//  * every product is bf16x3 on the K=32 MFMA (SA = T R and M_qa SA included; the f32 MFMA is only used for the T chain).
#pragma once
#include <gfx950_prims.h>
#include <wkv7_bwd_v5.h>     // images, LaneAddr, dot64, mask_split, tiles_op, regmm_*

namespace wkv7v5 {

using wkv7::FwdArgs;

struct BufF5 {                       // produced per chunk, double buffered
    uint16_t opnd[8][IMG];           // Zt_h Zt_l Qt_h Qt_l Ah_h Ah_l Kh_h Kh_l      [t][j]
    uint16_t ab[4][IMG];             // Ab_h Ab_l Kb_h Kb_l                          [t][j]
    uint16_t v[IMG];                 // V                                            [t][i]
    uint16_t dz[2][IMG];             // "DZ" images ([h h] / [l 0]) of M_qa[t][s] and T[t][t']
    uint16_t sc[2][HLI];             // M_zk, M_qk   image[t][s], [hi4 lo4] per 16 bytes
    float cl[N];                     // c_L[j]
};
struct LdsF5 { BufF5 b[2]; unsigned prep_done; unsigned pad_[3]; };
static_assert(sizeof(LdsF5) <= 80 * 1024, "two workgroups per CU");

struct RawF { uint2 w, q, k, z, a, v; };

DEVFN void prep_f(BufF5& B, const RawF& raw, int c16, int j0, const LaneAddr& la) {
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(raw.w, wr); unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
    float zt[4], qt[4], ah[4], kh[4], ab[4], kb[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp2(wr[e] * LOG2E) * LOG2E;          // log2 w_t   (w_t = exp(-exp(w_raw)), wkv7_cuda.cu:21)
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        const float cc = fast_exp2(x), ic = fast_exp2(-x);
        const float cp = dpp_shr1_fill(cc, 1.f), cb = dpp_row_last(cc) * ic;
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; ah[e] = a[e] * ic; kh[e] = k[e] * ic;
        ab[e] = a[e] * cb; kb[e] = k[e] * cb; cend[e] = cc;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&B.opnd[0][la.own], hh); st8(&B.opnd[1][la.own], ll);
    split4(qt, hh, ll); st8(&B.opnd[2][la.own], hh); st8(&B.opnd[3][la.own], ll);
    split4(ah, hh, ll); st8(&B.opnd[4][la.own], hh); st8(&B.opnd[5][la.own], ll);
    split4(kh, hh, ll); st8(&B.opnd[6][la.own], hh); st8(&B.opnd[7][la.own], ll);
    split4(ab, hh, ll); st8(&B.ab[0][la.own], hh); st8(&B.ab[1][la.own], ll);
    split4(kb, hh, ll); st8(&B.ab[2][la.own], hh); st8(&B.ab[3][la.own], ll);
    st8(&B.v[la.own], raw.v);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}

// score images with rows t: D[m = s][n = t] = X_s . Y_t
template <bool DBL_BF16>
DEVFN void scores_f(BufF5& B, int pw, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (pw == 1) {            // M_zk[t][s] = Zt_t . Kh_s , s < t
        mask_split<false, false>(dot64<true, true>(B.opnd[6], B.opnd[7], B.opnd[0], B.opnd[1], la), c16, g, hh, ll);
        st16(B.sc[0] + la.hl, hh, ll);
    } else if (pw == 3) {     // M_qk[t][s] = Qt_t . Kh_s , s <= t
        mask_split<true, false>(dot64<true, true>(B.opnd[6], B.opnd[7], B.opnd[2], B.opnd[3], la), c16, g, hh, ll);
        st16(B.sc[1] + la.hl, hh, ll);
    } else if (pw == 2) {     // M_qa[t][s] = Qt_t . Ah_s , s <= t      (DZ image)
        mask_split<true, false>(dot64<true, true>(B.opnd[4], B.opnd[5], B.opnd[2], B.opnd[3], la), c16, g, hh, ll);
        st16(B.dz[0] + la.row[0], hh, hh);
        st16(B.dz[0] + la.row[1], ll, make_uint2(0u, 0u));
    } else {                  // T = (I - M_za)^-1, image rows t: the nilpotent doubling of wkv7_bwd_v5.h run on M_za^T
        f32x4 X, XT, Tc;      // X[r] = M_za^T[4g+r][c16], XT[r] = M_za[4g+r][c16]
        {
            const bf16x8 zh0 = ld16(B.opnd[0] + la.row[0]), zl0 = ld16(B.opnd[1] + la.row[0]);
            const bf16x8 ah0 = ld16(B.opnd[4] + la.row[0]), al0 = ld16(B.opnd[5] + la.row[0]);
            const bf16x8 zh1 = ld16(B.opnd[0] + la.row[1]), zl1 = ld16(B.opnd[1] + la.row[1]);
            const bf16x8 ah1 = ld16(B.opnd[4] + la.row[1]), al1 = ld16(B.opnd[5] + la.row[1]);
            XT = mfma32(zh0, ah0, zero4()); X = mfma32(ah0, zh0, zero4());
            XT = mfma32(zh0, al0, XT);      X = mfma32(al0, zh0, X);
            XT = mfma32(zl0, ah0, XT);      X = mfma32(ah0, zl0, X);
            XT = mfma32(zh1, ah1, XT);      X = mfma32(ah1, zh1, X);
            XT = mfma32(zh1, al1, XT);      X = mfma32(al1, zh1, X);
            XT = mfma32(zl1, ah1, XT);      X = mfma32(ah1, zl1, X);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            XT[r] = (c16 < 4 * g + r) ? XT[r] : 0.f;
            X[r] = (4 * g + r < c16) ? X[r] : 0.f;
            Tc[r] = X[r] + ((4 * g + r == c16) ? 1.f : 0.f);
        }
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const f32x4 XTn = DBL_BF16 ? regmm_x3(X, XT) : regmm_f32x2(X, XT);
            f32x4 Xn = X;
            if (level < 2) Xn = DBL_BF16 ? regmm_x3(XT, X) : regmm_f32x2(XT, X);
            const f32x4 D = DBL_BF16 ? regmm_x3(XTn, Tc) : regmm_f32x2(XTn, Tc);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tc[r] += D[r];
            X = Xn; XT = XTn;
        }
        split4(Tc, hh, ll);                                      // Tc[r] = T^T[4g+r][c16] -> image[c16][4g+r] = T[c16][4g+r]
        st16(B.dz[1] + la.row[0], hh, hh);
        st16(B.dz[1] + la.row[1], ll, make_uint2(0u, 0u));
    }
}

template <bool PROF, int MODE = 0>
__global__ __launch_bounds__(512, 2) void fwd_kernel_v5(FwdArgs p) {
    LdsF5& lds = *reinterpret_cast<LdsF5*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const size_t head_base = ((size_t)(blockIdx.x / H) * T * H + (blockIdx.x % H)) * N;
    WKV_STAMP_DECL

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (loads only)
        const int pw = wave - 4;
        wave_priority<1>();                              // the younger half would otherwise lose VALU arbitration
        const LaneAddr la = lane_addr(c16, g, pw);
        const unsigned lane_off = (unsigned)c16 * ts + 16u * pw + 4u * g;
        auto fetch = [&](RawF& r, int c) {
            const size_t o = head_base + (size_t)c * L * ts + lane_off;
            r.w = *reinterpret_cast<const uint2*>(p.w + o); r.q = *reinterpret_cast<const uint2*>(p.q + o);
            r.k = *reinterpret_cast<const uint2*>(p.k + o); r.z = *reinterpret_cast<const uint2*>(p.z + o);
            r.a = *reinterpret_cast<const uint2*>(p.a + o); r.v = *reinterpret_cast<const uint2*>(p.v + o);
        };
        RawF rc;
        fetch(rc, 0);
        block_sync_lds();                               // prep_done is zeroed
        for (int c = 0; c <= nchunk; ++c) {            // iteration c produces chunk c (one ahead of the consumers)
            if (c < nchunk) {
                const RawF cur = rc;
                if (c + 1 < nchunk) fetch(rc, c + 1);
                prep_f(lds.b[c & 1], cur, c16, 16 * pw + 4 * g, la);
                lds_flag_add(&lds.prep_done);
            }
            WKV_STAMP(0)
            if (c < nchunk) lds_flag_wait(&lds.prep_done, 4u * (unsigned)(c + 1));     // all four producers' images
            WKV_STAMP(1)
            if (c < nchunk) scores_f<(MODE & 2) != 0>(lds.b[c & 1], pw, c16, g, la);
            WKV_STAMP(2)
            block_sync_lds();                           // B
            WKV_STAMP(3)
        }
        WKV_STAMP_FLUSH(256, 8, 4)
        return;
    }

    // ---------------------------------------------------------------------- consumers (stores only)
    const LaneAddr la = lane_addr(c16, g, wave);
    f32x4 ST[4];                  // S^T tiles: [jb][r] = S[i = 16w + c16][j = tix(jb, 4g + r)]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) ST[jb] = zero4();
    if (p.s0) {
        const float* sp = p.s0 + ((size_t)blockIdx.x * N + 16 * wave + c16) * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 x = *reinterpret_cast<const float4*>(sp + tix(jb, 4 * g));
            ST[jb][0] = x.x; ST[jb][1] = x.y; ST[jb][2] = x.z; ST[jb][3] = x.w;
        }
    }
    const unsigned out_off = (unsigned)c16 * ts + 16u * wave + 4u * g;      // token c16, channels 16w + 4g ..+3
    float* psa = p.sa ? p.sa + head_base : nullptr;
    uint16_t* py = p.y + head_base;
    float* ps = p.s ? p.s + (size_t)blockIdx.x * nchunk * N * N : nullptr;
    uint2 idp;                    // identity as a K=16 operand: element e of lane (c16, g) = (c16 == 4g + e)
    idp.x = (c16 == 4 * g ? 0x3F80u : 0u) | (c16 == 4 * g + 1 ? 0x3F800000u : 0u);
    idp.y = (c16 == 4 * g + 2 ? 0x3F80u : 0u) | (c16 == 4 * g + 3 ? 0x3F800000u : 0u);
    const bf16x8 ident = mk8(idp, idp);
    uint2 th[4], tl[4];           // hi / lo split of the state tiles (operands of this chunk, transposed copies of the last)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) split4(ST[jb], th[jb], tl[jb]);

    if (tid == 0) lds.prep_done = 0u;
    block_sync_lds();      // prep_done is zeroed
    block_sync_lds();      // B  (producers have filled buffer 0)
    for (int c = 0; c < nchunk; ++c) {
        const BufF5& B = lds.b[c & 1];
        WKV_STAMP(0)
        const bf16x8 sh[2] = {mk8(th[0], th[1]), mk8(th[2], th[3])};
        const bf16x8 sl[2] = {mk8(tl[0], tl[1]), mk8(tl[2], tl[3])};
        const uint2 vv = lds_read_tr16(&B.v[la.trc]);                          // V[4g+e][i]
        const bf16x8 vvd = mk8(vv, vv);
        // R[t][i] = sum_s M_zk[t][s] V[s][i] + sum_j Zt[t][j] S0[i][j]
        f32x4 R = mfma32(ld16(&B.sc[0][la.hl]), vvd, zero4());
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 zh = ld16(&B.opnd[0][la.row[kb]]);
            R = mfma32(zh, sh[kb], R);
            R = mfma32(zh, sl[kb], R);
            R = mfma32(ld16(&B.opnd[1][la.row[kb]]), sh[kb], R);
        }
        uint2 rh, rl, sah, sal;
        split4(R, rh, rl);
        const bf16x8 rhl = mk8(rh, rl);
        // SA = T R in [t][i] (operand of what follows) and, for the store, [i][t]
        const bf16x8 t1 = ld16(&B.dz[1][la.row[0]]), t2 = ld16(&B.dz[1][la.row[1]]);          // [T_h T_h], [T_l 0]
        f32x4 SA = mfma32(t1, rhl, zero4());
        SA = mfma32(t2, rhl, SA);
        if (psa) {
            f32x4 SAT = mfma32(rhl, t1, zero4());
            SAT = mfma32(rhl, t2, SAT);
            *reinterpret_cast<float4*>(psa + (size_t)c * L * ts + out_off) = make_float4(SAT[0], SAT[1], SAT[2], SAT[3]);
        }
        split4(SA, sah, sal);
        WKV_STAMP(1)
        // Y^T[i][t] = sum_s V[s][i] M_qk[t][s] + sum_j S0[i][j] Qt[t][j] + sum_s SA[s][i] M_qa[t][s]
        {
            f32x4 YT = mfma32(vvd, ld16(&B.sc[1][la.hl]), zero4());
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 qh = ld16(&B.opnd[2][la.row[kb]]);
                YT = mfma32(sh[kb], qh, YT);
                YT = mfma32(sl[kb], qh, YT);
                YT = mfma32(sh[kb], ld16(&B.opnd[3][la.row[kb]]), YT);
            }
            const bf16x8 sahl = mk8(sah, sal);
            YT = mfma32(sahl, ld16(&B.dz[0][la.row[0]]), YT);                      // [M_qa_h M_qa_h]
            YT = mfma32(sahl, ld16(&B.dz[0][la.row[1]]), YT);                      // [M_qa_l 0]
            *reinterpret_cast<uint2*>(py + (size_t)c * L * ts + out_off) = make_uint2(cvt_pk_bf16(YT[0], YT[1]), cvt_pk_bf16(YT[2], YT[3]));
        }
        WKV_STAMP(2)
        // S_L^T = diag(c_L) S0^T + [Ab^T | Kb^T] [SA ; V]
        const bf16x8 y1 = mk8(sah, vv), y2 = mk8(sal.x, sal.y, 0u, 0u);
        float* s_c = ps + (size_t)c * N * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
            f32x4 acc = ST[jb];
            acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
            const int o = la.tri[jb >> 1] + 4 * (jb & 1);
            const bf16x8 xh8 = mk8(lds_read_tr16(&B.ab[0][o]), lds_read_tr16(&B.ab[2][o]));
            const bf16x8 xl8 = mk8(lds_read_tr16(&B.ab[1][o]), lds_read_tr16(&B.ab[3][o]));
            acc = mfma32(xh8, y1, acc);
            acc = mfma32(xl8, y1, acc);
            acc = mfma32(xh8, y2, acc);
            ST[jb] = acc;
            split4(acc, th[jb], tl[jb]);                 // next chunk's operand, and the transposed checkpoint:
            if (ps) {                                    // [S_h | S_l] [I ; I]: lane = key row, registers = 4 value columns
                const f32x4 tt = mfma32(mk8(th[jb], tl[jb]), ident, zero4());
                *reinterpret_cast<float4*>(s_c + (size_t)tix(jb, c16) * N + 16 * wave + 4 * g) = make_float4(tt[0], tt[1], tt[2], tt[3]);
            }
        }
        WKV_STAMP(3)
        WKV_STAMP(4)
        block_sync_lds();                                        // B
        WKV_STAMP(5)
    }
    if (p.s_final) {
        float* sp = p.s_final + ((size_t)blockIdx.x * N + 16 * wave + c16) * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) *reinterpret_cast<float4*>(sp + tix(jb, 4 * g)) = make_float4(ST[jb][0], ST[jb][1], ST[jb][2], ST[jb][3]);
    }
    WKV_STAMP_FLUSH(0, 0, 6)
}

}  // namespace wkv7v5
