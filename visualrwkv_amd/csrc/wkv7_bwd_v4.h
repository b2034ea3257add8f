// WKV7 backward, 12 waves per (b,h): the consumer work of wkv7_bwd_v3.h is split by ROLE so that the two halves of a
// chunk's backward that do not depend on each other run side by side (3 waves per SIMD instead of 2):
//
//                 before X                         X .. Y                              Y .. Z
//   I waves 4-7   i-split(c): dSA dR dV, dS^T       -                                   -
//   J waves 0-3   dY S0, SA dU, V dU, glast         dR S0, dS update                    dM products + tail(c)
//   P waves 8-11  prep_a(c-1)                       prep_b(c-1), score gradients dM(c)  scores / T of chunk c-1
//
// X: dR(c) is in LDS; Y: the dM(c) images are in LDS (and nobody reads dr/drT any more: the tail strips alias them);
// Z: chunk done.  Measured per-wave work in the 8-wave kernel: i-split 2.4 k cycles, the dR-independent part of the
// j-split 1.4 k, the rest 1.2 k, dM products + tail 3.0 k; producers 2.4 / 1.8 / 2.4 k.  There the consumers run
// i-split, j-split and tail back to back (8.1 k per chunk); here the critical path is max(2.4, 1.4, 2.4) +
// max(1.2, 1.8) + max(3.0, 2.4) = 7.2 k.  Same LDS image as wkv7_bwd_v3.h (buffers, strips, helper functions).
//
// MEASURED (MI355X, B = 8/16/32): 0.762 / 1.521 / 3.022 ms against 0.681 / 1.332 / 2.651 ms for the 8-wave kernel.
// With three waves per SIMD every role's segment takes 1.5-2x longer (prep_a 4.9 k cycles instead of 2.4 k): the
// SIMDs were already 65-70 % issue-busy with two waves (PMC: VALU 44 %, MFMA 19 %, LDS/SALU the rest), so the
// backward is bound by instruction issue -- the cost of the split-precision (bf16x3) arithmetic -- not by latency, and
// a third wave only adds contention (plus 5 spilled VGPRs at the 168-register limit).  Kept as variant 6 for A/B.
#pragma once
#include <wkv7_bwd_v3.h>

namespace wkv7c {

template <bool PROF>
__global__ __launch_bounds__(768) void bwd_kernel_v4(BwdArgs p) {
    LdsB3& lds = *reinterpret_cast<LdsB3*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = uniform_i32(tid >> 6);
    const int role = wave_all >> 2;                  // 0: J consumers, 1: I consumers, 2: producers
    const int wave = wave_all & 3;
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const size_t head_base = ((size_t)(blockIdx.x / H) * T * H + (blockIdx.x % H)) * N;
    WKV_STAMP_DECL

    if (role == 2) {
        // ================================================================== producers (as in wkv7_bwd_v3.h)
        const int pw = wave;
        const unsigned lane_off = (unsigned)c16 * ts + 16u * pw + 4u * g;
        auto fetch = [&](RawB& r, int c) {
            const size_t o = head_base + (size_t)c * L * ts + lane_off;
            r.w = *reinterpret_cast<const uint2*>(p.w + o); r.q = *reinterpret_cast<const uint2*>(p.q + o);
            r.k = *reinterpret_cast<const uint2*>(p.k + o); r.z = *reinterpret_cast<const uint2*>(p.z + o);
            r.a = *reinterpret_cast<const uint2*>(p.a + o); r.v = *reinterpret_cast<const uint2*>(p.v + o);
            r.dy = *reinterpret_cast<const uint2*>(p.dy + o); r.sa = *reinterpret_cast<const float4*>(p.sa + o);
        };
        RawB raw;
        fetch(raw, nchunk - 1);
        {
            KeepB keep;
            RawB cur = raw;
            if (nchunk > 1) fetch(raw, nchunk - 2);
            bwd_prep_a(lds, lds.b[(nchunk - 1) & 1], cur, pw, lane, keep);
            bwd_prep_b(lds, lds.b[(nchunk - 1) & 1], cur, pw, lane, keep);
            block_sync_lds();   // X
            block_sync_lds();   // Y
            bwd_scores<true>(lds, lds.b[(nchunk - 1) & 1], pw, lane);
            block_sync_lds();   // Z
        }
        for (int c = nchunk - 1; c >= 0; --c) {
            const bool more = c > 0;
            KeepB keep;
            RawB cur = raw;
            if (more) {
                if (c > 1) fetch(raw, c - 2);
                bwd_prep_a(lds, lds.b[(c - 1) & 1], cur, pw, lane, keep);
            }
            WKV_STAMP(0)
            block_sync_lds();   // X
            WKV_STAMP(1)
            bwd_dscores(lds, lds.b[c & 1], pw, lane);          // first: the J waves' last segment waits for these
            if (more) bwd_prep_b(lds, lds.b[(c - 1) & 1], cur, pw, lane, keep);
            WKV_STAMP(2)
            block_sync_lds();   // Y
            WKV_STAMP(3)
            if (more) bwd_scores<true>(lds, lds.b[(c - 1) & 1], pw, lane);
            WKV_STAMP(4)
            block_sync_lds();   // Z
            WKV_STAMP(5)
        }
        WKV_STAMP_FLUSH(512, 8, 6)
        return;
    }

    if (role == 1) {
        // ================================================================== I consumers: value columns 16w + c16
        f32x4 dS1[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) dS1[x] = zero4();
        const unsigned dv_off = (unsigned)(4 * g) * ts + 16u * wave + c16;
        block_sync_lds(); block_sync_lds(); block_sync_lds();      // prologue X, Y, Z
        for (int c = nchunk - 1; c >= 0; --c) {
            const BufB& B = lds.b[c & 1];
            const size_t cbase = head_base + (size_t)c * L * ts;
            uint2 rh, rl;
            const uint2 dy = ld8(&B.dyT[16 * wave + c16][4 * g]);
            bf16x8 bh[2], bl[2];
            tiles_to_b(dS1, 1.f, bh, bl);
            f32x4 dSA = mm_small_exact(zero4(), B.sc[0][0], B.sc[0][1], c16, g, dy);                // M_qa^T dY
            dSA = mm_perm<true>(dSA, B.ab[0], B.ab[1], c16, g, bh, bl);                             // Ab dS^T
            uint2 xh, xl;
            split4(dSA, xh, xl);
            const f32x4 dR = mm_small2(zero4(), B.sc[3][0], B.sc[3][1], c16, g, make_bpair(xh, xl)); // T^T dSA
            split4(dR, rh, rl);
            st_b16x4_col(lds.dr[0], 4 * g, 16 * wave + c16, rh);
            st_b16x4_col(lds.dr[1], 4 * g, 16 * wave + c16, rl);
            st8(&lds.drT[0][16 * wave + c16][4 * g], rh);
            st8(&lds.drT[1][16 * wave + c16][4 * g], rl);
            block_sync_lds();   // X : dR(c) is in LDS -- everything below overlaps the other roles' next segments
            f32x4 dV = mm_small_exact(zero4(), B.sc[1][0], B.sc[1][1], c16, g, dy);                 // M_qk^T dY
            dV = mm_perm<true>(dV, B.ab[2], B.ab[3], c16, g, bh, bl);                               // Kb dS^T
            dV = mm_small2(dV, B.sc[2][0], B.sc[2][1], c16, g, make_bpair(rh, rl));                 // M_zk^T dR
            uint16_t* dvp = p.dv + cbase;
            const uint32_t v01 = cvt_pk_bf16(dV[0], dV[1]), v23 = cvt_pk_bf16(dV[2], dV[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) dvp[dv_off + r * ts] = (uint16_t)((r < 2 ? v01 : v23) >> (16 * (r & 1)));
            // dS^T <- diag(c_L) dS^T + [Qt^T | Zt^T] [dY ; dR]
            const bf16x8 b1 = mk8(dy, rh), b2 = mk8(0u, 0u, rl.x, rl.y);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float4 cl = *reinterpret_cast<const float4*>(&B.cl[16 * jb + 4 * g]);
                f32x4 acc = dS1[jb];
                acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
                const int j = 16 * jb + c16;
                const bf16x8 ah = mk8(ld8(&B.trn[2][j][4 * g]), ld8(&B.trn[0][j][4 * g]));
                const bf16x8 al = mk8(ld8(&B.trn[3][j][4 * g]), ld8(&B.trn[1][j][4 * g]));
                acc = mfma_16x16x32_bf16(ah, b1, acc);
                acc = mfma_16x16x32_bf16(ah, b2, acc);
                acc = mfma_16x16x32_bf16(al, b1, acc);
                dS1[jb] = acc;
            }
            block_sync_lds();   // Y
            block_sync_lds();   // Z
        }
        return;
    }

    // ====================================================================== J consumers: key column j = 16w + c16
    const float* sbase = p.s + (size_t)blockIdx.x * nchunk * N * N;
    // Register diet for 3 waves per SIMD (168 VGPRs): the chunk-start state of the NEXT iteration is fetched during the
    // tail (low pressure) and no copy of the previous S0 is kept -- sum_i dS[i][j] S_L[i][j] of the next chunk is
    // formed right after this chunk's dS update, while S0 (= the next chunk's S_L) is still live.
    f32x4 dS2[4], S0n[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) dS2[x] = zero4();
    auto load_state = [&](f32x4* dst, int cidx) {      // s[cidx] as S[i][j] tiles: [ib][r] = S[16ib+4g+r][16w+c16]
        const float* sp = sbase + (size_t)cidx * N * N + (size_t)(16 * wave + c16) * N + 4 * g;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float4 x = *reinterpret_cast<const float4*>(sp + 16 * ib);
            dst[ib][0] = x.x; dst[ib][1] = x.y; dst[ib][2] = x.z; dst[ib][3] = x.w;
        }
    };
    if (nchunk > 1) load_state(S0n, nchunk - 2);
    else {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) S0n[ib] = zero4();
    }
    float gl_next = 0.f;                               // sum_i dS[i][j] S_L[i][j] for the chunk about to be processed
    const int c0 = 16 * wave + 4 * g;
    const unsigned row_off = (unsigned)c16 * ts + (unsigned)c0;
    const int j = 16 * wave + c16;

    block_sync_lds(); block_sync_lds(); block_sync_lds();      // prologue X, Y, Z
    for (int c = nchunk - 1; c >= 0; --c) {
        const BufB& B = lds.b[c & 1];
        const size_t cbase = head_base + (size_t)c * L * ts;
        const float clj = B.cl[j];
        f32x4 S0[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) S0[ib] = S0n[ib];
        // ---------------------------------------------------------------- part A: everything that does not need dR(c)
        bf16x8 s0h[2], s0l[2], duh[2], dul[2];
        tiles_to_b(S0, 1.f, s0h, s0l);
        tiles_to_b(dS2, clj, duh, dul);
        f32x4 dQt = mm_perm<false>(zero4(), B.ti[1], B.ti[1], c16, g, s0h, s0l);                   // dY S0
        f32x4 dAh = mm_perm<true>(zero4(), B.ti[2], B.ti[3], c16, g, duh, dul);                    // SA dU
        f32x4 dKh = mm_perm<false>(zero4(), B.ti[0], B.ti[0], c16, g, duh, dul);                   // V dU
        if (g == 0) lds.glast[j] = gl_next;
        WKV_STAMP(0)
        block_sync_lds();       // X
        WKV_STAMP(1)
        // ---------------------------------------------------------------- part B: products with dR(c)
        f32x4 dZt = mm_perm<true>(zero4(), lds.dr[0], lds.dr[1], c16, g, s0h, s0l);                // dR S0
        const uint2 zth = ld8(&B.trn[0][j][4 * g]), ztl = ld8(&B.trn[1][j][4 * g]);
        const uint2 qth = ld8(&B.trn[2][j][4 * g]), qtl = ld8(&B.trn[3][j][4 * g]);
        {
            // dS <- dS diag(c_L) + [dY^T | dR^T] [Qt ; Zt]
            const bf16x8 bqh = mk8(qth, zth), bql = mk8(qtl, ztl);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                f32x4 acc = dS2[ib];
                acc[0] *= clj; acc[1] *= clj; acc[2] *= clj; acc[3] *= clj;
                const int i = 16 * ib + c16;
                const bf16x8 ah = mk8(ld8(&B.dyT[i][4 * g]), ld8(&lds.drT[0][i][4 * g]));
                const uint2 rl2 = ld8(&lds.drT[1][i][4 * g]);
                acc = mfma_16x16x32_bf16(ah, bqh, acc);
                acc = mfma_16x16x32_bf16(ah, bql, acc);
                acc = mfma_16x16x32_bf16(mk8(0u, 0u, rl2.x, rl2.y), bqh, acc);
                dS2[ib] = acc;
            }
            float gl = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) gl = fmaf(dS2[ib][r], S0[ib][r], gl);
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            gl_next = gl;
        }
        WKV_STAMP(2)
        block_sync_lds();       // Y
        WKV_STAMP(3)
        // ---------------------------------------------------------------- part C: dM products + tail
        const uint2 tw = *reinterpret_cast<const uint2*>(p.w + cbase + row_off);      // raw decay for the tail (read twice)
        if (c > 1) load_state(S0n, c - 2);                 // next iteration's chunk-start state
        else {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) S0n[ib] = zero4();
        }
        {
            const uint2 ahh = ld8(&B.trn[4][j][4 * g]), ahl = ld8(&B.trn[5][j][4 * g]);
            const uint2 khh = ld8(&B.trn[6][j][4 * g]), khl = ld8(&B.trn[7][j][4 * g]);
            const BPair bpa = make_bpair(ahh, ahl), bpk = make_bpair(khh, khl), bpz = make_bpair(zth, ztl), bpq = make_bpair(qth, qtl);
            dZt = mm_small2(dZt, lds.dsc[0][0], lds.dsc[0][1], c16, g, bpa);                         // dM_za Ah
            dZt = mm_small2(dZt, lds.dsc[2][0], lds.dsc[2][1], c16, g, bpk);                         // dM_zk Kh
            dQt = mm_small2(dQt, lds.dsc[4][0], lds.dsc[4][1], c16, g, bpa);                         // dM_qa Ah
            dQt = mm_small2(dQt, lds.dsc[6][0], lds.dsc[6][1], c16, g, bpk);                         // dM_qk Kh
            dAh = mm_small2(dAh, lds.dsc[1][0], lds.dsc[1][1], c16, g, bpz);                         // dM_za^T Zt
            dAh = mm_small2(dAh, lds.dsc[5][0], lds.dsc[5][1], c16, g, bpq);                         // dM_qa^T Qt
            dKh = mm_small2(dKh, lds.dsc[3][0], lds.dsc[3][1], c16, g, bpz);                         // dM_zk^T Zt
            dKh = mm_small2(dKh, lds.dsc[7][0], lds.dsc[7][1], c16, g, bpq);                         // dM_qk^T Qt
            float pz[4], gq[4];
            {
                float hh[4], ll[4];
                unpack4(zth, hh); unpack4(ztl, ll);
#pragma unroll
                for (int r = 0; r < 4; ++r) pz[r] = dZt[r] * (hh[r] + ll[r]);
                unpack4(qth, hh); unpack4(qtl, ll);
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] = dQt[r] * (hh[r] + ll[r]);
                unpack4(ahh, hh); unpack4(ahl, ll);
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] -= dAh[r] * (hh[r] + ll[r]);
                unpack4(khh, hh); unpack4(khl, ll);
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[r] -= dKh[r] * (hh[r] + ll[r]);
            }
            float pzn = lane_bcast(pz[0], (lane + 16) & 63);
            pzn = g == 3 ? 0.f : pzn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float nx = r < 3 ? pz[r < 3 ? r + 1 : 3] : pzn;
                res_mat(lds, 4)[(4 * g + r) * RS + j] = gq[r] + nx;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            res_mat(lds, 0)[(4 * g + r) * RS + j] = dZt[r];
            res_mat(lds, 1)[(4 * g + r) * RS + j] = dQt[r];
            res_mat(lds, 2)[(4 * g + r) * RS + j] = dAh[r];
            res_mat(lds, 3)[(4 * g + r) * RS + j] = dKh[r];
        }
        wave_lds_fence();
        {
            const float4 rz = *reinterpret_cast<const float4*>(res_mat(lds, 0) + c16 * RS + c0);
            const float4 rq = *reinterpret_cast<const float4*>(res_mat(lds, 1) + c16 * RS + c0);
            const float4 ra = *reinterpret_cast<const float4*>(res_mat(lds, 2) + c16 * RS + c0);
            const float4 rk = *reinterpret_cast<const float4*>(res_mat(lds, 3) + c16 * RS + c0);
            const float4 rg = *reinterpret_cast<const float4*>(res_mat(lds, 4) + c16 * RS + c0);
            const float4 gl4 = *reinterpret_cast<const float4*>(&lds.glast[c0]);
            const float dzt[4] = {rz.x, rz.y, rz.z, rz.w}, dqt[4] = {rq.x, rq.y, rq.z, rq.w};
            const float dah[4] = {ra.x, ra.y, ra.z, ra.w}, dkh[4] = {rk.x, rk.y, rk.z, rk.w};
            const float glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w}, gin[4] = {rg.x, rg.y, rg.z, rg.w};
            float wr[4];
            unpack4(tw, wr);
            float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lw = -fast_exp(wr[e]);
                float x = lw;
                x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
                const float cc = fast_exp(x), cp = fast_exp(x - lw), ic = fast_exp(-x);
                dz[e] = dzt[e] * cp; dq[e] = dqt[e] * cc; da[e] = dah[e] * ic; dk[e] = dkh[e] * ic;
                float gt = gin[e];
                if (c16 == 15) gt += glv[e];
                gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);
                dw[e] = gt * lw;
            }
            const size_t o = cbase + row_off;
            *reinterpret_cast<uint2*>(p.dw + o) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
            *reinterpret_cast<uint2*>(p.dq + o) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
            *reinterpret_cast<uint2*>(p.dk + o) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
            *reinterpret_cast<uint2*>(p.dz + o) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
            *reinterpret_cast<uint2*>(p.da + o) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
        }
        WKV_STAMP(4)
        block_sync_lds();       // Z
        WKV_STAMP(5)
    }
    WKV_STAMP_FLUSH(0, 0, 6)
}

}  // namespace wkv7c
