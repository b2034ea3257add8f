// WKV6 backward, chunked MFMA form, three-role pipeline -- gfx950.
//
// Same math as bwd6_kernel (wkv6_chunked.h; reference: kernel_backward_111 / kernel_backward_222 of
// VisualRWKV-v6/v6.0/cuda/wkv6_cuda.cu:64-227), rescheduled like the WKV7 backward (wkv7_bwd_v6.h).  bwd6_kernel runs a chunk
// as prepare -> barrier -> scores, i-split, j-split, tail -> barrier on FOUR waves per (b, h): at B x H = 256 that is one wave
// per SIMD walking ~1100 dependent instructions per chunk (8.2k cycles, 0.29 of the HBM roofline at B = 4, 0.44 at B = 8 with two
// workgroups per CU; profiles/r4_wkv6_micro.jsonl), with 36 two-byte LDS scatter stores per lane for the transposed operand copies
// and four two-byte global stores per lane for dV.  Here twelve waves per (b, h) work on three consecutive chunks at once:
//
//   P (waves 8-11)  step n: requests S0 of chunk cp = nchunk-1-n (LDS-DMA, 4 KB per wave) and the rows of chunk cp-1 (registers),
//                   finishes chunk cp+2 (element-wise tail from the J waves' fp32 results, stores gr gk gw), prepares the images
//                   of chunk cp (decay scan, hi/lo operand images [16][64] bf16, XOR-swizzled: wkv7_bwd_v5.h)
//   I (waves 0-3)   chunk cp+1: A = tril(Rt Kh^T) + diag, dV = A^T dY + Kb dS^T (stored from registers, 8 B per lane: products
//                   whose result leaves the chip are issued with swapped operands so that lane = token), dS^T update
//   J (waves 4-7)   chunk cp+1: dA = dY V^T in both orientations, dRe = dY S0, dKb = V dS, dRa = dAl Kh, dKh = dAl^T Rt, the
//                   decay-gradient term sum_i dS S0 c_L, its own copy of dS ([i][j] tiles) and its update
//
// The I and J waves share nothing but the chunk's images (each keeps its own orientation of dL/dS, as bwd6_kernel does), and the
// tail runs a full step behind them: ONE workgroup barrier per step and no other synchronisation.  Operands whose contraction
// index is the token are fetched with ds_read_b64_tr_b16 from the row-major images: no transposed copies.
// LDS: 2 x 20.5 KB images + 2 x 16 KB S0 + 2 x 16 KB results = 106 KB, one workgroup per CU.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#include <wkv7_bwd_v5.h>     // IMG, img_off, f32_off, tix, LaneAddr, lane_addr, ld16, mfma32, dot64, tiles_op, dma_state
#include <wkv6_chunked.h>    // Bwd6Args, Decay6, decay_factors

namespace wkv6v2 {

using namespace wkv7c;       // N, L, mk8, split4, unpack4, st8, zero4
using wkv6c::Bwd6Args;
using wkv6c::Decay6;
using wkv6c::decay_factors;
using wkv7v5::IMG;
using wkv7v5::LaneAddr;
using wkv7v5::dma_state;
using wkv7v5::dot64;
using wkv7v5::f32_off;
using wkv7v5::lane_addr;
using wkv7v5::ld16;
using wkv7v5::mfma32;
using wkv7v5::tiles_op;
using wkv7v5::tix;

enum { RT_H, RT_L, KH_H, KH_L, KB_H, KB_L, RE_H, RE_L, VV, DY, NIMG };
struct Chunk6 {
    uint16_t img[NIMG][IMG];         // Rt Kh Kb Re hi, lo [t][j]; v, dy [t][i]
    float cl[N];                     // c_L[j]
    float dpart[4][L];               // per P wave: sum over its 16 key columns of r u k
};
struct Lds6V2 {
    Chunk6 b[2];                     // by chunk parity: written by P in step n, read by I / J in step n + 1
    float s0[2][N * N];              // chunk-start state s[c] ([j][i] fp32, f32_off swizzle), LDS-DMA in step n for the J waves' step n + 1
    float res[2][4][IMG];            // dRe dRa dKh dKb of the J waves' chunk, token-per-lane fp32 (f32_off): the tail of step n + 1
    float dd[2][L];                  // diagonal of dA
    float glast[2][N];               // c_L[j] sum_i dS[i][j] S0[i][j]
};
static_assert(sizeof(Lds6V2) <= 160 * 1024, "LDS budget");

struct Raw6 { uint2 r, k, v, gy; float4 ew; };
struct Tail6 { uint2 r, k; float ew[4], e_re[4], e_r[4], e_h[4], e_b[4]; };     // what the tail of a chunk needs from its prepare, two steps later

__global__ __launch_bounds__(768) void bwd6_kernel_v2(Bwd6Args p) {
    Lds6V2& lds = *reinterpret_cast<Lds6V2*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int role = wave >> 2, w = wave & 3;          // 0: I, 1: J, 2: P
    const int c16 = lane & 15, g = lane >> 4;
    const size_t ts = (size_t)H * N;
    const int hh = blockIdx.x % H;
    const size_t head_base = ((size_t)(blockIdx.x / H) * T * H + hh) * N;
    const int nchunk = (T + L - 1) / L;
    const int nsteps = nchunk + 2;
    const LaneAddr la = lane_addr(c16, g, w);
    const int j0 = 16 * w + 4 * g;                      // P: token c16, channels j0..j0+3;  I / J results: the same piece
    const float* sbase = p.s + (size_t)blockIdx.x * nchunk * N * N;

    if (role == 2) {
        // ================================================================== P
        float uu[4];
        unpack4(*reinterpret_cast<const uint2*>(p.u + (size_t)hh * N + j0), uu);
        auto fetch = [&](Raw6& rc, int c) {
            const int tt = c * L + c16;
            if (tt < T) {
                const size_t o = head_base + (size_t)tt * ts + j0;
                rc.r = *reinterpret_cast<const uint2*>(p.r + o); rc.k = *reinterpret_cast<const uint2*>(p.k + o);
                rc.v = *reinterpret_cast<const uint2*>(p.v + o); rc.gy = *reinterpret_cast<const uint2*>(p.gy + o);
                rc.ew = *reinterpret_cast<const float4*>(p.ew + o);
            } else {
                rc.r = make_uint2(0, 0); rc.k = rc.r; rc.v = rc.r; rc.gy = rc.r;
                rc.ew = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        Raw6 rc;
        fetch(rc, nchunk - 1);
        Tail6 q0{}, q1{};                                   // prepared one step ago | two steps ago (the tail's chunk)
        float gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < nsteps; ++n) {
            const int cp = nchunk - 1 - n, ct = cp + 2;
            const Raw6 cur = rc;
            // requests first (S0 for the J waves' next step, the rows of the next chunk), the tail's stores after them: the wait
            // before the barrier leaves the three stores in flight
            if (cp >= 0) dma_state(lds.s0[cp & 1], sbase + (size_t)cp * N * N, 4 * w, 4 * w + 4, lane);
            if (cp >= 1) fetch(rc, cp - 1);
            // ---------------------------------------------------------------- tail of chunk ct (token c16, channels j0..j0+3)
            if (ct <= nchunk - 1) {
                const int pb = ct & 1;
                const float4 x0 = *reinterpret_cast<const float4*>(&lds.res[pb][0][la.f32]);
                const float4 x1 = *reinterpret_cast<const float4*>(&lds.res[pb][1][la.f32]);
                const float4 x2 = *reinterpret_cast<const float4*>(&lds.res[pb][2][la.f32]);
                const float4 x3 = *reinterpret_cast<const float4*>(&lds.res[pb][3][la.f32]);
                const float4 g4 = *reinterpret_cast<const float4*>(&lds.glast[pb][j0]);
                const float dd = lds.dd[pb][c16];
                const float vre[4] = {x0.x, x0.y, x0.z, x0.w}, vra[4] = {x1.x, x1.y, x1.z, x1.w};
                const float vkh[4] = {x2.x, x2.y, x2.z, x2.w}, vkb[4] = {x3.x, x3.y, x3.z, x3.w};
                const float gls[4] = {g4.x, g4.y, g4.z, g4.w};
                float r[4], k[4], gr[4], gk[4], gw[4];
                unpack4(q1.r, r); unpack4(q1.k, k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ddu = dd * uu[e];
                    gr[e] = vre[e] * q1.e_re[e] + vra[e] * q1.e_r[e] + ddu * k[e];
                    gk[e] = vkh[e] * q1.e_h[e] + vkb[e] * q1.e_b[e] + ddu * r[e];
                    gu_acc[e] = fmaf(dd * r[e], k[e], gu_acc[e]);
                    const float pa = vra[e] * (r[e] * q1.e_r[e]);                      // dRa Rt
                    const float pr = vre[e] * (r[e] * q1.e_re[e]) + pa;                // dRe Re + dRa Rt
                    const float ph = vkh[e] * (k[e] * q1.e_h[e]);
                    const float pbb = vkb[e] * (k[e] * q1.e_b[e]);
                    float gx = dpp_shl<1>(pr) - ph - pbb;
                    const float sum_b = group_sum<4>(pbb), sum_m = group_sum<4>(ph - pa);
                    if (c16 == 15) gx += sum_b + gls[e];
                    if (c16 == 7) gx += sum_m;
                    gx += dpp_shl<1>(gx); gx += dpp_shl<2>(gx); gx += dpp_shl<4>(gx); gx += dpp_shl<8>(gx);   // suffix sum over t
                    gw[e] = gx * q1.ew[e];
                }
                if (ct * L + c16 < T) {
                    const size_t o = head_base + (size_t)(ct * L + c16) * ts + j0;
                    *reinterpret_cast<uint2*>(p.gr + o) = make_uint2(cvt_pk_bf16(gr[0], gr[1]), cvt_pk_bf16(gr[2], gr[3]));
                    *reinterpret_cast<uint2*>(p.gk + o) = make_uint2(cvt_pk_bf16(gk[0], gk[1]), cvt_pk_bf16(gk[2], gk[3]));
                    *reinterpret_cast<uint2*>(p.gw + o) = make_uint2(cvt_pk_bf16(gw[0], gw[1]), cvt_pk_bf16(gw[2], gw[3]));
                }
            }
            q1 = q0;
            // ---------------------------------------------------------------- images of chunk cp
            if (cp >= 0) {
                Chunk6& B = lds.b[cp & 1];
                float r[4], k[4];
                unpack4(cur.r, r); unpack4(cur.k, k);
                const float ew[4] = {cur.ew.x, cur.ew.y, cur.ew.z, cur.ew.w};
                const Decay6 d = decay_factors(ew, lane);
                float re[4], rt[4], kh[4], kb[4], dp = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[e] = r[e] * d.e_re[e]; rt[e] = r[e] * d.e_r[e]; kh[e] = k[e] * d.e_h[e]; kb[e] = k[e] * d.e_b[e];
                    dp = fmaf(r[e] * uu[e], k[e], dp);
                    q0.ew[e] = ew[e]; q0.e_re[e] = d.e_re[e]; q0.e_r[e] = d.e_r[e]; q0.e_h[e] = d.e_h[e]; q0.e_b[e] = d.e_b[e];
                }
                q0.r = cur.r; q0.k = cur.k;
                dp += lane_xor16(dp);
                dp += lane_xor32(dp);
                if (g == 0) B.dpart[w][c16] = dp;
                uint2 h, l;
                split4(rt, h, l); st8(&B.img[RT_H][la.own], h); st8(&B.img[RT_L][la.own], l);
                split4(kh, h, l); st8(&B.img[KH_H][la.own], h); st8(&B.img[KH_L][la.own], l);
                split4(kb, h, l); st8(&B.img[KB_H][la.own], h); st8(&B.img[KB_L][la.own], l);
                split4(re, h, l); st8(&B.img[RE_H][la.own], h); st8(&B.img[RE_L][la.own], l);
                st8(&B.img[VV][la.own], cur.v);
                st8(&B.img[DY][la.own], cur.gy);
                if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(d.c_l[0], d.c_l[1], d.c_l[2], d.c_l[3]);
            }
            vmem_wait<3>();
            block_sync_lds();
        }
        // gu[b, h, j] = sum over the tokens of this sample
#pragma unroll
        for (int e = 0; e < 4; ++e) gu_acc[e] = group_sum<4>(gu_acc[e]);
        if (c16 == 0)
            *reinterpret_cast<uint2*>(p.gu + (size_t)blockIdx.x * N + j0) =
                make_uint2(cvt_pk_bf16(gu_acc[0], gu_acc[1]), cvt_pk_bf16(gu_acc[2], gu_acc[3]));
        return;
    }

    if (role == 0) {
        // ================================================================== I: value column i = 16w + c16
        f32x4 dS1[4];                                       // dS1[jb][r] = dS[i][j = tix(jb, 4g+r)]
#pragma unroll
        for (int x = 0; x < 4; ++x) dS1[x] = zero4();
        for (int n = 0; n < nsteps; ++n) {
            const int ci = nchunk - n;
            if (ci >= 0 && ci <= nchunk - 1) {
                const Chunk6& B = lds.b[ci & 1];
                // A[t][s] = sum_j Rt[t][j] Kh[s][j]: lane s = c16, registers t = 4g + r; strictly lower part + the diagonal sum_j r u k
                f32x4 ac = dot64<true, true>(B.img[RT_H], B.img[RT_L], B.img[KH_H], B.img[KH_L], la);
                const float dt_row = B.dpart[0][c16] + B.dpart[1][c16] + B.dpart[2][c16] + B.dpart[3][c16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int t = 4 * g + r4;
                    ac[r4] = c16 < t ? ac[r4] : (c16 == t ? dt_row : 0.f);
                }
                uint2 ah, al;
                split4(ac, ah, al);
                const uint2 dyv = lds_read_tr16(&B.img[DY][la.trc]);             // dY[4g+e][i]
                const bf16x8 bdy = mk8(dyv, dyv);
                bf16x8 sh[2], sl[2];
                tiles_op(dS1, sh, sl);
                // dV^T[i][s] = sum_t dY[t][i] A[t][s] + sum_j dS[i][j] Kb[s][j]: lane = token s, registers = channels 16w + 4g + r
                f32x4 dV = mfma32(bdy, mk8(ah, al), zero4());
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 kbh = ld16(&B.img[KB_H][la.row[kb]]);
                    dV = mfma32(sh[kb], kbh, dV);
                    dV = mfma32(sl[kb], kbh, dV);
                    dV = mfma32(sh[kb], ld16(&B.img[KB_L][la.row[kb]]), dV);
                }
                if (ci * L + c16 < T)
                    *reinterpret_cast<uint2*>(p.gv + head_base + (size_t)(ci * L + c16) * ts + j0) =
                        make_uint2(cvt_pk_bf16(dV[0], dV[1]), cvt_pk_bf16(dV[2], dV[3]));
                // dS^T <- diag(c_L) dS^T + Re^T dY
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
                    f32x4 acc = dS1[jb];
                    acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
                    const int o = la.tri[jb >> 1] + 4 * (jb & 1);
                    dS1[jb] = mfma32(mk8(lds_read_tr16(&B.img[RE_H][o]), lds_read_tr16(&B.img[RE_L][o])), bdy, acc);
                }
            }
            block_sync_lds();
        }
        return;
    }

    // ====================================================================== J: key column j = 16w + c16
    const int j = 16 * w + c16;
    f32x4 dS2[4];                                           // dS2[ib][r] = dS[i = tix(ib, 4g+r)][j]
#pragma unroll
    for (int x = 0; x < 4; ++x) dS2[x] = zero4();
    for (int n = 0; n < nsteps; ++n) {
        const int ci = nchunk - n;
        if (ci >= 0 && ci <= nchunk - 1) {
            const Chunk6& B = lds.b[ci & 1];
            const int pb = ci & 1;
            const float clj = B.cl[j];
            f32x4 S0[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float4 x = *reinterpret_cast<const float4*>(&lds.s0[pb][f32_off(j, tix(ib, 4 * g))]);
                S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
            }
            // dA[t][s] = sum_i dY[t][i] V[s][i] in both orientations (exact bf16 operands)
            f32x4 da = dot64<false, false>(B.img[DY], nullptr, B.img[VV], nullptr, la);        // lane s = c16, registers t = 4g + r
            f32x4 dat = dot64<false, false>(B.img[VV], nullptr, B.img[DY], nullptr, la);       // lane t = c16, registers s = 4g + r
            {
                const int r = c16 & 3;                      // the diagonal element of column c16: lane group g == c16 >> 2, register c16 & 3
                const float d01 = r & 1 ? da[1] : da[0], d23 = r & 1 ? da[3] : da[2];
                if (w == 0 && (c16 >> 2) == g) lds.dd[pb][c16] = r & 2 ? d23 : d01;
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = 4 * g + r4;
                da[r4] = m > c16 ? da[r4] : 0.f;            // t > s
                dat[r4] = m < c16 ? dat[r4] : 0.f;          // s < t
            }
            bf16x8 s0h[2], s0l[2], d2h[2], d2l[2];
            tiles_op(S0, s0h, s0l);
            tiles_op(dS2, d2h, d2l);
            // results D[m = j][n = t]: lane = token, registers = channels 16w + 4g + r
            f32x4 dRe = zero4(), dKb = zero4();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 dyr = ld16(&B.img[DY][la.row[kb]]), vr = ld16(&B.img[VV][la.row[kb]]);
                dRe = mfma32(s0h[kb], dyr, dRe);            // dY S0
                dRe = mfma32(s0l[kb], dyr, dRe);
                dKb = mfma32(d2h[kb], vr, dKb);             // V dS
                dKb = mfma32(d2l[kb], vr, dKb);
            }
            f32x4 dRa, dKh;
            {
                uint2 th, tl;
                split4(dat, th, tl);
                const uint2 kh_h = lds_read_tr16(&B.img[KH_H][la.trc]), kh_l = lds_read_tr16(&B.img[KH_L][la.trc]);      // Kh[4g+e][j]
                dRa = mfma32(mk8(kh_h, kh_h), mk8(th, tl), zero4());                       // dAl Kh
                dRa = mfma32(mk8(kh_l.x, kh_l.y, 0u, 0u), mk8(th.x, th.y, 0u, 0u), dRa);
                split4(da, th, tl);
                const uint2 rt_h = lds_read_tr16(&B.img[RT_H][la.trc]), rt_l = lds_read_tr16(&B.img[RT_L][la.trc]);
                dKh = mfma32(mk8(rt_h, rt_h), mk8(th, tl), zero4());                       // dAl^T Rt
                dKh = mfma32(mk8(rt_l.x, rt_l.y, 0u, 0u), mk8(th.x, th.y, 0u, 0u), dKh);
            }
            *reinterpret_cast<float4*>(&lds.res[pb][0][la.f32]) = make_float4(dRe[0], dRe[1], dRe[2], dRe[3]);
            *reinterpret_cast<float4*>(&lds.res[pb][1][la.f32]) = make_float4(dRa[0], dRa[1], dRa[2], dRa[3]);
            *reinterpret_cast<float4*>(&lds.res[pb][2][la.f32]) = make_float4(dKh[0], dKh[1], dKh[2], dKh[3]);
            *reinterpret_cast<float4*>(&lds.res[pb][3][la.f32]) = make_float4(dKb[0], dKb[1], dKb[2], dKb[3]);
            // decay-gradient term of the chunk's last token, then dS <- dS diag(c_L) + dY^T Re
            float gl = 0.f;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) gl = fmaf(dS2[ib][r4], S0[ib][r4], gl);
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            if (g == 0) lds.glast[pb][j] = gl * clj;
            const bf16x8 bre = mk8(lds_read_tr16(&B.img[RE_H][la.trc]), lds_read_tr16(&B.img[RE_L][la.trc]));       // Re[4g+e][j] hi | lo
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                f32x4 acc = dS2[ib];
                acc[0] *= clj; acc[1] *= clj; acc[2] *= clj; acc[3] *= clj;
                const uint2 dyi = lds_read_tr16(&B.img[DY][la.tri[ib >> 1] + 4 * (ib & 1)]);                         // dY[4g+e][i = tix(ib, c16)]
                dS2[ib] = mfma32(mk8(dyi, dyi), bre, acc);
            }
        }
        block_sync_lds();
    }
}

}  // namespace wkv6v2
