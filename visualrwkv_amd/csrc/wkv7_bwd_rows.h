// WKV7 backward: the FULL-ROW memory role shared by the 12-wave kernels (wkv7_bwd_v8.h; the forward's wkv7_fwd_v4.h uses the lane
// map) -- gfx950.  Reference for the math: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130.
//
// Every input crosses the chip boundary as full rows by LDS-DMA (global_load_lds_dwordx4: 8 lanes = one 128-byte token row of a head,
// the XOR swizzle of the operand images applied on the SOURCE address): w q k z a and sa into a staging image the P waves read their
// own 8-byte pieces from, v and dy straight into the images the I and J waves read (a ring of four).  benchmarks/mem_role_probe.hip
// prices the access shape (profiles/r4_mem_role_probe.jsonl): "one token per lane, four channels = 8 bytes" 0.94 ms against 0.78 ms
// for the same bytes as full rows at 16 B per lane.  Staging is single buffered: the P waves lift their pieces into registers at the
// top of the step, meet on an LDS counter, and the next chunk's rows are requested into the same bytes.
// The first kernel built on these pieces (the v6 schedule with this memory role, "v7") is an A/B partner only and lives in
// benchmarks/experiments/wkv7_bwd_v7.h.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#include <wkv7_bwd_v6.h>     // Decay, decay_scan, TailRaw, Split16, regmm_pre, BoolTag and (through it) the v5 building blocks

namespace wkv7v7 {

using wkv7::BwdArgs;
using namespace wkv7c;
using namespace wkv7v5;      // IMG, HLI, img_off, f32_off, LaneAddr, lane_addr, ld16, st16, mfma32, dot64, mask_split, tiles_op, dma_state
using wkv7v6::Decay;
using wkv7v6::decay_scan;
using wkv7v6::TailRaw;
using wkv7v6::BoolTag;

struct ChunkImg7 {                   // per chunk; three alive: P builds c-2, I reads c-1, J reads c
    uint16_t opnd[8][IMG];           // Zt_h Zt_l Qt_h Qt_l Ah_h Ah_l Kh_h Kh_l      [t][j]
    uint16_t sa[2][IMG];             // SA_h  SA_l                                   [t][i]
    float cl[N];                     // c_L[j]
};

// ------------------------------------------------------------------------------------------ P: rows in, images, tail
struct RawP { uint2 w, q, k, z, a; float4 sa; };         // one lane's 4 channels of one token, from the staging image

// Full-row requests of one chunk: 18 instructions of 1 KB -- i = 2 arr + half for w q k z a (staging) and v dy (ring slot),
// then the four quarters of sa -- dealt round-robin to the four P waves (5 5 4 4).  A bf16 row of a head is 128 B = 8 lanes,
// an fp32 row 256 B = 16 lanes; LDS slot s' of row r receives source slot s' ^ (r & 7) (bf16) / s' ^ (r & 15) (fp32).
struct DmaLane { unsigned b16, f32; };                   // per-lane byte offsets inside a chunk of a (B,T,H,N) array
DEVFN DmaLane dma_lane(int lane, unsigned ts) {
    DmaLane d;
    const unsigned r8 = (unsigned)lane >> 3, r4 = (unsigned)lane >> 4;
    d.b16 = r8 * ts * 2u + 16u * (((unsigned)lane & 7u) ^ (r8 & 7u));          // rows 8 half + r8: (row & 7) == r8
    d.f32 = r4 * ts * 4u + 16u * (((unsigned)lane & 15u) ^ r4);                 // rows 4 qd + r4: (row & 15) == 4 qd + r4 -> ^ 4 qd below
    return d;
}
template <class LdsT, int NW = 4>                   // LdsT: .vdy .stg .stg_sa (wkv7_bwd_v8.h reuses these with its own layout); NW issuing waves, this one is w
DEVFN void dma_chunk(LdsT& lds, const BwdArgs& p, size_t chunk_base /* elements, uniform */, int c, int w, unsigned ts, const DmaLane& dl) {
    uint16_t* vd = lds.vdy[c & 3][0];
#pragma unroll
    for (int k = 0; k < (18 + NW - 1) / NW; ++k) {
        const int i = w + NW * k;                         // wave-uniform
        if (i >= 18) break;
        if (i < 14) {
            const int arr = i >> 1, half = i & 1;
            const uint16_t* src = arr == 0 ? p.w : arr == 1 ? p.q : arr == 2 ? p.k : arr == 3 ? p.z : arr == 4 ? p.a : arr == 5 ? p.v : p.dy;
            uint16_t* dst = (arr < 5 ? lds.stg[arr] : vd + (arr - 5) * IMG) + half * 8 * N;
            lds_dma16_sbase(src + chunk_base + (size_t)half * 8 * ts, dl.b16, dst);
        } else {
            const int qd = i - 14;
            lds_dma16_sbase(p.sa + chunk_base + (size_t)qd * 4 * ts, dl.f32 ^ (unsigned)(64 * qd), lds.stg_sa + qd * 4 * N);
        }
    }
}
template <class LdsT>
DEVFN RawP read_stage(const LdsT& lds, const LaneAddr& la) {
    RawP r;
    r.w = ld8(&lds.stg[0][la.own]); r.q = ld8(&lds.stg[1][la.own]); r.k = ld8(&lds.stg[2][la.own]);
    r.z = ld8(&lds.stg[3][la.own]); r.a = ld8(&lds.stg[4][la.own]);
    r.sa = *reinterpret_cast<const float4*>(&lds.stg_sa[la.f32]);
    return r;
}

DEVFN Decay prep7(ChunkImg7& B, const RawP& raw, int c16, int j0, const LaneAddr& la) {
    float q[4], k[4], z[4], a[4];
    unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
    const Decay d = decay_scan(raw.w);
    float zt[4], qt[4], ah[4], kh[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // c_t = 2^x ; c_{t-1} is the previous lane's c_t (1 for the first token)
        const float cc = fast_exp2(d.x2[e]), ic = fast_exp2(-d.x2[e]);
        const float cp = dpp_shr1_fill(cc, 1.f);
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; ah[e] = a[e] * ic; kh[e] = k[e] * ic; cend[e] = cc;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&B.opnd[0][la.own], hh); st8(&B.opnd[1][la.own], ll);
    split4(qt, hh, ll); st8(&B.opnd[2][la.own], hh); st8(&B.opnd[3][la.own], ll);
    split4(ah, hh, ll); st8(&B.opnd[4][la.own], hh); st8(&B.opnd[5][la.own], ll);
    split4(kh, hh, ll); st8(&B.opnd[6][la.own], hh); st8(&B.opnd[7][la.own], ll);
    const float sav[4] = {raw.sa.x, raw.sa.y, raw.sa.z, raw.sa.w};
    split4(sav, hh, ll); st8(&B.sa[0][la.own], hh); st8(&B.sa[1][la.own], ll);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
    return d;
}

// element-wise tail of one chunk: lane = token c16, channels 16 pw + 4g + e (the lane's own prep columns)
template <class LdsT>                               // LdsT: .res .glast .flag
DEVFN void tail7(LdsT& lds, int par, const TailRaw& tr, const BwdArgs& p, size_t u, unsigned lane_boff, int c16, int pw, int g, const LaneAddr& la) {
    const float4 zt4 = *reinterpret_cast<const float4*>(&lds.res[0][la.f32]);
    const float4 qt4 = *reinterpret_cast<const float4*>(&lds.res[1][la.f32]);
    const float4 ah4 = *reinterpret_cast<const float4*>(&lds.res[2][la.f32]);
    const float4 kh4 = *reinterpret_cast<const float4*>(&lds.res[3][la.f32]);
    const float4 gl4 = *reinterpret_cast<const float4*>(&lds.glast[par][16 * pw + 4 * g]);
    lds_flag_add(&lds.flag[4]);                           // (waits for the reads above) the J waves may overwrite `res`
    const float dZt[4] = {zt4.x, zt4.y, zt4.z, zt4.w}, dQt[4] = {qt4.x, qt4.y, qt4.z, qt4.w};
    const float dAh[4] = {ah4.x, ah4.y, ah4.z, ah4.w}, dKh[4] = {kh4.x, kh4.y, kh4.z, kh4.w};
    const float glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
    float q[4], k[4], z[4], a[4];
    unpack4(tr.q, q); unpack4(tr.k, k); unpack4(tr.z, z); unpack4(tr.a, a);
    float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x2 = tr.x2[e], l2 = x2 - dpp_shr1_fill(x2, 0.f);      // log2 c_t from the queue; log2 w_t = its difference along t
        const float cc = fast_exp2(x2), ic = fast_exp2(-x2), cp = dpp_shr1_fill(cc, 1.f);
        dz[e] = dZt[e] * cp; dq[e] = dQt[e] * cc; da[e] = dAh[e] * ic; dk[e] = dKh[e] * ic;
        // decay-gradient integrand g_t = dq q - da a - dk k + (dz z)[t+1]  (+ sum_i dS.S_L at the last token)
        float gt = dq[e] * q[e] - da[e] * a[e] - dk[e] * k[e] + dpp_shl<1>(dz[e] * z[e]);
        if (c16 == 15) gt += glv[e];
        gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
        dw[e] = gt * (l2 * LN2);
    }
    auto out = [&](uint16_t* base) { return reinterpret_cast<uint2*>(reinterpret_cast<char*>(base + u) + lane_boff); };   // uniform base + lane offset
    *out(p.dw) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
    *out(p.dq) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
    *out(p.dk) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
    *out(p.dz) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
    *out(p.da) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
}

// score gradients image[t][s] = dM[t][s]: D[m = s][n = t] = X_s . Y_t with X in {SA, V}, Y in {dR, dY}; pair images
// piece 0 dM_za  1 dM_zk  2 dM_qa  3 dM_qk
template <class LdsT>                               // LdsT: .dsc
DEVFN void dscores7(LdsT& lds, const uint16_t* sah, const uint16_t* sal, const uint16_t* vi, const uint16_t* dyi,
                    const uint16_t* drh, const uint16_t* drl, int piece, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (piece == 0) mask_split<false, false>(dot64<true, true>(sah, sal, drh, drl, la), c16, g, hh, ll);              // tril_(dR SA^T)
    else if (piece == 1) mask_split<false, false>(dot64<false, true>(vi, vi, drh, drl, la), c16, g, hh, ll);         // tril_(dR V^T)
    else if (piece == 2) mask_split<true, false>(dot64<true, false>(sah, sal, dyi, dyi, la), c16, g, hh, ll);        // tril(dY SA^T)
    else mask_split<true, false>(dot64<false, false>(vi, vi, dyi, dyi, la), c16, g, hh, ll);                         // tril(dY V^T)
    const int o = la.hl + 4 * (piece & 1);                   // za / qa first, zk / qk second of the pair
    st8(&lds.dsc[piece & 2][o], hh);
    st8(&lds.dsc[(piece & 2) + 1][o], ll);
}

}  // namespace wkv7v7
