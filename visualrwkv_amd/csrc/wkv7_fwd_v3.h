// WKV7 forward, chunked MFMA form, producer/consumer wave specialisation -- gfx950.
//
// Same algorithm and numerics class as wkv7_chunked.h (see its header), restructured after profiling the
// 4-wave kernel on MI355X (profiles/r1_wkv7_pmc_b8.txt): MFMA pipe 9 % busy, ~580 VALU instructions per wave
// and chunk, 39 % of wave time in s_waitcnt/barriers -- the prefetched input loads and the output stores
// shared one vmcnt counter (a wait for the loads also waited out the HBM write round trip of the stores), and
// the phases (prep -> scores -> main) ran back to back on one wave per SIMD.
//
// One workgroup = 8 waves per (b,h):
//   producers (waves 4..7) run ONE CHUNK AHEAD: global loads (prefetched a further chunk ahead into registers),
//       decay scan / scaling / hi-lo split of key-columns j in [16p,16p+16) into LDS buffer (c+1)&1, barrier A,
//       then one score matrix each (M_zk, M_qa, M_qk; producer 0 builds T = (I - M_za)^-1 by nilpotent
//       doubling on fp32 MFMA, register resident), barrier B.  They issue no global stores.
//   consumers (waves 0..3) own value-columns i in [16w,16w+16) of S^T as accumulator tiles and run the
//       S -> R -> SA -> Y -> S chain of chunk c from buffer c&1, plus all stores.  They issue no global loads.
// Two workgroup barriers per chunk (LDS-only: s_waitcnt lgkmcnt(0); s_barrier).  On each SIMD a producer and a
// consumer wave are co-resident, so VALU-heavy preparation overlaps the MFMA chain.  Register-resident 16x16
// products (T doubling, T*R, M_qa*SA) use the f32 MFMA (exact fp32, no operand splitting); products against
// LDS-resident operands stay bf16x3.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>

namespace wkv7c {

constexpr int SF = 20;   // fp32 [t][s] image row stride (80 B)
constexpr int TS = 80;   // [t][j] row stride (160 B) of images read with ds_read_b64_tr_b16: the 8 rows x 32 B a half-wave
                         // touches fall into 8 distinct 32-byte bank groups

struct BufF {
    uint16_t opnd[8][L][TJ];      // Zt Qt Ah Kh (hi,lo)  [t][j]
    union {                       // the operands whose MFMA k index is the token:
        struct {                  //   transposed copies built by the producers with 2-byte LDS stores ...
            uint16_t trn[4][N][JT];       // Ab Kb (hi,lo)        [j][t]
            uint16_t vt[N][JT];           // v                    [i][t]
        };
        struct {                  //   ... or (TRD) natural row-major images read with ds_read_b64_tr_b16
            uint16_t abn[4][L][TS];       // Ab_hi Ab_lo Kb_hi Kb_lo  [t][j]
            uint16_t vn[L][TS];           // v                        [t][i]
        };
    };
    uint16_t scb[2][2][L][SS];    // 0 M_zk  1 M_qk ; [hi,lo][t][s]
    float scf[2][L][SF];          // 0 M_qa  1 T    ; fp32 [t][s]
    float cl[N];                  // c_L[j]
};
struct LdsF { BufF b[2]; unsigned prep_done; unsigned pad_[3]; };   // prep_done: producer-only hand-off counter

// D = P*Q on the f32 matrix core: pt = P^T in C layout (A operand), qc = Q in C layout (B operand);
// MFMA #r contracts the k-slots (g) <-> index 4g+r, which is exactly register r of both fragments.
DEVFN f32x4 regmm_f32(f32x4 pt, f32x4 qc) {
    f32x4 acc = zero4();
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = mfma_16x16x4_f32(pt[r], qc[r], acc);
    return acc;
}
// The same product on the bf16 matrix core with split operands (P_h Q_h + P_h Q_l + P_l Q_h, ~2^-16 relative):
//   MFMA 1:  A = [P_h | P_h],  B = [Q_h ; Q_l]       MFMA 2:  A = [P_l | 0],  B = [Q_h ; 0]
// (k-slots 0-3 and 4-7 of lane group g both stand for index 4g+e).  2 x 16 cycles instead of 4 dependent f32 MFMAs
// of 32-40 cycles: the nilpotent doubling for T is the critical path of the producers' second half.
DEVFN f32x4 regmm_bf16x3(f32x4 pt, f32x4 qc) {
    uint2 ph, pl, qh, ql;
    split4(pt, ph, pl);
    split4(qc, qh, ql);
    const f32x4 acc = mfma_16x16x32_bf16(mk8(ph, ph), mk8(qh, ql), zero4());
    return mfma_16x16x32_bf16(mk8(pl.x, pl.y, 0u, 0u), mk8(qh.x, qh.y, 0u, 0u), acc);
}
// regmm_bf16x3 with the operands already split (a level of the doubling uses every matrix twice)
struct SplitF { uint2 h, l; };
DEVFN SplitF splitf(f32x4 x) { SplitF s; split4(x, s.h, s.l); return s; }
DEVFN f32x4 regmm_pre(const SplitF& p_, const SplitF& q_) {
    const f32x4 acc = mfma_16x16x32_bf16(mk8(p_.h, p_.h), mk8(q_.h, q_.l), zero4());
    return mfma_16x16x32_bf16(mk8(p_.l.x, p_.l.y, 0u, 0u), mk8(q_.h.x, q_.h.y, 0u, 0u), acc);
}
// acc += M[row c16][s] * B[s][col] with M an fp32 [t][s] image in LDS and B a C-layout fragment
DEVFN f32x4 mm_f32_image(f32x4 acc, const float (*M)[SF], int c16, int g, f32x4 bfrag) {
    const float4 m = *reinterpret_cast<const float4*>(&M[c16][4 * g]);
    acc = mfma_16x16x4_f32(m.x, bfrag[0], acc);
    acc = mfma_16x16x4_f32(m.y, bfrag[1], acc);
    acc = mfma_16x16x4_f32(m.z, bfrag[2], acc);
    acc = mfma_16x16x4_f32(m.w, bfrag[3], acc);
    return acc;
}

template <bool SFX, bool TRD = false, bool NOAB = false>
DEVFN void prep_v3(BufF& B, const RawChunk& rc, int pw, int lane) {
    const int t = lane & 15, g = lane >> 4, j0 = 16 * pw + 4 * g;
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(rc.w, wr); unpack4(rc.q, q); unpack4(rc.k, k); unpack4(rc.z, z); unpack4(rc.a, a);
    float zt[4], qt[4], ah[4], kh[4], ab[4], kb[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp(wr[e]);
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        float rest = 0.f;                              // sum of log-decays of the later tokens of the chunk
        if (NOAB) {                                    // no Ab / Kb at all: the state update uses Ah / Kh and scales by c_L afterwards
        } else if (SFX) {                                     // exclusive suffix scan with DPP row shifts (no LDS round trip)
            float y = lw;
            y += dpp_shl<1>(y); y += dpp_shl<2>(y); y += dpp_shl<4>(y); y += dpp_shl<8>(y);
            rest = y - lw;
        } else {
            rest = lane_bcast(x, (lane & 48) | 15) - x;
        }
        const float c = fast_exp(x), cp = fast_exp(x - lw), ic = fast_exp(-x), cb = NOAB ? 0.f : fast_exp(rest);
        zt[e] = z[e] * cp; qt[e] = q[e] * c; ah[e] = a[e] * ic; kh[e] = k[e] * ic;
        ab[e] = a[e] * cb; kb[e] = k[e] * cb; cend[e] = c;
    }
    uint2 h, l;
    split4(zt, h, l); st8(&B.opnd[0][t][j0], h); st8(&B.opnd[1][t][j0], l);
    split4(qt, h, l); st8(&B.opnd[2][t][j0], h); st8(&B.opnd[3][t][j0], l);
    split4(ah, h, l); st8(&B.opnd[4][t][j0], h); st8(&B.opnd[5][t][j0], l);
    split4(kh, h, l); st8(&B.opnd[6][t][j0], h); st8(&B.opnd[7][t][j0], l);
    uint2 abh, abl, kbh, kbl;                          // the operands of the state update whose k index is the token
    if (NOAB) {                                        // Ah / Kh once more (their split is already done): S_L = diag(c_L)(S0 + Ah^T SA + Kh^T V)
        split4(ah, abh, abl); split4(kh, kbh, kbl);    // (common sub-expressions of the two splits above)
    } else {
        split4(ab, abh, abl); split4(kb, kbh, kbl);
    }
    if (TRD) {
        st8(&B.abn[0][t][j0], abh); st8(&B.abn[1][t][j0], abl);
        st8(&B.abn[2][t][j0], kbh); st8(&B.abn[3][t][j0], kbl);
        st8(&B.vn[t][j0], rc.v);
    } else {
        B.trn[0][j0 + 0][t] = (uint16_t)abh.x; B.trn[0][j0 + 1][t] = (uint16_t)(abh.x >> 16);
        B.trn[0][j0 + 2][t] = (uint16_t)abh.y; B.trn[0][j0 + 3][t] = (uint16_t)(abh.y >> 16);
        B.trn[1][j0 + 0][t] = (uint16_t)abl.x; B.trn[1][j0 + 1][t] = (uint16_t)(abl.x >> 16);
        B.trn[1][j0 + 2][t] = (uint16_t)abl.y; B.trn[1][j0 + 3][t] = (uint16_t)(abl.y >> 16);
        B.trn[2][j0 + 0][t] = (uint16_t)kbh.x; B.trn[2][j0 + 1][t] = (uint16_t)(kbh.x >> 16);
        B.trn[2][j0 + 2][t] = (uint16_t)kbh.y; B.trn[2][j0 + 3][t] = (uint16_t)(kbh.y >> 16);
        B.trn[3][j0 + 0][t] = (uint16_t)kbl.x; B.trn[3][j0 + 1][t] = (uint16_t)(kbl.x >> 16);
        B.trn[3][j0 + 2][t] = (uint16_t)kbl.y; B.trn[3][j0 + 3][t] = (uint16_t)(kbl.y >> 16);
        B.vt[j0 + 0][t] = (uint16_t)rc.v.x; B.vt[j0 + 1][t] = (uint16_t)(rc.v.x >> 16);
        B.vt[j0 + 2][t] = (uint16_t)rc.v.y; B.vt[j0 + 3][t] = (uint16_t)(rc.v.y >> 16);
    }
    if (t == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}

// D[x_row][y_row] = sum_j X[x_row][j] Y[y_row][j]  (bf16x3 from the [t][j] images)
DEVFN f32x4 score_v3(const BufF& B, int mx, int my, int c16, int g) {
    f32x4 a0 = zero4(), a1 = zero4();
    {
        const bf16x8 xh = ld_nat(B.opnd[mx], c16, 0, g), xl = ld_nat(B.opnd[mx + 1], c16, 0, g);
        const bf16x8 yh = ld_nat(B.opnd[my], c16, 0, g), yl = ld_nat(B.opnd[my + 1], c16, 0, g);
        a0 = mfma_16x16x32_bf16(xh, yh, a0); a0 = mfma_16x16x32_bf16(xh, yl, a0); a0 = mfma_16x16x32_bf16(xl, yh, a0);
    }
    {
        const bf16x8 xh = ld_nat(B.opnd[mx], c16, 1, g), xl = ld_nat(B.opnd[mx + 1], c16, 1, g);
        const bf16x8 yh = ld_nat(B.opnd[my], c16, 1, g), yl = ld_nat(B.opnd[my + 1], c16, 1, g);
        a1 = mfma_16x16x32_bf16(xh, yh, a1); a1 = mfma_16x16x32_bf16(xh, yl, a1); a1 = mfma_16x16x32_bf16(xl, yh, a1);
    }
    a0[0] += a1[0]; a0[1] += a1[1]; a0[2] += a1[2]; a0[3] += a1[3];
    return a0;
}

template <bool SHARED_SPLIT = false>
DEVFN void scores_v3(BufF& B, int pw, int lane) {
    const int c16 = lane & 15, g = lane >> 4;
    if (pw == 1 || pw == 3) {                 // transposed scores against Kh: lane (g, c16 = t), reg r <-> s = 4g+r
        f32x4 d = score_v3(B, 6, pw == 1 ? 0 : 2, c16, g);      // (Kh Zt^T) = M_zk^T ,  (Kh Qt^T) = M_qk^T
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * g + r;
            d[r] = (pw == 1 ? (s < c16) : (s <= c16)) ? d[r] : 0.f;
        }
        uint2 h, l;
        split4(d, h, l);
        st8(&B.scb[pw == 1 ? 0 : 1][0][c16][4 * g], h);
        st8(&B.scb[pw == 1 ? 0 : 1][1][c16][4 * g], l);
    } else if (pw == 2) {                     // (Ah Qt^T)[s][t] = M_qa[t][s], kept in fp32
        f32x4 d = score_v3(B, 4, 2, c16, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = (4 * g + r <= c16) ? d[r] : 0.f;
        *reinterpret_cast<float4*>(&B.scf[0][c16][4 * g]) = make_float4(d[0], d[1], d[2], d[3]);
    } else {                                  // T^T in C layout by doubling, all on the f32 matrix core
        // X = tril_strict(Zt Ah^T) in C layout; its transpose comes from a bounce through this wave's own output slot
        // (scf[1], written for real at the end) instead of a second 64-deep score product: 4 + 1 LDS instructions
        // instead of 8 ds_read_b128 + 6 MFMAs on the producers' critical path.
        f32x4 X = score_v3(B, 0, 4, c16, g), XT, TT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r] = (c16 < 4 * g + r) ? X[r] : 0.f;
            B.scf[1][4 * g + r][c16] = X[r];
        }
        wave_lds_fence();
        {
            const float4 t4 = *reinterpret_cast<const float4*>(&B.scf[1][c16][4 * g]);     // X[c16][4g..4g+3]
            XT[0] = t4.x; XT[1] = t4.y; XT[2] = t4.z; XT[3] = t4.w;
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) TT[r] = XT[r] + ((4 * g + r == c16) ? 1.f : 0.f);
        if (SHARED_SPLIT) {                    // every matrix of a level is split once (4 splits instead of 6 per level)
            SplitF sx = splitf(X), sxt = splitf(XT);
#pragma unroll
            for (int level = 0; level < 3; ++level) {
                const f32x4 X2 = regmm_pre(sxt, sx);
                f32x4 XT2 = XT;
                if (level < 2) XT2 = regmm_pre(sx, sxt);
                const SplitF sx2 = splitf(X2);
                const f32x4 D = regmm_pre(sx2, splitf(TT));
#pragma unroll
                for (int r = 0; r < 4; ++r) TT[r] += D[r];
                X = X2; XT = XT2;
                sx = sx2;
                if (level < 2) sxt = splitf(XT2);
            }
        } else {
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const f32x4 X2 = regmm_bf16x3(XT, X);
            f32x4 XT2 = XT;
            if (level < 2) XT2 = regmm_bf16x3(X, XT);
            const f32x4 D = regmm_bf16x3(X2, TT);
#pragma unroll
            for (int r = 0; r < 4; ++r) TT[r] += D[r];
            X = X2; XT = XT2;
        }
        }
        *reinterpret_cast<float4*>(&B.scf[1][c16][4 * g]) = make_float4(TT[0], TT[1], TT[2], TT[3]);   // T[c16][4g+r]
    }
}

// PF: chunks of input prefetch held in registers by the producers (HBM latency under load exceeds one chunk time);
// SFX: decay suffix by DPP scan instead of ds_bpermute.
// NOAB: no Ab / Kb images (the state update multiplies by c_L afterwards) and a T chain that splits every matrix once per level.
// ISPLIT: two workgroups per (b,h), each owning 32 of the 64 value rows of the state (rows are independent in the forward:
// sa_i, y_i and S[i][:] only involve row i) -- for few heads (B*H <= 128 leaves half of the 256 CUs idle): every workgroup
// repeats the producers' work, the consumer chain is halved (consumer waves 2, 3 only keep the barriers company).
template <bool PROF, bool WIDE = true, int PRIO = 1, int PF = 1, bool SFX = false, bool TRD = false, bool NOAB = false, bool ISPLIT = false>
__global__ __launch_bounds__(512) void fwd_kernel_v3(FwdArgs p) {
    LdsF& lds = *reinterpret_cast<LdsF*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);                              // token stride (elements)
    const unsigned bh = ISPLIT ? blockIdx.x >> 1 : blockIdx.x;        // (b, h)
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    WKV_STAMP_DECL

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        if (PRIO > 0) wave_priority<PRIO>();             // producers are the younger half: without this they lose VALU arbitration
        const unsigned lane_off = (unsigned)c16 * ts + 16u * pw + 4u * g;    // token c16, columns 16pw+4g..+3
        const uint16_t *pw_ = p.w + head_base, *pq = p.q + head_base, *pk = p.k + head_base;
        const uint16_t *pz = p.z + head_base, *pa = p.a + head_base, *pv = p.v + head_base;
        auto fetch = [&](RawChunk& rc, int c) {
            const size_t o = (size_t)c * L * ts + lane_off;
            rc.w = *reinterpret_cast<const uint2*>(pw_ + o); rc.q = *reinterpret_cast<const uint2*>(pq + o);
            rc.k = *reinterpret_cast<const uint2*>(pk + o); rc.z = *reinterpret_cast<const uint2*>(pz + o);
            rc.a = *reinterpret_cast<const uint2*>(pa + o); rc.v = *reinterpret_cast<const uint2*>(pv + o);
        };
        RawChunk rc, rc2;
        fetch(rc, 0);
        if (PF > 1 && nchunk > 1) fetch(rc2, 1);
        block_sync_lds();                               // prep_done is zeroed
        for (int c = 0; c <= nchunk; ++c) {            // iteration c produces chunk c (one ahead of the consumers)
            if (c < nchunk) {
                RawChunk cur = rc;
                if (PF > 1) {
                    rc = rc2;
                    if (c + 2 < nchunk) fetch(rc2, c + 2);
                } else if (c + 1 < nchunk) fetch(rc, c + 1);
                prep_v3<SFX, TRD, NOAB>(lds.b[c & 1], cur, pw, lane);
                lds_flag_add(&lds.prep_done);
            }
            WKV_STAMP(0)
            // A: the scores need the operand images of all four producer waves.  Only the producers wait (a counter,
            // not s_barrier): the consumers' second half does not depend on anything produced in this iteration.
            if (c < nchunk) lds_flag_wait(&lds.prep_done, 4u * (unsigned)(c + 1));
            WKV_STAMP(1)
            if (c < nchunk) scores_v3<NOAB>(lds.b[c & 1], pw, lane);
            WKV_STAMP(2)
            block_sync_lds();                           // B
            WKV_STAMP(3)
        }
        WKV_STAMP_FLUSH(256, 8, 4)
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wt = ISPLIT ? 2 * (int)(blockIdx.x & 1) + wave : wave;      // which 16 value rows this wave owns
    const bool idle = ISPLIT && wave >= 2;                                // wave-uniform
    if (idle) {                                                           // same barrier sequence as the working consumers
        block_sync_lds(); block_sync_lds();
        for (int c = 0; c < nchunk; ++c) block_sync_lds();
        return;
    }
    f32x4 S[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) S[jb] = zero4();
    if (p.s0) {                                      // S^T[j = 16jb+4g+r][i = 16w+c16] = s0[i][j]: 4 consecutive j per load
        const float* sp = p.s0 + ((size_t)bh * N + 16 * wt + c16) * N + 4 * g;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 x = *reinterpret_cast<const float4*>(sp + 16 * jb);
            S[jb][0] = x.x; S[jb][1] = x.y; S[jb][2] = x.z; S[jb][3] = x.w;
        }
    }
    // after quad_transpose lane (g, c16) owns row 4g + (c16&3) and the 4 consecutive columns 16w + (c16&~3)..+3
    const unsigned out_off = WIDE ? (unsigned)(4 * g + (c16 & 3)) * ts + 16u * wt + (c16 & ~3)
                                  : (unsigned)(4 * g) * ts + 16u * wt + c16;
    float* psa = p.sa ? p.sa + head_base : nullptr;
    uint16_t* py = p.y + head_base;
    float* ps = p.s ? p.s + (size_t)bh * nchunk * N * N : nullptr;
    const unsigned s_off = WIDE ? (unsigned)(4 * g + (c16 & 3)) * N + 16u * wt + (c16 & ~3)   // s[j = 16jb+4g+(c16&3)][i..i+3]
                                : (unsigned)(4 * g) * N + 16u * wt + c16;

    if (tid == 0) lds.prep_done = 0u;
    block_sync_lds();      // prep_done is zeroed
    block_sync_lds();      // B  (producers have filled buffer 0)
    for (int c = 0; c < nchunk; ++c) {
        const BufF& B = lds.b[c & 1];
        WKV_STAMP(0)
        uint2 sh[4], sl[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) split4(S[jb], sh[jb], sl[jb]);
        const bf16x8 bsh[2] = {mk8(sh[0], sh[1]), mk8(sh[2], sh[3])};
        const bf16x8 bsl[2] = {mk8(sl[0], sl[1]), mk8(sl[2], sl[3])};
        // lane (c16, g) needs v[t = 4g+e][i = 16w + c16]: column c16 of the 4 x 16 block at rows 4g.., columns 16w..
        const uint2 vv = TRD ? lds_read_tr16(&B.vn[4 * g + (c16 >> 2)][16 * wt + 4 * (c16 & 3)])
                             : ld8(&B.vt[16 * wt + c16][4 * g]);
        const bf16x8 bvv = mk8(vv, vv);

        // R = M_zk V + Zt S0^T  (three independent accumulator chains)
        f32x4 R = mfma_16x16x32_bf16(mk8(ld8(&B.scb[0][0][c16][4 * g]), ld8(&B.scb[0][1][c16][4 * g])), bvv, zero4());
        f32x4 Ra = zero4(), Rb = zero4();
        {
            const bf16x8 zh = ld_perm(B.opnd[0], c16, 0, g), zl = ld_perm(B.opnd[1], c16, 0, g);
            Ra = mfma_16x16x32_bf16(zh, bsh[0], Ra); Ra = mfma_16x16x32_bf16(zh, bsl[0], Ra); Ra = mfma_16x16x32_bf16(zl, bsh[0], Ra);
        }
        {
            const bf16x8 zh = ld_perm(B.opnd[0], c16, 1, g), zl = ld_perm(B.opnd[1], c16, 1, g);
            Rb = mfma_16x16x32_bf16(zh, bsh[1], Rb); Rb = mfma_16x16x32_bf16(zh, bsl[1], Rb); Rb = mfma_16x16x32_bf16(zl, bsh[1], Rb);
        }
        // Y partials that do not need SA
        f32x4 Y = mfma_16x16x32_bf16(mk8(ld8(&B.scb[1][0][c16][4 * g]), ld8(&B.scb[1][1][c16][4 * g])), bvv, zero4());
        f32x4 Ya = zero4(), Yb = zero4();
        {
            const bf16x8 qh = ld_perm(B.opnd[2], c16, 0, g), ql = ld_perm(B.opnd[3], c16, 0, g);
            Ya = mfma_16x16x32_bf16(qh, bsh[0], Ya); Ya = mfma_16x16x32_bf16(qh, bsl[0], Ya); Ya = mfma_16x16x32_bf16(ql, bsh[0], Ya);
        }
        {
            const bf16x8 qh = ld_perm(B.opnd[2], c16, 1, g), ql = ld_perm(B.opnd[3], c16, 1, g);
            Yb = mfma_16x16x32_bf16(qh, bsh[1], Yb); Yb = mfma_16x16x32_bf16(qh, bsl[1], Yb); Yb = mfma_16x16x32_bf16(ql, bsh[1], Yb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r] += Ra[r] + Rb[r];
        // SA = T R ,  Y += M_qa SA   (f32 matrix core, operands straight from the accumulators)
        const f32x4 SA = mm_f32_image(zero4(), B.scf[1], c16, g, R);
        const f32x4 Yc = mm_f32_image(zero4(), B.scf[0], c16, g, SA);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[r] += (Ya[r] + Yb[r]) + Yc[r];
        WKV_STAMP(1)
        WKV_STAMP(2)
        {
            float* sa_c = psa + (size_t)c * L * ts;
            uint16_t* y_c = py + (size_t)c * L * ts;
            if (WIDE) {
                const f32x4 sat = quad_transpose(SA), yt = quad_transpose(Y);
                if (psa) *reinterpret_cast<float4*>(sa_c + out_off) = make_float4(sat[0], sat[1], sat[2], sat[3]);
                *reinterpret_cast<uint2*>(y_c + out_off) = make_uint2(cvt_pk_bf16(yt[0], yt[1]), cvt_pk_bf16(yt[2], yt[3]));
            } else {
                const uint32_t y01 = cvt_pk_bf16(Y[0], Y[1]), y23 = cvt_pk_bf16(Y[2], Y[3]);   // one v_cvt_pk per pair
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (psa) sa_c[out_off + r * ts] = SA[r];
                    y_c[out_off + r * ts] = (uint16_t)((r < 2 ? y01 : y23) >> (16 * (r & 1)));
                }
            }
        }
        WKV_STAMP(3)
        // S_L^T = diag(c_L) S0^T + [Ab^T | Kb^T] [SA ; V]
        uint2 sah, sal;
        split4(SA, sah, sal);
        const bf16x8 b1 = mk8(sah, vv), b2 = mk8(sal.x, sal.y, 0u, 0u);
        float* s_c = ps + (size_t)c * N * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 cl = *reinterpret_cast<const float4*>(&B.cl[16 * jb + 4 * g]);
            f32x4 acc = S[jb];
            if (!NOAB) { acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w; }
            const int j = 16 * jb + c16;
            bf16x8 ah, al;
            if (TRD) {
                const int tr = 4 * g + (c16 >> 2), tc = 16 * jb + 4 * (c16 & 3);
                ah = mk8(lds_read_tr16(&B.abn[0][tr][tc]), lds_read_tr16(&B.abn[2][tr][tc]));
                al = mk8(lds_read_tr16(&B.abn[1][tr][tc]), lds_read_tr16(&B.abn[3][tr][tc]));
            } else {
                ah = mk8(ld8(&B.trn[0][j][4 * g]), ld8(&B.trn[2][j][4 * g]));
                al = mk8(ld8(&B.trn[1][j][4 * g]), ld8(&B.trn[3][j][4 * g]));
            }
            acc = mfma_16x16x32_bf16(ah, b1, acc);
            acc = mfma_16x16x32_bf16(ah, b2, acc);
            acc = mfma_16x16x32_bf16(al, b1, acc);
            if (NOAB) { acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w; }      // S_L = diag(c_L)(S0 + Ah^T SA + Kh^T V)
            S[jb] = acc;
            if (p.s) {
                if (WIDE) {
                    const f32x4 at = quad_transpose(acc);
                    *reinterpret_cast<float4*>(s_c + s_off + (unsigned)(16 * jb) * N) = make_float4(at[0], at[1], at[2], at[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_c[s_off + (unsigned)(16 * jb + r) * N] = acc[r];
                }
            }
        }
        WKV_STAMP(4)
        block_sync_lds();                                        // B
        WKV_STAMP(5)
    }
    if (p.s_final) {
        float* sp = p.s_final + ((size_t)bh * N + 16 * wt + c16) * N + 4 * g;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) *reinterpret_cast<float4*>(sp + 16 * jb) = make_float4(S[jb][0], S[jb][1], S[jb][2], S[jb][3]);
    }
    WKV_STAMP_FLUSH(0, 0, 6)
}

}  // namespace wkv7c
