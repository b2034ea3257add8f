// WKV7 backward, fourth-generation schedule ("v7"): the three-stage wave pipeline of wkv7_bwd_v6.h with the full-row memory role of
// csrc/wkv7_bwd_rows.h -- gfx950.  EXPERIMENT / A-B PARTNER ONLY: not part of libvisualrwkv_hip.so (builds with
// -DVRWKV_V6_EXPERIMENTS -I benchmarks/experiments select it as backward variant 7; benchmarks/build_alt.sh exp ...).
// It isolates the memory role: P alone 0.815 -> 0.783 ms against v6, the whole kernel 0.975 -> 1.026 ms
// (profiles/r4_wkv7_roles_v6_v7.jsonl) -- which is what led to wkv7_bwd_v8.h.  Same math, same roles (P: images + tail, I: scores /
// T / i-split, J: j-split + score-gradient products) and the same three-chunk pipeline as v6; the J -> P result image is single
// buffered (the tail signals when it has read it), which pays for the staging image: LDS 157 KB.
#pragma once
#include <wkv7_bwd_rows.h>

namespace wkv7v7 {

struct LdsV7 {
    ChunkImg7 b[3];
    uint16_t vdy[4][2][IMG];         // V, dY [t][i] of chunk c in slot c & 3, written by LDS-DMA (swizzled like every image)
    uint16_t stg[5][IMG];            // w q k z a of the chunk the P waves prepare next (LDS-DMA; read by the P waves only)
    float stg_sa[IMG];               // sa of that chunk, fp32, f32_off swizzle
    uint16_t dz[2][IMG];             // "DZ" images of M_zk and T^T (I waves, same step)
    uint16_t sc[2][HLI];             // M_qa, M_qk pair images (I waves, same step)
    uint16_t dsc[4][HLI];            // score gradients of the J waves' chunk (I waves 1-3 -> J waves, same step)
    uint16_t dr[2][2][IMG];          // dR hi, lo [t][i] by chunk parity (I waves -> next step's dM and j-split)
    float s0[2][N * N];              // S0 by chunk parity (P waves' LDS-DMA -> next step's j-split)
    float res[4][IMG];               // J -> P: dZt dQt dAh dKh before the decay factors, fp32 (single: flag 4 hands it back)
    float glast[2][N];               // sum_i dS_L[i][j] S_L[i][j] at the chunk's last token, by chunk parity
    unsigned flag[8];                // 0: M_qa, M_qk, M_zk written (3 per step)  1: dM written (3)  2: T written (1)  3: J operands split (4)
                                     // 4: tail has read `res` (4)  5: P waves hold their staging pieces (4)
};
static_assert(sizeof(LdsV7) <= 160 * 1024, "LDS budget");

// ------------------------------------------------------------------------------------------ kernel
// dbg (PROF): as wkv7_bwd_v6.h.  SKIP (timing experiments only, results are garbage): bit 0 P does nothing, bit 1 I only raises
// its flags, bit 2 J does nothing.
template <bool PROF, int PI = 0, int PJ = 0, int PP = 1, int SKIP = 0>
__global__ __launch_bounds__(768) void bwd_kernel_v7(BwdArgs p) {
    LdsV7& lds = *reinterpret_cast<LdsV7*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int role = wave >> 2, w = wave & 3;           // role 0: I, 1: J, 2: P;  w = index inside the role
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const unsigned bh = blockIdx.x;
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    const float* sbase = p.s + (size_t)bh * nchunk * N * N;
    const int nsteps = nchunk + 3;
    const LaneAddr la = lane_addr(c16, g, w);
    const unsigned out_off = (unsigned)c16 * ts + 16u * w + 4u * g;        // token c16, channels 16w+4g..+3
    WKV_STAMP_DECL
    const unsigned long long rt0_ = PROF ? realtime64_() : 0ull;

    if (tid < 8) lds.flag[tid] = 0u;
    if (role == 2 && !(SKIP & 1)) {                     // rows of the last chunk: staging + its V / dY slot
        const DmaLane dl = dma_lane(lane, ts);
        dma_chunk(lds, p, head_base + (size_t)(nchunk - 1) * L * ts, nchunk - 1, w, ts, dl);
        vmem_drain();
    }
    block_sync_lds();

    if (role == 2) {
        // ================================================================== P: images of chunk cp, tail of chunk cp + 3
        wave_priority<PP>();
        TailRaw q0{}, q1{}, q2{};                           // inputs of chunks cp+1, cp+2, cp+3 at the top of a step
        const unsigned lane_boff = out_off * 2u;
        const DmaLane dl = dma_lane(lane, ts);
        unsigned n_ps = 0;
        // One step.  FULL (steps 3 .. nchunk-2: a tail, a prep, a non-empty S0 and a next chunk every time) has no conditions:
        // every path issues [4-5 row DMAs, 4 S0 DMAs, 5 tail stores] in this order, so the wait before the barrier is
        // vmcnt(5): everything the other roles will read has landed, the stores stay in flight.
        auto pstep = [&](int n, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int cp = nchunk - 1 - n, cd = cp + 1, ct = cp + 3;      // images | S0 by DMA (the I waves' chunk) | tail
            WKV_STAMP(4)
            if (!(SKIP & 1)) {
                RawP raw;
                const bool do_prep = FULL || cp >= 0;
                if (do_prep) {
                    raw = read_stage(lds, la);
                    lds_flag_add(&lds.flag[5]);             // (waits for the reads) ...
                    n_ps += 4u;
                    lds_flag_wait(&lds.flag[5], n_ps);      // ... all four P waves hold their pieces: the staging bytes are free
                }
                if (FULL || cp >= 1) dma_chunk(lds, p, head_base + (size_t)(cp - 1) * L * ts, cp - 1, w, ts, dl);
                // S0 of chunk cd = s[cd-1] for the j-split of the next step; its buffer was last read two steps ago
                if (FULL) dma_state(lds.s0[cd & 1], sbase + (size_t)(cd - 1) * N * N, 4 * w, 4 * w + 4, lane);
                else if (cd >= 0 && cd <= nchunk - 1) dma_state(lds.s0[cd & 1], cd > 0 ? sbase + (size_t)(cd - 1) * N * N : nullptr, 4 * w, 4 * w + 4, lane);
#if VRWKV_PDELAY
                // the tail (VALU only) waits until the J waves have split their operands (VALU only as well): it then runs beside
                // their matrix-core phase instead; J is active in steps 2 .. nchunk + 1 and counts 4 per step
                if (!(SKIP & 4) && (FULL || (n >= 2 && n <= nchunk + 1))) lds_flag_wait(&lds.flag[3], 4u * (unsigned)(n - 1));
#endif
                if (FULL || (ct >= 0 && ct <= nchunk - 1)) tail7(lds, ct & 1, q2, p, head_base + (size_t)ct * L * ts, lane_boff, c16, w, g, la);
                WKV_STAMP(0)
                q2 = q1; q1 = q0;
                if (do_prep) {
                    const Decay dd = prep7(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la);
                    q0.q = raw.q; q0.k = raw.k; q0.z = raw.z; q0.a = raw.a;
                    q0.x2[0] = dd.x2[0]; q0.x2[1] = dd.x2[1]; q0.x2[2] = dd.x2[2]; q0.x2[3] = dd.x2[3];
                }
                WKV_STAMP(1)
                if (FULL) vmem_wait<5>(); else vmem_drain();
            }
            WKV_STAMP(2)
            block_sync_lds();
            WKV_STAMP(3)
        };
        int n = 0;
        for (; n < 3 && n < nsteps; ++n) pstep(n, BoolTag<false>{});
        for (; n < nchunk - 1; ++n) pstep(n, BoolTag<true>{});
        for (; n < nsteps; ++n) pstep(n, BoolTag<false>{});
        WKV_STAMP_FLUSH(512, 10, 5)
        return;
    }

    if (role == 0) {
        // ================================================================== I: chunk ci = nchunk - n  (steps 1 .. nchunk)
        wave_priority<PI>();
        f32x4 dS1[4];                                       // dS1[jb][r] = dS[i = 16w+c16][j = tix(jb, 4g+r)]
#pragma unroll
        for (int x = 0; x < 4; ++x) dS1[x] = zero4();
        unsigned n_sc = 0, n_t = 0;
        for (int n = 0; n < nsteps; ++n) {
            const int ci = nchunk - n, cj = ci + 1;         // this role's chunk | the J waves' chunk of this step
            WKV_STAMP(4)
            if (ci >= 0 && ci <= nchunk - 1) {
                const ChunkImg7& B = lds.b[ci % 3];
                if (!(SKIP & 2)) wkv7v6::scores6<true>(lds, B, w, c16, g, la);
                lds_flag_add(&lds.flag[w == 0 ? 2 : 0]);        // T has its own counter: nobody waits for it before dSA is done
                n_sc += 3; n_t += 1;
            }
            if (w > 0 && cj >= 0 && cj <= nchunk - 1) {     // score gradients of the J waves' chunk (their dR is one step old)
                const ChunkImg7& Bj = lds.b[cj % 3];
                const uint16_t* drh = lds.dr[cj & 1][0];
                const uint16_t* drl = lds.dr[cj & 1][1];
                const uint16_t* vi = lds.vdy[cj & 3][0];
                const uint16_t* dyi = lds.vdy[cj & 3][1];
                if (!(SKIP & 2)) {
                if (w == 1) { dscores7(lds, Bj.sa[0], Bj.sa[1], vi, dyi, drh, drl, 0, c16, g, la); dscores7(lds, Bj.sa[0], Bj.sa[1], vi, dyi, drh, drl, 3, c16, g, la); }
                else dscores7(lds, Bj.sa[0], Bj.sa[1], vi, dyi, drh, drl, w - 1, c16, g, la);
                }
                lds_flag_add(&lds.flag[1]);
            }
            WKV_STAMP(0)
            if (!(SKIP & 2) && ci >= 0 && ci <= nchunk - 1) {
                const ChunkImg7& B = lds.b[ci % 3];
                const uint16_t* dyi = lds.vdy[ci & 3][1];
                const size_t cbase = head_base + (size_t)ci * L * ts;
                lds_flag_wait(&lds.flag[0], n_sc);
                WKV_STAMP(1)
                // ------------------------------------------------------------ i-split (i = 16w + c16)
                f32x4 dSc[4];                               // diag(c_L) dS^T: operand of dSA / dV and start of the update
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
                    dSc[jb] = dS1[jb];
                    dSc[jb][0] *= cl.x; dSc[jb][1] *= cl.y; dSc[jb][2] *= cl.z; dSc[jb][3] *= cl.w;
                }
                bf16x8 sh[2], sl[2];
                tiles_op(dSc, sh, sl);
                const uint2 dyv = lds_read_tr16(&dyi[la.trc]);                   // dY[4g+e][i]
                const bf16x8 dyd = mk8(dyv, dyv);
                // dSA[t][i] = sum_s M_qa[s][t] dY[s][i] + sum_j Ah[t][j] c_L[j] dS[i][j]
                f32x4 dSA = mfma32(ld16(&lds.sc[0][la.hl]), dyd, zero4());
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 ah = ld16(&B.opnd[4][la.row[kb]]);
                    dSA = mfma32(ah, sh[kb], dSA);
                    dSA = mfma32(ah, sl[kb], dSA);
                    dSA = mfma32(ld16(&B.opnd[5][la.row[kb]]), sh[kb], dSA);
                }
                uint2 xh, xl, rh, rl;
                split4(dSA, xh, xl);
                const bf16x8 xhl = mk8(xh, xl);
                // dR = T^T dSA in both orientations: [t][i] stays in registers, [i][t] (token per lane) goes to LDS
                lds_flag_wait(&lds.flag[2], n_t);               // T of this chunk (wave 0's doubling chain) is in LDS
                const bf16x8 t1 = ld16(&lds.dz[1][la.row[0]]), t2 = ld16(&lds.dz[1][la.row[1]]);        // [T_h T_h], [T_l 0]
                f32x4 dR = mfma32(t1, xhl, zero4());
                dR = mfma32(t2, xhl, dR);
                f32x4 dRT = mfma32(xhl, t1, zero4());
                dRT = mfma32(xhl, t2, dRT);
                split4(dR, rh, rl);
                {
                    uint2 th, tl;
                    split4(dRT, th, tl);
                    st8(&lds.dr[ci & 1][0][la.own], th);
                    st8(&lds.dr[ci & 1][1][la.own], tl);
                }
                WKV_STAMP(5)
                // dV^T[i][t] = sum_j c_L[j] dS[i][j] Kh[t][j] + sum_s dY[s][i] M_qk[s][t] + sum_s dR[s][i] M_zk[s][t]
                {
                    f32x4 dV = mfma32(dyd, ld16(&lds.sc[1][la.hl]), zero4());
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 kh = ld16(&B.opnd[6][la.row[kb]]);
                        dV = mfma32(sh[kb], kh, dV);
                        dV = mfma32(sl[kb], kh, dV);
                        dV = mfma32(sh[kb], ld16(&B.opnd[7][la.row[kb]]), dV);
                    }
                    const bf16x8 rhl = mk8(rh, rl);
                    dV = mfma32(rhl, ld16(&lds.dz[0][la.row[0]]), dV);                 // [M_zk_h M_zk_h]
                    dV = mfma32(rhl, ld16(&lds.dz[0][la.row[1]]), dV);                 // [M_zk_l 0]
                    *reinterpret_cast<uint2*>(p.dv + cbase + out_off) = make_uint2(cvt_pk_bf16(dV[0], dV[1]), cvt_pk_bf16(dV[2], dV[3]));
                }
                WKV_STAMP(6)
                // dS^T <- diag(c_L) dS^T + [Qt^T | Zt^T] [dY ; dR]
                const bf16x8 y1 = mk8(dyv, rh), y2 = mk8(0u, 0u, rl.x, rl.y);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    f32x4 acc = dSc[jb];
                    const int o = la.tri[jb >> 1] + 4 * (jb & 1);
                    const bf16x8 xh8 = mk8(lds_read_tr16(&B.opnd[2][o]), lds_read_tr16(&B.opnd[0][o]));
                    const bf16x8 xl8 = mk8(lds_read_tr16(&B.opnd[3][o]), lds_read_tr16(&B.opnd[1][o]));
                    acc = mfma32(xh8, y1, acc);
                    acc = mfma32(xl8, y1, acc);
                    acc = mfma32(xh8, y2, acc);
                    dS1[jb] = acc;
                }
            }
            WKV_STAMP(2)
            block_sync_lds();
            WKV_STAMP(3)
        }
        WKV_STAMP_FLUSH(0, 0, 5)
        if (PROF && blockIdx.x == 0 && tid == 0) { p.dbg[15] = realtime64_() - rt0_; p.dbg[18] = tacc_[5]; p.dbg[19] = tacc_[6]; }   // i-split: dSA + dR | dV
        return;
    }

    // ====================================================================== J: chunk cj = nchunk + 1 - n  (steps 2 .. nchunk + 1)
    wave_priority<PJ>();
    const int j = 16 * w + c16;                         // key column of the j-split tiles
    f32x4 dS2[4];                                       // dS2[ib][r] = dS[i = tix(ib, 4g+r)][j]
#pragma unroll
    for (int x = 0; x < 4; ++x) dS2[x] = zero4();
    float gl_carry = 0.f;                               // sum_i dS[i][j] S_L[i][j] of the chunk about to be processed
    unsigned n_dm = 0;
    for (int n = 0; n < nsteps; ++n) {
        const int cj = nchunk + 1 - n;
        WKV_STAMP(4)
        if (!(SKIP & 4) && cj >= 0 && cj <= nchunk - 1) {
            const ChunkImg7& B = lds.b[cj % 3];
            const uint16_t* drh = lds.dr[cj & 1][0];
            const uint16_t* drl = lds.dr[cj & 1][1];
            const uint16_t* vi = lds.vdy[cj & 3][0];
            const uint16_t* dyi = lds.vdy[cj & 3][1];
            const float* s0img = lds.s0[cj & 1];
            n_dm += 3;
            // ---------------------------------------------------------------- j-split (j = 16w + c16)
            f32x4 dZt, dQt, dAh, dKh;
            bf16x8 qzh, qzl;
            {
                const float clj = B.cl[j];
                f32x4 S0[4], dU[4];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    // [ib][r] = S0[i = tix(ib, 4g+r)][j]  <-  image row j (zeros for the first chunk of the sequence)
                    const float4 x = *reinterpret_cast<const float4*>(&s0img[f32_off(j, tix(ib, 4 * g))]);
                    S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
                    dU[ib] = dS2[ib];
                    dU[ib][0] *= clj; dU[ib][1] *= clj; dU[ib][2] *= clj; dU[ib][3] *= clj;
                }
                bf16x8 s0h[2], s0l[2], duh[2], dul[2];
                tiles_op(S0, s0h, s0l);
                tiles_op(dU, duh, dul);
#if VRWKV_PDELAY
                lds_flag_add(&lds.flag[3]);
#endif
                WKV_STAMP(5)
                // transposed results: D[m = j][n = t]  (lane = token, registers = 4 consecutive channels of the wave's 16)
                {
                    const bf16x8 drr = ld16(&drh[la.row[0]]);
                    dZt = mfma32(s0h[0], drr, zero4());                                  // dR S0
                    dZt = mfma32(s0l[0], drr, dZt);
                    dZt = mfma32(s0h[0], ld16(&drl[la.row[0]]), dZt);
                    const bf16x8 dyr = ld16(&dyi[la.row[0]]);
                    dQt = mfma32(s0h[0], dyr, zero4());                                  // dY S0
                    dQt = mfma32(s0l[0], dyr, dQt);
                    const bf16x8 sah = ld16(&B.sa[0][la.row[0]]);
                    dAh = mfma32(duh[0], sah, zero4());                                  // SA dU
                    dAh = mfma32(dul[0], sah, dAh);
                    dAh = mfma32(duh[0], ld16(&B.sa[1][la.row[0]]), dAh);
                    const bf16x8 vr = ld16(&vi[la.row[0]]);
                    dKh = mfma32(duh[0], vr, zero4());                                   // V dU
                    dKh = mfma32(dul[0], vr, dKh);
                }
                {
                    const bf16x8 drr = ld16(&drh[la.row[1]]);
                    dZt = mfma32(s0h[1], drr, dZt);
                    dZt = mfma32(s0l[1], drr, dZt);
                    dZt = mfma32(s0h[1], ld16(&drl[la.row[1]]), dZt);
                    const bf16x8 dyr = ld16(&dyi[la.row[1]]);
                    dQt = mfma32(s0h[1], dyr, dQt);
                    dQt = mfma32(s0l[1], dyr, dQt);
                    const bf16x8 sah = ld16(&B.sa[0][la.row[1]]);
                    dAh = mfma32(duh[1], sah, dAh);
                    dAh = mfma32(dul[1], sah, dAh);
                    dAh = mfma32(duh[1], ld16(&B.sa[1][la.row[1]]), dAh);
                    const bf16x8 vr = ld16(&vi[la.row[1]]);
                    dKh = mfma32(duh[1], vr, dKh);
                    dKh = mfma32(dul[1], vr, dKh);
                }
                WKV_STAMP(6)
                // dS <- dU + [dY^T | dR^T] [Qt ; Zt]
                qzh = mk8(lds_read_tr16(&B.opnd[2][la.trc]), lds_read_tr16(&B.opnd[0][la.trc]));
                qzl = mk8(lds_read_tr16(&B.opnd[3][la.trc]), lds_read_tr16(&B.opnd[1][la.trc]));
                const u32x4v qz = __builtin_bit_cast(u32x4v, qzh);
                const bf16x8 zpad = mk8(qz[2], qz[3], 0u, 0u);                           // [Zt_h ; 0]
                float gl = 0.f;
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const int o = la.tri[ib >> 1] + 4 * (ib & 1);
                    const bf16x8 x8 = mk8(lds_read_tr16(&dyi[o]), lds_read_tr16(&drh[o]));
                    // [dR_l^T | finite filler]: the filler meets the zero half of zpad (another tile's dR_l: finite, not reused)
                    const bf16x8 xl8 = mk8(lds_read_tr16(&drl[o]), lds_read_tr16(&drl[o ^ 4]));
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
                if (g == 0) lds.glast[cj & 1][j] = gl_carry;          // the term of THIS chunk was formed a step ago
                gl_carry = gl;
            }
            WKV_STAMP(0)
            lds_flag_wait(&lds.flag[1], n_dm);
            WKV_STAMP(1)
            // ---------------------------------------------------------------- dM products
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
            // results: lane = token c16, registers = channels 16w + 4g + e -> the fp32 image of the P waves' tail, once the tail of
            // the chunk before (this step's, steps 3 ..) has read it: 4 P waves per tail
            if (!(SKIP & 1) && n >= 3) lds_flag_wait(&lds.flag[4], 4u * (unsigned)(n - 2));
            *reinterpret_cast<float4*>(&lds.res[0][la.f32]) = make_float4(dZt[0], dZt[1], dZt[2], dZt[3]);
            *reinterpret_cast<float4*>(&lds.res[1][la.f32]) = make_float4(dQt[0], dQt[1], dQt[2], dQt[3]);
            *reinterpret_cast<float4*>(&lds.res[2][la.f32]) = make_float4(dAh[0], dAh[1], dAh[2], dAh[3]);
            *reinterpret_cast<float4*>(&lds.res[3][la.f32]) = make_float4(dKh[0], dKh[1], dKh[2], dKh[3]);
        }
        WKV_STAMP(2)
        block_sync_lds();
        WKV_STAMP(3)
    }
    WKV_STAMP_FLUSH(256, 5, 5)
    if (PROF && blockIdx.x == 0 && tid == 256) { p.dbg[16] = tacc_[5]; p.dbg[17] = tacc_[6]; }   // j-split: operand split | output products
}

}  // namespace wkv7v7
