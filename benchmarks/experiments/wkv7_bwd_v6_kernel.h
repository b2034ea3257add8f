// EXPERIMENT / A-B PARTNER, not part of the product library: the round-3 backward kernel (three-stage wave pipeline, 8-byte register loads).
// Built only by benchmarks/build_alt.sh (through wkv7_experiments.h: variant 6 and the role-skip builds 61 .. 67) and by the host emulator's tests.
// The shared building blocks it uses stayed in csrc/wkv7_bwd_v6.h.  Its original header comment:
//
// WKV7 backward, chunked MFMA form, third-generation schedule: a three-stage wave pipeline -- gfx950.
//
// Same math and the same operand images as wkv7_bwd_v5.h (closed-form differentiation of a 16-token chunk from S0 = s[c-1]
// and the saved sa; reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130).  What changed is WHO does what and WHEN.
// The v5 counters (profiles/r2_wkv7_pmc_b16.txt) say the kernel is bound by dependent latency, not by any pipe: per wave
// and chunk 37 % of the cycles issue an instruction, 27 % stall on issue (MFMA results, LDS queue) and 36 % are parked at
// s_waitcnt / s_barrier; VALU and matrix pipes are each ~25 % busy.  With 149 KB of LDS only one workgroup fits a CU, so a
// SIMD holds exactly one consumer and one producer wave, and the consumers walk their three segments (i-split, j-split,
// score-gradient products + element-wise tail) strictly one after the other.  The only chain that really is sequential
// from chunk to chunk is dS -> dSA -> dR -> dS (the i-split); everything else hangs off it.
//
// Here a workgroup is 12 waves in three roles of four, each SIMD holding one wave of every role, and the roles work on
// THREE consecutive chunks at the same time (step n, chunks counted down from the end of the sequence):
//   P  (waves 8-11)  operand images of chunk c-2 (decay scan, scaling, hi/lo split), S0 of chunk c-1 by LDS-DMA, prefetch of
//                    chunk c-3 -- and the element-wise tail + the five gradient stores of chunk c+1, whose inputs w,q,k,z,a
//                    it still holds in registers (a four-deep register queue; no `raw` / `dec` images in LDS);
//   I  (waves 0-3)   scores and T = (I - M_za)^-1 of chunk c-1, then its i-split: dSA, dR, dV, the dS update.  While wave 0
//                    runs the T chain, waves 1-3 also form the score gradients dM of chunk c for the J waves;
//   J  (waves 4-7)   j-split of chunk c (products against S0 and dU, second copy of dS) and the dM products; results go to
//                    LDS as fp32 for the P waves' tail.
// One workgroup barrier per step; inside a step two LDS counters (scores ready: I -> I, dM ready: I -> J).
// Per-chunk images live for three steps (three buffers of 24 KB); everything else is single or double buffered by the parity
// of the chunk that owns it.  The Ab / Kb images of v5 are gone: Ab dS^T = Ah (diag(c_L) dS)^T, and the dS update starts
// from diag(c_L) dS anyway.  LDS 155 KB.
#pragma once
#include <wkv7_bwd_v6.h>

namespace wkv7v6 {

struct ChunkImg {                    // per chunk; three alive: P builds c-2, I reads c-1, J reads c
    uint16_t opnd[8][IMG];           // Zt_h Zt_l Qt_h Qt_l Ah_h Ah_l Kh_h Kh_l      [t][j]
    uint16_t ti[4][IMG];             // V  dY  SA_h  SA_l                            [t][i]
    float cl[N];                     // c_L[j]
};
struct ResImg {                      // J -> P: the four [t][j] results of a chunk before the decay factors, fp32
    float r[4][IMG];                 // dZt dQt dAh dKh   (f32_off swizzle)
    float glast[N];                  // sum_i dS_L[i][j] S_L[i][j] at the chunk's last token
};
struct LdsV6 {
    ChunkImg b[3];
    uint16_t dz[2][IMG];             // "DZ" images of M_zk and T^T (I waves, same step)
    uint16_t sc[2][HLI];             // M_qa, M_qk pair images (I waves, same step)
    uint16_t dsc[4][HLI];            // score gradients of the J waves' chunk (I waves 1-3 -> J waves, same step)
    uint16_t dr[2][2][IMG];          // dR hi, lo [t][i] by chunk parity (I waves -> next step's dM and j-split)
    float s0[2][N * N];              // S0 by chunk parity (P waves' LDS-DMA -> next step's j-split)
    ResImg res[2];                   // by chunk parity (J waves -> next step's tail)
    unsigned flag[4];                // 0: M_qa, M_qk, M_zk of this step written (3 per step)   1: dM written (3 per step)   2: T written (1)
};
static_assert(sizeof(LdsV6) <= 160 * 1024, "LDS budget");

// ------------------------------------------------------------------------------------------ P: operand images + tail
using RawIn = wkv7v5::RawB;        // w q k z a v dy (uint2) + sa (float4): one lane's 4 channels of one token

// The compiler waits for a load at the first USE of its destination register and counts vmcnt in issue order; it cannot see
// the LDS-DMA (inline asm) in between, so a counted wait placed after the DMA also waits for the DMA.  pin() is an empty asm
// that "uses" the prefetched registers right after the step's own vmem_drain(): the compiler's wait lands there, where
// everything has retired anyway, and the next step starts with no pending loads in its model.
DEVFN void pin(RawIn& r) {
    asm volatile("" : "+v"(r.w.x), "+v"(r.w.y), "+v"(r.q.x), "+v"(r.q.y), "+v"(r.k.x), "+v"(r.k.y), "+v"(r.z.x), "+v"(r.z.y));
    asm volatile("" : "+v"(r.a.x), "+v"(r.a.y), "+v"(r.v.x), "+v"(r.v.y), "+v"(r.dy.x), "+v"(r.dy.y));
    asm volatile("" : "+v"(r.sa.x), "+v"(r.sa.y), "+v"(r.sa.z), "+v"(r.sa.w));
}


DEVFN Decay prep(ChunkImg& B, const RawIn& raw, int c16, int j0, const LaneAddr& la) {
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
    st8(&B.ti[0][la.own], raw.v);
    st8(&B.ti[1][la.own], raw.dy);
    const float sav[4] = {raw.sa.x, raw.sa.y, raw.sa.z, raw.sa.w};
    split4(sav, hh, ll); st8(&B.ti[2][la.own], hh); st8(&B.ti[3][la.own], ll);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
    return d;
}

// element-wise tail of one chunk: lane = token c16, channels 16 pw + 4g + e (the lane's own prep columns)
DEVFN void tail(const ResImg& R, const TailRaw& tr, const BwdArgs& p, size_t u, unsigned lane_boff, int c16, int pw, int g, const LaneAddr& la) {
    const float4 zt4 = *reinterpret_cast<const float4*>(&R.r[0][la.f32]);
    const float4 qt4 = *reinterpret_cast<const float4*>(&R.r[1][la.f32]);
    const float4 ah4 = *reinterpret_cast<const float4*>(&R.r[2][la.f32]);
    const float4 kh4 = *reinterpret_cast<const float4*>(&R.r[3][la.f32]);
    const float4 gl4 = *reinterpret_cast<const float4*>(&R.glast[16 * pw + 4 * g]);
    const float dZt[4] = {zt4.x, zt4.y, zt4.z, zt4.w}, dQt[4] = {qt4.x, qt4.y, qt4.z, qt4.w};
    const float dAh[4] = {ah4.x, ah4.y, ah4.z, ah4.w}, dKh[4] = {kh4.x, kh4.y, kh4.z, kh4.w};
    const float glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
    float q[4], k[4], z[4], a[4];
    unpack4(tr.q, q); unpack4(tr.k, k); unpack4(tr.z, z); unpack4(tr.a, a);
    Decay d;                                              // log2 c_t from the queue; log2 w_t = its difference along t
#pragma unroll
    for (int e = 0; e < 4; ++e) { d.x2[e] = tr.x2[e]; d.l2[e] = tr.x2[e] - dpp_shr1_fill(tr.x2[e], 0.f); }
    float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float cc = fast_exp2(d.x2[e]), ic = fast_exp2(-d.x2[e]), cp = dpp_shr1_fill(cc, 1.f);
        dz[e] = dZt[e] * cp; dq[e] = dQt[e] * cc; da[e] = dAh[e] * ic; dk[e] = dKh[e] * ic;
        // decay-gradient integrand g_t = dq q - da a - dk k + (dz z)[t+1]  (+ sum_i dS.S_L at the last token)
        float gt = dq[e] * q[e] - da[e] * a[e] - dk[e] * k[e] + dpp_shl<1>(dz[e] * z[e]);
        if (c16 == 15) gt += glv[e];
        gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
        dw[e] = gt * (d.l2[e] * LN2);
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
DEVFN void dscores6(LdsV6& lds, const ChunkImg& B, const uint16_t* drh, const uint16_t* drl, int piece, int c16, int g, const LaneAddr& la) {
    uint2 hh, ll;
    if (piece == 0) mask_split<false, false>(dot64<true, true>(B.ti[2], B.ti[3], drh, drl, la), c16, g, hh, ll);              // tril_(dR SA^T)
    else if (piece == 1) mask_split<false, false>(dot64<false, true>(B.ti[0], B.ti[0], drh, drl, la), c16, g, hh, ll);       // tril_(dR V^T)
    else if (piece == 2) mask_split<true, false>(dot64<true, false>(B.ti[2], B.ti[3], B.ti[1], B.ti[1], la), c16, g, hh, ll);  // tril(dY SA^T)
    else mask_split<true, false>(dot64<false, false>(B.ti[0], B.ti[0], B.ti[1], B.ti[1], la), c16, g, hh, ll);                // tril(dY V^T)
    const int o = la.hl + 4 * (piece & 1);                   // za / qa first, zk / qk second of the pair
    st8(&lds.dsc[piece & 2][o], hh);
    st8(&lds.dsc[(piece & 2) + 1][o], ll);
}

// ------------------------------------------------------------------------------------------ kernel
// dbg (PROF): [0..4] I wave 0: scores | flag wait | i-split | barrier | -   [5..9] J wave 0: j-split | dM wait | products |
// barrier | -   [10..14] P wave 0: tail | prep | drain | barrier | -   [15] life of workgroup 0 on the 100 MHz counter
// PI / PJ / PP: static wave priority of the three roles; SWAP: the J role on the oldest waves (0-3), I on 4-7 (VALU issue is
// arbitrated by priority, then age: MI355X_MICROARCH.md "Two waves per SIMD").
// SKIP (timing experiments only, results are garbage): bit 0 P does nothing, bit 1 I only raises its flags, bit 2 J does nothing.
template <bool PROF, int PI = 0, int PJ = 0, int PP = 1, bool SWAP = false, bool TBF16 = true, int SKIP = 0>
__global__ __launch_bounds__(768) void bwd_kernel_v6(BwdArgs p) {
    LdsV6& lds = *reinterpret_cast<LdsV6*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int slot = wave >> 2, w = wave & 3;            // w = index inside the role
    const int role = SWAP && slot < 2 ? 1 - slot : slot; // role 0: I, 1: J, 2: P
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

    if (tid < 4) lds.flag[tid] = 0u;
    block_sync_lds();

    if (role == 2) {
        // ================================================================== P: images of chunk cp, tail of chunk cp + 3
        wave_priority<PP>();
        RawIn raw;
        TailRaw q0{}, q1{}, q2{};                           // inputs of chunks cp+1, cp+2, cp+3 at the top of a step
        // wave-uniform 64-bit base (SGPRs) + one 32-bit lane offset: the loads / stores take the scalar-base form and need no
        // per-array 64-bit address arithmetic on the VALU (25 v_lshl_add_u64 per step before)
        const unsigned lane_boff = out_off * 2u, lane_foff = out_off * 4u;
        auto at16 = [&](const uint16_t* base, size_t uoff) { return reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base + uoff) + lane_boff); };
        auto fetch = [&](RawIn& r, int c) {
            const size_t u = head_base + (size_t)c * L * ts;              // uniform
            r.w = *at16(p.w, u); r.q = *at16(p.q, u); r.k = *at16(p.k, u); r.z = *at16(p.z, u);
            r.a = *at16(p.a, u); r.v = *at16(p.v, u); r.dy = *at16(p.dy, u);
            r.sa = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.sa + u) + lane_foff);
        };
        fetch(raw, nchunk - 1);
        vmem_drain();
        pin(raw);
        // One step.  FULL (steps 3 .. nchunk-1: a tail, a prep and a non-empty S0 every time) has no conditions, so the
        // compiler sees the same issue order on every path -- LDS-DMA, 8 prefetch loads, 5 tail stores -- and its wait for the
        // prefetch at pin() is vmcnt(5): the DMA (older) has landed, the stores (younger) stay in flight.  A drain to
        // vmcnt(0) there waited out the HBM write round trip in every step (the P role alone then takes 5.1k cycles per
        // step, more than either of the other two).  The ragged first / last steps keep the full drain.
        auto pstep = [&](int n, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int cp = nchunk - 1 - n, cd = cp + 1, ct = cp + 3;      // images | S0 by DMA (the I waves' chunk) | tail
            WKV_STAMP(4)
            if (!(SKIP & 1)) {
                // S0 of chunk cd = s[cd-1] for the j-split of the next step; its buffer was last read two steps ago
                if (FULL) dma_state(lds.s0[cd & 1], sbase + (size_t)(cd - 1) * N * N, 4 * w, 4 * w + 4, lane);
                else if (cd >= 0 && cd <= nchunk - 1) dma_state(lds.s0[cd & 1], cd > 0 ? sbase + (size_t)(cd - 1) * N * N : nullptr, 4 * w, 4 * w + 4, lane);
                // prefetch of the next chunk first, a full step ahead of its use (issued after the prep instead, the P role alone
                // ran 0.77 -> 0.97 ms: when every CU streams, the loads need most of a step to come back)
                RawIn nxt;
                fetch(nxt, cp - 1 > 0 ? cp - 1 : 0);        // unconditional (chunk 0 again past the end): no copies, no early wait
#if VRWKV_PDELAY
                // the tail (VALU only) waits until the J waves have split their operands (VALU only as well): it then runs beside
                // their matrix-core phase instead; J is active in steps 2 .. nchunk + 1 and counts 4 per step
                if (!(SKIP & 4) && (FULL || (n >= 2 && n <= nchunk + 1))) lds_flag_wait(&lds.flag[3], 4u * (unsigned)(n - 1));
#endif
                if (FULL || (ct >= 0 && ct <= nchunk - 1)) tail(lds.res[ct & 1], q2, p, head_base + (size_t)ct * L * ts, lane_boff, c16, w, g, la);
                WKV_STAMP(0)
                q2 = q1; q1 = q0;
                if (FULL || cp >= 0) {
                    const Decay dd = prep(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la);
                    q0.q = raw.q; q0.k = raw.k; q0.z = raw.z; q0.a = raw.a;
                    q0.x2[0] = dd.x2[0]; q0.x2[1] = dd.x2[1]; q0.x2[2] = dd.x2[2]; q0.x2[3] = dd.x2[3];
                }
                WKV_STAMP(1)
                if (!FULL) vmem_drain();
                pin(nxt);
                raw = nxt;
            }
            WKV_STAMP(2)
            block_sync_lds();
            WKV_STAMP(3)
        };
        int n = 0;
        for (; n < 3 && n < nsteps; ++n) pstep(n, BoolTag<false>{});
        // unrolled by 6 = lcm(queue depth 3, prefetch ping-pong 2): the queue shifts and the raw <- nxt copy become renaming
        // (120 of the ~400 VALU instructions of a step were v_mov)
        // (unrolling this loop so that the queue shifts and `raw = nxt` become renaming was tried by 2, 3, 4 and 6: the P role's
        // live set is at the 168-register limit of a 12-wave workgroup and every variant spilled inside the loop: 0.99 -> 1.28 ms)
        for (; n < nchunk; ++n) pstep(n, BoolTag<true>{});
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
                const ChunkImg& B = lds.b[ci % 3];
                if (!(SKIP & 2)) scores6<TBF16>(lds, B, w, c16, g, la);
                lds_flag_add(&lds.flag[w == 0 ? 2 : 0]);        // T has its own counter: nobody waits for it before dSA is done
                n_sc += 3; n_t += 1;
            }
            if (w > 0 && cj >= 0 && cj <= nchunk - 1) {     // score gradients of the J waves' chunk (their dR is one step old)
                const ChunkImg& Bj = lds.b[cj % 3];
                const uint16_t* drh = lds.dr[cj & 1][0];
                const uint16_t* drl = lds.dr[cj & 1][1];
                if (!(SKIP & 2)) {
                if (w == 1) { dscores6(lds, Bj, drh, drl, 0, c16, g, la); dscores6(lds, Bj, drh, drl, 3, c16, g, la); }
                else dscores6(lds, Bj, drh, drl, w - 1, c16, g, la);
                }
                lds_flag_add(&lds.flag[1]);
            }
            WKV_STAMP(0)
            if (!(SKIP & 2) && ci >= 0 && ci <= nchunk - 1) {
                const ChunkImg& B = lds.b[ci % 3];
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
                const uint2 dyv = lds_read_tr16(&B.ti[1][la.trc]);               // dY[4g+e][i]
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
        WKV_STAMP_FLUSH(SWAP ? 256 : 0, 0, 5)
        if (PROF && blockIdx.x == 0 && tid == (SWAP ? 256 : 0)) { p.dbg[15] = realtime64_() - rt0_; p.dbg[18] = tacc_[5]; p.dbg[19] = tacc_[6]; }   // i-split: dSA + dR | dV
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
            const ChunkImg& B = lds.b[cj % 3];
            const uint16_t* drh = lds.dr[cj & 1][0];
            const uint16_t* drl = lds.dr[cj & 1][1];
            const float* s0img = lds.s0[cj & 1];
            ResImg& R = lds.res[cj & 1];
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
                    const bf16x8 drr = ld16(&drh[la.row[1]]);
                    dZt = mfma32(s0h[1], drr, dZt);
                    dZt = mfma32(s0l[1], drr, dZt);
                    dZt = mfma32(s0h[1], ld16(&drl[la.row[1]]), dZt);
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
                    const bf16x8 x8 = mk8(lds_read_tr16(&B.ti[1][o]), lds_read_tr16(&drh[o]));
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
                if (g == 0) R.glast[j] = gl_carry;          // the term of THIS chunk was formed a step ago
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
            // results: lane = token c16, registers = channels 16w + 4g + e -> fp32 images for the P waves' tail
            *reinterpret_cast<float4*>(&R.r[0][la.f32]) = make_float4(dZt[0], dZt[1], dZt[2], dZt[3]);
            *reinterpret_cast<float4*>(&R.r[1][la.f32]) = make_float4(dQt[0], dQt[1], dQt[2], dQt[3]);
            *reinterpret_cast<float4*>(&R.r[2][la.f32]) = make_float4(dAh[0], dAh[1], dAh[2], dAh[3]);
            *reinterpret_cast<float4*>(&R.r[3][la.f32]) = make_float4(dKh[0], dKh[1], dKh[2], dKh[3]);
        }
        WKV_STAMP(2)
        block_sync_lds();
        WKV_STAMP(3)
    }
    WKV_STAMP_FLUSH(SWAP ? 0 : 256, 5, 5)
    if (PROF && blockIdx.x == 0 && tid == (SWAP ? 0 : 256)) { p.dbg[16] = tacc_[5]; p.dbg[17] = tacc_[6]; }   // j-split: operand split | output products
}


}  // namespace wkv7v6
