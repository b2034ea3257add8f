// WKV7 backward, chunked MFMA form, producer/consumer wave specialisation -- gfx950.
//
// Same algorithm as wkv7_chunked_bwd.h (see its header for the math); the schedule is rebuilt from the phase
// times measured on MI355X for the 4-wave kernel (per chunk, B=8: prep 3.4k cycles, scores 1.9k, i-split 2.4k,
// score gradients 0.7k, j-split 2.5k, tail 1.3k, all back to back on one wave per SIMD).
// One workgroup = 8 waves per (b,h); chunks are walked from last to first; three hand-off points per chunk.  They are
// counters in LDS (lds_flag_add / lds_flag_wait), not s_barrier: a wave waits only for the waves whose data it needs,
// e.g. the consumers never wait for the producers' first segment (measured with workgroup barriers: consumers idle
// 0.8k of 9k cycles per chunk at X, producers 1.6k at Y and Z).
//
//              segment 1                    X   segment 2                         Y   segment 3            Z
//   producers  prep(c-1), first part            prep(c-1) rest, dM(c) gradients      scores(c-1): M^T, T
//   consumers  i-split(c): dSA dR dV, dS^T      j-split(c) products against S0/dU    + dM products, tail(c)
//
// Producers (waves 4..7) own all input loads of the NEXT chunk (prefetched one iteration ahead) and fill LDS
// buffer (c-1)&1; consumers (waves 0..3) read buffer c&1, own the two register copies of dS (S^T tiles split
// over value columns i, S tiles split over key columns j -- no cross-wave reduction anywhere) and issue all
// stores.  For the element-wise tail the four C-layout results are bounced through a wave-private LDS strip into
// "one token, 4 consecutive channels per lane" (with the decay-gradient integrand, formed in C layout from the
// operand images, so that of the raw inputs only w is read a second time), and the decay
// prefix / gradient suffix sums over the 16 tokens are in-row DPP scans (a register-only variant with quad
// transposes and cross-row shuffles was measured 3x slower: 477 vs ~150 instructions).
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked_bwd.h>
#include <wkv7_fwd_v3.h>     // regmm_bf16x3

namespace wkv7c {

constexpr int RS = 68;              // row stride (floats) of the tail bounce strips: C-layout writes (rows 4g+r) and
                                    // token-per-lane float4 reads (rows c16) are both bank-conflict free
struct BufB {                       // produced per chunk, double buffered
    uint16_t ab[4][L][TJ];          // Ab_hi Ab_lo Kb_hi Kb_lo            [t][j]
    uint16_t trn[8][N][JT];         // ZtT QtT AhT KhT (hi,lo)            [j][t]
    uint16_t ti[4][L][TJ];          // V  dY  SA_hi  SA_lo                [t][i]
    uint16_t dyT[N][JT];            // dY^T                               [i][t]
    uint16_t sc[4][2][L][SS];       // M_qa^T  M_qk^T  M_zk^T  T^T  (hi,lo) A-operand images
    float cl[N];
};
struct LdsB3 {
    BufB b[2];
    uint16_t opnd[8][L][TJ];        // producers only: Zt Qt Ah Kh (hi,lo) [t][j]
    uint16_t dr[2][L][TJ];          // dR hi,lo    [t][i]   (consumers -> producers' dM, consumers' j-split)
    uint16_t drT[2][N][JT];         // dR^T hi,lo  [i][t]
    uint16_t dsc[8][2][L][SS];      // dM images (producers -> consumers)
    float glast[N];
    unsigned cphase, pphase;        // hand-off counters: +1 per consumer / producer wave at the end of each segment
    unsigned pad_[2];
    float res[3][L][RS];            // dAh dKh G bounced from C layout to "token per lane" for the tail; dZt and dQt
                                    // use the dr/drT area, which is dead in segment 3 (res_mat below)
};
static_assert(sizeof(uint16_t) * (2 * L * TJ + 2 * N * JT) >= sizeof(float) * 2 * L * RS, "dZt,dQt bounce must fit in dr+drT");
static_assert(__builtin_offsetof(LdsB3, dr) % 16 == 0 && __builtin_offsetof(LdsB3, res) % 16 == 0, "float4 reads of the bounce strips");
static_assert(__builtin_offsetof(LdsB3, drT) == __builtin_offsetof(LdsB3, dr) + sizeof(uint16_t) * 2 * L * TJ, "dr and drT contiguous");
DEVFN float* res_mat(LdsB3& lds, int m) {       // m: 0 dZt, 1 dQt, 2 dAh, 3 dKh, 4 G ; row stride RS floats
    return m < 2 ? reinterpret_cast<float*>(&lds.dr[0][0][0]) + m * L * RS : &lds.res[m - 2][0][0];
}

struct RawB { uint2 w, q, k, z, a, v, dy; float4 sa; };

// ------------------------------------------------------------------------------------------ producers
struct KeepB { float ab[4], kb[4], ah[4], kh[4]; };     // values prep_a computes and prep_b stores (balances the segments)
DEVFN void bwd_prep_a(LdsB3& lds, BufB& B, const RawB& raw, int pw, int lane, KeepB& keep) {
    const int c16 = lane & 15, g = lane >> 4, c0 = 16 * pw + 4 * g;
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(raw.w, wr); unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
    float zt[4], qt[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp(wr[e]);
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        const float tot = lane_bcast(x, (lane & 48) | 15);
        const float cc = fast_exp(x), cp = fast_exp(x - lw), ic = fast_exp(-x), cb = fast_exp(tot - x);
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; keep.ah[e] = a[e] * ic; keep.kh[e] = k[e] * ic;
        keep.ab[e] = a[e] * cb; keep.kb[e] = k[e] * cb; cend[e] = cc;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&lds.opnd[0][c16][c0], hh); st8(&lds.opnd[1][c16][c0], ll);
    st_b16x4_T(B.trn[0], c0, c16, hh); st_b16x4_T(B.trn[1], c0, c16, ll);
    split4(qt, hh, ll); st8(&lds.opnd[2][c16][c0], hh); st8(&lds.opnd[3][c16][c0], ll);
    st_b16x4_T(B.trn[2], c0, c16, hh); st_b16x4_T(B.trn[3], c0, c16, ll);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[c0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}
// The Ah / Kh images are first read by the scores (third segment) and by the consumers of the NEXT iteration, so they
// are written here, in the producers' short second segment, not in prep_a (first segment: 3.1k vs the consumers' 2.4k).
DEVFN void bwd_prep_b(LdsB3& lds, BufB& B, const RawB& raw, int pw, int lane, const KeepB& keep) {
    const int c16 = lane & 15, g = lane >> 4, c0 = 16 * pw + 4 * g;
    uint2 hh, ll;
    split4(keep.ah, hh, ll); st8(&lds.opnd[4][c16][c0], hh); st8(&lds.opnd[5][c16][c0], ll);
    st_b16x4_T(B.trn[4], c0, c16, hh); st_b16x4_T(B.trn[5], c0, c16, ll);
    split4(keep.kh, hh, ll); st8(&lds.opnd[6][c16][c0], hh); st8(&lds.opnd[7][c16][c0], ll);
    st_b16x4_T(B.trn[6], c0, c16, hh); st_b16x4_T(B.trn[7], c0, c16, ll);
    split4(keep.ab, hh, ll); st8(&B.ab[0][c16][c0], hh); st8(&B.ab[1][c16][c0], ll);
    split4(keep.kb, hh, ll); st8(&B.ab[2][c16][c0], hh); st8(&B.ab[3][c16][c0], ll);
    st8(&B.ti[0][c16][c0], raw.v);
    st8(&B.ti[1][c16][c0], raw.dy);
    st_b16x4_T(B.dyT, c0, c16, raw.dy);
    const float sav[4] = {raw.sa.x, raw.sa.y, raw.sa.z, raw.sa.w};
    split4(sav, hh, ll); st8(&B.ti[2][c16][c0], hh); st8(&B.ti[3][c16][c0], ll);
}

template <bool DBL_BF16>
DEVFN void bwd_scores(LdsB3& lds, BufB& B, int pw, int lane) {
    const int c16 = lane & 15, g = lane >> 4;
    if (pw != 0) {
        // (Qt Ah^T) / (Qt Kh^T) / (Zt Kh^T) [t][s] held as lane c16 = s, r <-> t  = A image of the transposed score
        const int mx = pw == 3 ? 0 : 2, my = pw == 1 ? 4 : 6;
        f32x4 d = dot64<true, true>(lds.opnd[mx], lds.opnd[mx + 1], lds.opnd[my], lds.opnd[my + 1], c16, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = (pw == 3 ? (c16 < 4 * g + r) : (c16 <= 4 * g + r)) ? d[r] : 0.f;
        uint2 hh, ll; split4(d, hh, ll);
        st8(&B.sc[pw - 1][0][c16][4 * g], hh); st8(&B.sc[pw - 1][1][c16][4 * g], ll);
    } else {
        f32x4 X = dot64<true, true>(lds.opnd[0], lds.opnd[1], lds.opnd[4], lds.opnd[5], c16, g);    // [t][s]
        f32x4 XT, Tc;                                                                                // XT = [s][t]
        // transpose through this wave's own output slot (B.sc[3], 1536 B, written for real below) instead of a second
        // 64-deep product
        float (*scratch)[20] = reinterpret_cast<float (*)[20]>(&B.sc[3][0][0][0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r] = (c16 < 4 * g + r) ? X[r] : 0.f;
            scratch[4 * g + r][c16] = X[r];
        }
        wave_lds_fence();
        {
            const float4 t4 = *reinterpret_cast<const float4*>(&scratch[c16][4 * g]);
            XT[0] = t4.x; XT[1] = t4.y; XT[2] = t4.z; XT[3] = t4.w;
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) Tc[r] = X[r] + ((4 * g + r == c16) ? 1.f : 0.f);
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            const f32x4 XTn = DBL_BF16 ? regmm_bf16x3(X, XT) : regmm_f32(X, XT);         // (X^T)^2
            f32x4 Xn = X;
            if (level < 2) Xn = DBL_BF16 ? regmm_bf16x3(XT, X) : regmm_f32(XT, X);        // X^2
            const f32x4 D = DBL_BF16 ? regmm_bf16x3(XTn, Tc) : regmm_f32(XTn, Tc);        // X_k T
#pragma unroll
            for (int r = 0; r < 4; ++r) Tc[r] += D[r];
            X = Xn; XT = XTn;
        }
        uint2 hh, ll; split4(Tc, hh, ll);                      // Tc[r] = T[4g+r][c16] = T^T[c16][4g+r]
        st8(&B.sc[3][0][c16][4 * g], hh); st8(&B.sc[3][1][c16][4 * g], ll);
    }
}

DEVFN void st_dsc(LdsB3& lds, int slot, f32x4 d, int c16, int g, int mode) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int x = 4 * g + r;
        const bool keep = mode == 0 ? (c16 < x) : mode == 1 ? (x < c16) : mode == 2 ? (c16 <= x) : (x <= c16);
        d[r] = keep ? d[r] : 0.f;
    }
    uint2 h, l;
    split4(d, h, l);
    st8(&lds.dsc[slot][0][c16][4 * g], h);
    st8(&lds.dsc[slot][1][c16][4 * g], l);
}
// Both orientations of a score gradient from ONE 64-deep product: d[r] at lane (g, c16) is D[4g+r][c16]; the image of
// D^T is four 8-byte row stores (st_dsc), the image of D itself a 2-byte scatter -- instead of a second product
// (8 ds_read_b128 + up to 6 MFMAs per wave and chunk; the LDS pipeline is one of the two things that bound the kernel).
DEVFN void st_dsc_both(LdsB3& lds, int slot_t, int slot_n, f32x4 d, int c16, int g, bool inclusive) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int x = 4 * g + r;
        d[r] = (inclusive ? (c16 <= x) : (c16 < x)) ? d[r] : 0.f;
    }
    uint2 h, l;
    split4(d, h, l);
    st8(&lds.dsc[slot_t][0][c16][4 * g], h);                 // image[c16][4g+r] = D[4g+r][c16]
    st8(&lds.dsc[slot_t][1][c16][4 * g], l);
    st_b16x4_T(lds.dsc[slot_n][0], 4 * g, c16, h);           // image[4g+r][c16] = D[4g+r][c16]
    st_b16x4_T(lds.dsc[slot_n][1], 4 * g, c16, l);
}
DEVFN void bwd_dscores(LdsB3& lds, const BufB& B, int pw, int lane) {
    const int c16 = lane & 15, g = lane >> 4;
    if (pw == 0) {          // dM_za = tril_(dR SA^T)
        st_dsc_both(lds, 1, 0, dot64<true, true>(lds.dr[0], lds.dr[1], B.ti[2], B.ti[3], c16, g), c16, g, false);
    } else if (pw == 1) {   // dM_zk = tril_(dR V^T)
        st_dsc_both(lds, 3, 2, dot64<true, false>(lds.dr[0], lds.dr[1], B.ti[0], B.ti[0], c16, g), c16, g, false);
    } else if (pw == 2) {   // dM_qa = tril(dY SA^T)
        st_dsc_both(lds, 5, 4, dot64<false, true>(B.ti[1], B.ti[1], B.ti[2], B.ti[3], c16, g), c16, g, true);
    } else {                // dM_qk = tril(dY V^T)
        st_dsc_both(lds, 7, 6, dot64<false, false>(B.ti[1], B.ti[1], B.ti[0], B.ti[0], c16, g), c16, g, true);
    }
}

// acc += M B for a 16x16 score-type A (hi/lo images) and a register B given as (hi, lo): with A = [M_h | M_l] (one
// ds_read2_b64, no register assembly), B1 = [b_h ; b_h] gives M_h b_h + M_l b_h and B2 = [b_l ; 0] adds M_h b_l.
// The two B operands are built once per (hi, lo) pair and shared by the products that use it (mm_small of
// wkv7_chunked_bwd.h duplicates M_h instead: 7 register moves per product).
struct BPair { bf16x8 dup, lo0; };
DEVFN BPair make_bpair(uint2 bh, uint2 bl) { return BPair{mk8(bh, bh), mk8(bl.x, bl.y, 0u, 0u)}; }
DEVFN f32x4 mm_small2(f32x4 acc, const uint16_t (*Mh)[SS], const uint16_t (*Ml)[SS], int row, int g, const BPair& b) {
    const bf16x8 a = mk8(ld8(&Mh[row][4 * g]), ld8(&Ml[row][4 * g]));
    acc = mfma_16x16x32_bf16(a, b.dup, acc);
    return mfma_16x16x32_bf16(a, b.lo0, acc);
}

// ------------------------------------------------------------------------------------------ kernel
// MODE bit 0: hand-off counters instead of workgroup barriers; bit 1: T doubling on the bf16 matrix core (bf16x3)
// instead of the f32 one.  Both are kept selectable for same-process A/B timing (benchmarks/wkv7_micro.py).
// TPAR: sequence-parallel launch -- blockIdx.x = (b*H + h) * nseg + seg, the workgroup walks chunks [c_lo, c_hi) of its
// head, starting from dL/dS = ds_in[b,h,seg] and leaving dL/dS at the start of the range in ds_out (BwdArgs).  The
// recurrence is linear in dS (dS_start = dS_end M^T + C with the forward's segment map M), so the host runs the
// segments twice: once from zero for C, a 64x64 scan over the segments, once from the true dS_end (wkv7.py).
template <bool PROF, int MODE = 0, bool TPAR = false>
__global__ __launch_bounds__(512) void bwd_kernel_v3(BwdArgs p) {
    constexpr bool FLAGS = (MODE & 1) != 0;
    LdsB3& lds = *reinterpret_cast<LdsB3*>(dyn_lds());
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
    WKV_STAMP_DECL
    // hand-off points: counters (FLAGS) or, at the three places marked X / Y / Z, workgroup barriers
    auto wait_c = [&](unsigned target) { if (FLAGS) lds_flag_wait(&lds.cphase, target); };
    auto wait_p = [&](unsigned target) { if (FLAGS) lds_flag_wait(&lds.pphase, target); };
    auto done_c = [&]() { if (FLAGS) lds_flag_add(&lds.cphase); };
    auto done_p = [&]() { if (FLAGS) lds_flag_add(&lds.pphase); };
    auto bar = [&]() { if (!FLAGS) block_sync_lds(); };

    if (wave >= 4) {
        // ================================================================== producers
        const int pw = wave - 4;
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
        block_sync_lds();                                   // counters are zeroed
        // prologue ("iteration -1"): produce the last chunk completely; 3 increments per wave like every iteration
        {
            KeepB keep;
            RawB cur = raw;
            if (c_hi - 1 > c_lo) fetch(raw, c_hi - 2);
            bwd_prep_a(lds, lds.b[(c_hi - 1) & 1], cur, pw, lane, keep);
            done_p();
            bwd_prep_b(lds, lds.b[(c_hi - 1) & 1], cur, pw, lane, keep);
            done_p();
            wait_p(8u);                                     // every producer's operand images are in LDS
            bar(); bar();                                   // X, Y
            bwd_scores<(MODE & 2) != 0>(lds, lds.b[(c_hi - 1) & 1], pw, lane);
            done_p();
            bar();                                          // Z
        }
        unsigned it = 0;                                    // iteration k: consumers process chunk c, producers build c-1
        for (int c = c_hi - 1; c >= c_lo; --c, ++it) {
            const bool more = c > c_lo;
            KeepB keep;
            RawB cur = raw;
            // P1: operand images of chunk c-1.  `opnd` is free once every producer finished the previous scores; the
            // target buffer b[(c-1)&1] once every consumer finished the previous chunk.
            wait_p(12u * (it + 1));
            wait_c(12u * it);
            WKV_STAMP(0)
            if (more) {
                if (c - 1 > c_lo) fetch(raw, c - 2);
                bwd_prep_a(lds, lds.b[(c - 1) & 1], cur, pw, lane, keep);
            }
            done_p();
            bar();                                          // X
            WKV_STAMP(1)
            // P2: rest of the images (nobody reads them before the next iteration), then the score gradients of
            // chunk c, which need the consumers' dR(c)
            if (more) bwd_prep_b(lds, lds.b[(c - 1) & 1], cur, pw, lane, keep);
            WKV_STAMP(2)
            wait_c(12u * it + 4u);
            wait_p(12u * (it + 1) + 4u);                    // keeps "count >= base + 4s  =>  all waves finished segment s"
            WKV_STAMP(3)
            bwd_dscores(lds, lds.b[c & 1], pw, lane);
            done_p();
            // P3: scores of chunk c-1 from all four producers' operand images (waiting for the P2 count also covers P1)
            wait_p(12u * (it + 1) + 8u);
            bar();                                          // Y
            WKV_STAMP(4)
            if (more) bwd_scores<(MODE & 2) != 0>(lds, lds.b[(c - 1) & 1], pw, lane);
            done_p();
            bar();                                          // Z
            WKV_STAMP(5)
        }
        WKV_STAMP_FLUSH(256, 8, 6)
        return;
    }

    // ====================================================================== consumers
    const float* sbase = p.s + (size_t)bh * nchunk * N * N;
    f32x4 dS1[4], dS2[4], SL[4], S0n[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { dS1[x] = zero4(); dS2[x] = zero4(); }
    if (TPAR && p.ds_in) {       // dS1[jb][r] = dS[16w+c16][16jb+4g+r] (the S^T tiles), dS2[ib][r] = dS[16ib+4g+r][16w+c16]
        const float* di = p.ds_in + (size_t)blockIdx.x * N * N;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float4 t = *reinterpret_cast<const float4*>(di + (size_t)(16 * wave + c16) * N + 16 * x + 4 * g);
            dS1[x][0] = t.x; dS1[x][1] = t.y; dS1[x][2] = t.z; dS1[x][3] = t.w;
#pragma unroll
            for (int r = 0; r < 4; ++r) dS2[x][r] = di[(size_t)(16 * x + 4 * g + r) * N + 16 * wave + c16];
        }
    }
    auto load_state = [&](f32x4* dst, int cidx) {      // s[cidx] as S[i][j] tiles: [ib][r] = S[16ib+4g+r][16w+c16]
        const float* sp = sbase + (size_t)cidx * N * N + (size_t)(16 * wave + c16) * N + 4 * g;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float4 x = *reinterpret_cast<const float4*>(sp + 16 * ib);
            dst[ib][0] = x.x; dst[ib][1] = x.y; dst[ib][2] = x.z; dst[ib][3] = x.w;
        }
    };
    load_state(SL, c_hi - 1);
    if (c_hi > 1) load_state(S0n, c_hi - 2);
    else {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) S0n[ib] = zero4();
    }
    // tail layout: lane (t = c16, g) <-> token c16, channels 16w + 4g .. +3 (in-row DPP scans over the 16 tokens)
    const int c0 = 16 * wave + 4 * g;
    const unsigned row_off = (unsigned)c16 * ts + (unsigned)c0;
    const unsigned dv_off = (unsigned)(4 * g) * ts + 16u * wave + c16;

    if (tid == 0) { lds.cphase = 0u; lds.pphase = 0u; }
    block_sync_lds();                                          // counters are zeroed
    bar(); bar(); bar();                                       // prologue X, Y, Z
    unsigned it = 0;
    for (int c = c_hi - 1; c >= c_lo; --c, ++it) {
        const BufB& B = lds.b[c & 1];
        // C1 may start when buffer c&1 is complete (producers' previous iteration) and every consumer has left the
        // previous tail (its bounce strips alias dr/drT, which this segment overwrites)
        wait_p(12u * (it + 1));
        wait_c(12u * it);
        WKV_STAMP(5)
        const size_t cbase = head_base + (size_t)c * L * ts;
        // ---------------------------------------------------------------- segment 1: i-split (i = 16w + c16)
        uint2 rh, rl;
        const uint2 dy = ld8(&B.dyT[16 * wave + c16][4 * g]);
        {
            bf16x8 bh[2], bl[2];
            tiles_to_b(dS1, 1.f, bh, bl);
            f32x4 dSA = mm_small_exact(zero4(), B.sc[0][0], B.sc[0][1], c16, g, dy);                // M_qa^T dY
            dSA = mm_perm<true>(dSA, B.ab[0], B.ab[1], c16, g, bh, bl);                             // Ab dS^T
            uint2 xh, xl;
            split4(dSA, xh, xl);
            const f32x4 dR = mm_small2(zero4(), B.sc[3][0], B.sc[3][1], c16, g, make_bpair(xh, xl));             // T^T dSA
            split4(dR, rh, rl);
            f32x4 dV = mm_small_exact(zero4(), B.sc[1][0], B.sc[1][1], c16, g, dy);                 // M_qk^T dY
            dV = mm_perm<true>(dV, B.ab[2], B.ab[3], c16, g, bh, bl);                               // Kb dS^T
            dV = mm_small2(dV, B.sc[2][0], B.sc[2][1], c16, g, make_bpair(rh, rl));                              // M_zk^T dR
            st_b16x4_col(lds.dr[0], 4 * g, 16 * wave + c16, rh);
            st_b16x4_col(lds.dr[1], 4 * g, 16 * wave + c16, rl);
            st8(&lds.drT[0][16 * wave + c16][4 * g], rh);
            st8(&lds.drT[1][16 * wave + c16][4 * g], rl);
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
        }
        done_c();
        WKV_STAMP(0)
        wait_c(12u * it + 4u);                              // X: every consumer's dR(c) is in LDS
        bar();
        WKV_STAMP(1)
        // ---------------------------------------------------------------- segment 2: j-split against S0 / dU (j = 16w + c16)
        const int j = 16 * wave + c16;
        const float clj = B.cl[j];
        f32x4 S0[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) S0[ib] = S0n[ib];
        if (c > 1) load_state(S0n, c - 2);
        else {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) S0n[ib] = zero4();
        }
        // raw decay of this chunk in row layout for the tail (the only input read twice; the producers read it two
        // chunks ago).  q,k,z,a are NOT re-read: their products with the gradients are formed in C layout from the
        // hi/lo operand images already in registers (dq q = dQt Qt, da a = dAh Ah, dk k = dKh Kh, dz z = dZt Zt).
        const uint2 tw = *reinterpret_cast<const uint2*>(p.w + cbase + row_off);
        const uint2 zth = ld8(&B.trn[0][j][4 * g]), ztl = ld8(&B.trn[1][j][4 * g]);
        const uint2 qth = ld8(&B.trn[2][j][4 * g]), qtl = ld8(&B.trn[3][j][4 * g]);
        f32x4 dZt, dQt, dAh, dKh;
        float gl = 0.f;
        {
            bf16x8 s0h[2], s0l[2], duh[2], dul[2];
            tiles_to_b(S0, 1.f, s0h, s0l);
            tiles_to_b(dS2, clj, duh, dul);
            dZt = mm_perm<true>(zero4(), lds.dr[0], lds.dr[1], c16, g, s0h, s0l);                  // dR S0
            dQt = mm_perm<false>(zero4(), B.ti[1], B.ti[1], c16, g, s0h, s0l);                     // dY S0
            dAh = mm_perm<true>(zero4(), B.ti[2], B.ti[3], c16, g, duh, dul);                      // SA dU
            dKh = mm_perm<false>(zero4(), B.ti[0], B.ti[0], c16, g, duh, dul);                     // V dU
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) gl = fmaf(dS2[ib][r], SL[ib][r], gl);
            gl += lane_xor16(gl);
            gl += lane_xor32(gl);
            if (g == 0) lds.glast[j] = gl;
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
                SL[ib] = S0[ib];
            }
        }
        done_c();
        WKV_STAMP(2)
        wait_c(12u * it + 8u);                              // Y: nobody reads dr/drT any more (bounce strips alias them)
        wait_p(12u * (it + 1) + 8u);                        //    and the dM(c) images are ready
        bar();
        WKV_STAMP(3)
        // ---------------------------------------------------------------- segment 3: dM products + tail
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
            // decay-gradient integrand G[t][j] = dq q - da a - dk k + (dz z)[t+1]   (t = 4g + r, C layout)
            float zh[4], zl[4], qh[4], ql[4], ah[4], al[4], kh[4], kl[4], pz[4];
            unpack4(zth, zh); unpack4(ztl, zl); unpack4(qth, qh); unpack4(qtl, ql);
            unpack4(ahh, ah); unpack4(ahl, al); unpack4(khh, kh); unpack4(khl, kl);
#pragma unroll
            for (int r = 0; r < 4; ++r) pz[r] = dZt[r] * (zh[r] + zl[r]);
            float pzn = lane_bcast(pz[0], (lane + 16) & 63);          // token 4(g+1) of the same column
            pzn = g == 3 ? 0.f : pzn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float nx = r < 3 ? pz[r < 3 ? r + 1 : 3] : pzn;
                res_mat(lds, 4)[(4 * g + r) * RS + j] = dQt[r] * (qh[r] + ql[r]) - dAh[r] * (ah[r] + al[r]) - dKh[r] * (kh[r] + kl[r]) + nx;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            res_mat(lds, 0)[(4 * g + r) * RS + j] = dZt[r];
            res_mat(lds, 1)[(4 * g + r) * RS + j] = dQt[r];
            res_mat(lds, 2)[(4 * g + r) * RS + j] = dAh[r];
            res_mat(lds, 3)[(4 * g + r) * RS + j] = dKh[r];
        }
        wave_lds_fence();           // res columns [16w,16w+16) and glast are written and read by this wave only
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
                gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
                dw[e] = gt * lw;
            }
            const size_t o = cbase + row_off;
            *reinterpret_cast<uint2*>(p.dw + o) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
            *reinterpret_cast<uint2*>(p.dq + o) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
            *reinterpret_cast<uint2*>(p.dk + o) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
            *reinterpret_cast<uint2*>(p.dz + o) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
            *reinterpret_cast<uint2*>(p.da + o) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
        }
        done_c();
        bar();                                              // Z
        WKV_STAMP(4)
    }
    if (TPAR && p.ds_out) {
        float* dout = p.ds_out + (size_t)blockIdx.x * N * N;
#pragma unroll
        for (int x = 0; x < 4; ++x)
            *reinterpret_cast<float4*>(dout + (size_t)(16 * wave + c16) * N + 16 * x + 4 * g) = make_float4(dS1[x][0], dS1[x][1], dS1[x][2], dS1[x][3]);
    }
    WKV_STAMP_FLUSH(0, 0, 6)
}

}  // namespace wkv7c
