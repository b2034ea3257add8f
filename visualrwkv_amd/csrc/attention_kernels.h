// Softmax attention forward for the frozen ViT towers (SigLIP so400m: head dim 72, DINOv2-L / SAM ViT-B: 64) on the
// gfx950 matrix cores.  o = softmax(q k^T / sqrt(D) [+ decomposed relative-position bias]) v, no mask, any sequence
// length (tail keys masked).
//
// Replaces (forward only -- the towers are frozen, src/model.py:349,368) the attention inside timm's
// VisionTransformer blocks that `SamDinoSigLIPViTBackbone.forward` runs (VisualRWKV-v7/v7.00/src/vision.py:123-134)
// and `Attention.forward` of the SAM encoder INCLUDING `add_decomposed_rel_pos` (src/sam.py:289-305, 392-426): the
// (B, heads, L, L) bias tensor of the reference is never formed; each workgroup derives the 2 S numbers per query it
// needs (S = window side) from q and the two (2S-1, D) tables with a few MFMAs and keeps them in LDS / registers.
//
// One workgroup = 4 waves = 64 QT query rows of one (batch, head); each wave owns QT tiles of 16 queries and re-uses
// every K / V fragment it reads from LDS for all of them (QT = 2 balances LDS bandwidth against the matrix cores).
// Keys / values are streamed in tiles of 64, both ROW-MAJOR, through a double-buffered LDS image (global loads of tile
// j+1 are in flight during the MFMAs of tile j; one barrier per tile).  The score tile is computed TRANSPOSED,
// S^T = K Q^T, so a lane holds scores of ONE query (column) for 16 keys: the online-softmax max/sum are in-lane
// reductions plus two cross-row xors, and P^T in C layout is directly the B operand of O^T += V^T P^T (k-slots permuted
// to the accumulator map, as in the WKV7 kernels) -- no LDS round trip for P.  V^T fragments come from the row-major V
// image through ds_read_b64_tr_b16 (no transposed copy).  Row strides: K = odd multiple of 16 B (b128 reads of 16
// consecutive rows hit 16 distinct bank quads), V = 160 B (the 8 rows of a tr-read phase tile the 64 banks).
// Workgroups of one (batch, head) are placed on one XCD (they share K / V through that XCD's L2).
#pragma once
#include <gfx950_prims.h>
#include <type_traits>

namespace vattn {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
DEVFN bf16x8 mk8(uint4 u) { return mk8(u.x, u.y, u.z, u.w); }

struct Args {
    const uint16_t *q, *k, *v;     // bf16; element (b, l, h, d) at b*sb + l*sl + h*sh + d
    uint16_t* o;                   // bf16 (B, L, H, D) contiguous
    long sb, sl, sh;               // strides of q/k/v in elements (shared by the three: slices of one qkv tensor)
    int L, H;
    float scale_log2e;             // 1/sqrt(D) * log2(e)
    const uint16_t *rel_h, *rel_w; // bf16 (2S-1, D) tables of the decomposed relative-position bias, or null
    int nqb, BH;                   // query blocks per (batch, head); batch * heads
};

constexpr int KT = 64;             // keys per tile

template <int D, int QT, int S> struct Geo {
    static constexpr int DP = (D + 31) / 32 * 32;      // contraction length of QK^T padded to the MFMA K
    static constexpr int NKB = DP / 32;
    static constexpr int DT = (D + 15) / 16;           // 16-wide tiles of the head dim in O^T
    static constexpr int KS = DP + 8;                  // K tile row stride (elements): DP/2 + 4 dwords = 4 mod 8
    static constexpr int VS = 80;                      // V tile row stride (elements) = 40 dwords
    static constexpr int KCPR = DP / 8, VCPR = DT * 2; // 16-byte chunks per staged row
    static constexpr int NKC = (KT * KCPR + 255) / 256, NVC = (KT * VCPR + 255) / 256;
    static constexpr int SP = S + 1;                   // bias table row stride (floats)
    static constexpr int NQ = 64 * QT;
    static constexpr int K_BYTES = 2 * KT * KS * 2, V_BYTES = 2 * KT * VS * 2;
    static constexpr int TAB_BYTES = S == 0 ? 0 : (S == KT ? 1 : 2) * NQ * SP * 4;
    static constexpr int LDS_BYTES = K_BYTES + V_BYTES + TAB_BYTES;
};

// Registers are capped at 168 (three waves per SIMD instead of two) where that pays: QT = 2 at D = 72 (196 -> 168 VGPRs, 11 spilled outside the
// key loop: SigLIP 0.1655 -> 0.154 ms) and SAM's global blocks with the rel-pos bias (1.31 -> 1.27 ms); the windowed rel-pos instantiation
// spills inside its loop under the cap (+25 %) and D = 64 plain already runs three waves (profiles/r5_attention_occupancy.txt).
constexpr int min_waves_per_simd(int D, int QT, int S) { return (QT == 2 && ((D == 72 && S == 0) || S == KT)) ? 3 : 1; }
template <int D, int QT, int S>
__global__ __launch_bounds__(256) KERNEL_MIN_WAVES(min_waves_per_simd(D, QT, S)) void fwd_kernel(Args p) {
    using G = Geo<D, QT, S>;
    constexpr int DP = G::DP, NKB = G::NKB, DT = G::DT, KS = G::KS, VS = G::VS, SP = G::SP;
    static_assert(DT * 16 <= VS, "V row too short");
    char* lds = dyn_lds();
    uint16_t* kbuf = reinterpret_cast<uint16_t*>(lds);                       // [2][KT][KS]
    uint16_t* vbuf = reinterpret_cast<uint16_t*>(lds + G::K_BYTES);          // [2][KT][VS]
    float* tab_h = reinterpret_cast<float*>(lds + G::K_BYTES + G::V_BYTES);  // [NQ][SP]
    float* tab_w = S == KT ? tab_h : tab_h + G::NQ * SP;                     // S == KT: used once, before tab_h

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    int bh, qb;
    {
        const int id = blockIdx.x;
        if ((p.BH & 7) == 0) {                      // consecutive workgroup ids go to consecutive XCDs
            const int slot = id >> 3;
            bh = (slot / p.nqb) * 8 + (id & 7);
            qb = slot % p.nqb;
        } else {
            bh = id / p.nqb;
            qb = id % p.nqb;
        }
    }
    const int b = bh / p.H, h = bh % p.H;
    const int qw0 = wave * (16 * QT);                           // first query of this wave inside the workgroup
    const int q0 = qb * G::NQ + qw0;
    const long base = (long)b * p.sb + (long)h * p.sh;
    const int L = p.L;

    // Q fragments (B operand: [k = d][n = q]): lane (g, c16 = q) holds d = 32kb + 8g .. +7
    bf16x8 qf[QT][NKB];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int qrow = q0 + 16 * i + c16;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int d0 = 32 * kb + 8 * g;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (qrow < L && d0 < D) u = *reinterpret_cast<const uint4*>(p.q + base + (long)qrow * p.sl + d0);
            qf[i][kb] = mk8(u);
        }
    }

    // ---- decomposed relative-position bias: tab[q][kc] = log2e * q . rel[qc - kc + S - 1]  (unscaled q, sam.py:415-420)
    f32x2 bw[S == KT ? QT : 1][4][2];
    if constexpr (S > 0) {
        constexpr int NJ = (2 * S - 1 + 15) / 16;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {                  // 0: width table, 1: height table
            const uint16_t* rel = pass == 0 ? p.rel_w : p.rel_h;
            float* tab = pass == 0 ? tab_w : tab_h;
            for (int jt = 0; jt < NJ; ++jt) {
                bf16x8 rf[NKB];
                const int jrow = 16 * jt + c16;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const int d0 = 32 * kb + 8 * g;
                    uint4 u = make_uint4(0, 0, 0, 0);
                    if (jrow < 2 * S - 1 && d0 < D) u = *reinterpret_cast<const uint4*>(rel + (long)jrow * D + d0);
                    rf[kb] = mk8(u);
                }
#pragma unroll
                for (int i = 0; i < QT; ++i) {
                    f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) r4 = mfma_16x16x32_bf16(rf[kb], qf[i][kb], r4);
                    const int qi = q0 + 16 * i + c16;
                    const int qc = pass == 0 ? qi % S : qi / S;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kc = qc + S - 1 - (16 * jt + 4 * g + r);
                        if (qi < L && kc >= 0 && kc < S) tab[(qw0 + 16 * i + c16) * SP + kc] = r4[r] * 1.4426950408889634f;
                    }
                }
            }
            if constexpr (S == KT) {
                if (pass == 0) {                                 // key column of a lane's 16 scores is the same in every tile
                    wave_lds_fence();
#pragma unroll
                    for (int i = 0; i < QT; ++i)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) bw[i][t][r >> 1][r & 1] = tab_w[(qw0 + 16 * i + c16) * SP + 16 * t + 4 * g + r];
                    wave_lds_fence();
                }
            }
        }
    }

    f32x4 acc[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        m_run[i] = -1e30f; l_run[i] = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) { acc[i][t][0] = 0.f; acc[i][t][1] = 0.f; acc[i][t][2] = 0.f; acc[i][t][3] = 0.f; }
    }

    // ---- K / V staging: global -> registers (in flight during the previous tile's MFMAs) -> LDS image.
    // Per-thread chunk offsets are fixed; per tile only the wave-uniform base moves (scalar base + 32-bit lane offset
    // loads).  Rows past L (last tile) are clamped to row L-1: their scores are masked to -inf, so p = 0 exactly.
    constexpr bool K_GUARD = DP != D || (KT * G::KCPR) % 256 != 0, V_GUARD = DT * 16 != D || (KT * G::VCPR) % 256 != 0;
    uint4 kr[G::NKC], vr[G::NVC];
    int krow[G::NKC], vrow[G::NVC];
    uint32_t kcol[G::NKC], vcol[G::NVC], klds[G::NKC], vlds[G::NVC];       // byte offsets
    bool kok[G::NKC], vok[G::NVC];
#pragma unroll
    for (int c = 0; c < G::NKC; ++c) {
        const int idx = tid + 256 * c, key = idx / G::KCPR, d0 = (idx % G::KCPR) * 8;
        kok[c] = key < KT && d0 < D;
        krow[c] = kok[c] ? key : 0;
        kcol[c] = kok[c] ? 2u * d0 : 0u;
        klds[c] = 2u * ((key < KT ? key : 0) * KS + d0);
    }
#pragma unroll
    for (int c = 0; c < G::NVC; ++c) {
        const int idx = tid + 256 * c, key = idx / G::VCPR, d0 = (idx % G::VCPR) * 8;
        vok[c] = key < KT && d0 < D;
        vrow[c] = vok[c] ? key : 0;
        vcol[c] = vok[c] ? 2u * d0 : 0u;
        vlds[c] = 2u * ((key < KT ? key : 0) * VS + d0);
    }
    const uint32_t row_bytes = (uint32_t)(2 * p.sl);
    auto fetch = [&](auto last_tag, const int k0) {
        constexpr bool LAST = decltype(last_tag)::value;       // tile may reach past L
        const char* kb_ = reinterpret_cast<const char*>(p.k + base + (long)k0 * p.sl);
        const char* vb_ = reinterpret_cast<const char*>(p.v + base + (long)k0 * p.sl);
#pragma unroll
        for (int c = 0; c < G::NKC; ++c) {
            const int row = LAST ? (k0 + krow[c] < L ? krow[c] : L - 1 - k0) : krow[c];
            uint4 u = *reinterpret_cast<const uint4*>(kb_ + ((uint32_t)row * row_bytes + kcol[c]));
            if (K_GUARD && !kok[c]) u = make_uint4(0, 0, 0, 0);
            kr[c] = u;
        }
#pragma unroll
        for (int c = 0; c < G::NVC; ++c) {
            const int row = LAST ? (k0 + vrow[c] < L ? vrow[c] : L - 1 - k0) : vrow[c];
            uint4 u = *reinterpret_cast<const uint4*>(vb_ + ((uint32_t)row * row_bytes + vcol[c]));
            if (V_GUARD && !vok[c]) u = make_uint4(0, 0, 0, 0);
            vr[c] = u;
        }
    };
    auto stage = [&](const int buf) {
        char* kd = reinterpret_cast<char*>(kbuf + buf * KT * KS);
        char* vd = reinterpret_cast<char*>(vbuf + buf * KT * VS);
#pragma unroll
        for (int c = 0; c < G::NKC; ++c)
            if ((KT * G::KCPR) % 256 == 0 || tid + 256 * c < KT * G::KCPR) *reinterpret_cast<uint4*>(kd + klds[c]) = kr[c];
#pragma unroll
        for (int c = 0; c < G::NVC; ++c)
            if ((KT * G::VCPR) % 256 == 0 || tid + 256 * c < KT * G::VCPR) *reinterpret_cast<uint4*>(vd + vlds[c]) = vr[c];
    };

    const int ntiles = (L + KT - 1) / KT;
    const int nfull = L / KT;
    auto fetch_tile = [&](const int jt) {
        if (jt < nfull) fetch(std::false_type{}, jt * KT);
        else fetch(std::true_type{}, jt * KT);
    };
    fetch_tile(0);
    stage(0);
    if (ntiles > 1) fetch_tile(1);
    block_sync();

    // One key tile.  MASKED (last, partial tile only): keys >= L get -inf.  Softmax statistics are kept in the log2
    // domain; the running max is only raised when a tile exceeds it by more than 2^8 (p <= 256 stays exact enough in
    // fp32 / bf16 and the accumulator rescale -- 4 DT multiplies per query tile -- becomes rare, wave-uniformly skipped).
    auto tile = [&](auto masked_tag, const int j) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int k0 = j * KT, buf = j & 1;
        const uint16_t* kt = kbuf + buf * KT * KS;
        const uint16_t* vt = vbuf + buf * KT * VS;

        // S^T tiles: st[i][t][r] = q.k of (key = k0 + 16 t + 4 g + r, query = q0 + 16 i + c16)
        f32x4 st[QT][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x8 kf[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) kf[kb] = mk8(*reinterpret_cast<const uint4*>(kt + (16 * t + c16) * KS + 32 * kb + 8 * g));
#pragma unroll
            for (int i = 0; i < QT; ++i) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) s = mfma_16x16x32_bf16(kf[kb], qf[i][kb], s);
                st[i][t] = s;
            }
        }
        bf16x8 pf[QT][2];
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            // y = logit * log2e without the per-(query, tile) constant bhv (S == 0: the raw dot product, scaled inside the exp)
            float bhv = 0.f;
            if constexpr (S == KT) bhv = tab_h[(qw0 + 16 * i + c16) * SP + j];
            float mx = -1e30f;
            const f32x2 sc2 = {p.scale_log2e, p.scale_log2e};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (S == KT) {
                    st[i][t].lo = pk_fma(st[i][t].lo, sc2, bw[i][t][0]);
                    st[i][t].hi = pk_fma(st[i][t].hi, sc2, bw[i][t][1]);
                } else if constexpr (S > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + 16 * t + 4 * g + r;
                        const int kk = (MASKED && key >= L) ? L - 1 : key, kh = kk / S, kw = kk - kh * S;
                        st[i][t][r] = fmaf(st[i][t][r], p.scale_log2e,
                                           tab_h[(qw0 + 16 * i + c16) * SP + kh] + tab_w[(qw0 + 16 * i + c16) * SP + kw]);
                    }
                }
                if constexpr (MASKED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * t + 4 * g + r >= L) st[i][t][r] = -1e30f;
                }
                mx = max3_f32(mx, st[i][t][0], st[i][t][1]);
                mx = max3_f32(mx, st[i][t][2], st[i][t][3]);
            }
            mx = max3_f32(mx, lane_xor16(mx), mx);
            mx = max3_f32(mx, lane_xor32(mx), mx);
            mx = S == 0 ? mx * p.scale_log2e : mx + bhv;
            const float m_old = m_run[i];
            const float m_new = mx > m_old + 8.f ? mx : m_old;
            if (wave_any(m_new != m_old)) {
                const float alpha = fast_exp2(m_old - m_new);
                l_run[i] *= alpha;
                m_run[i] = m_new;
#pragma unroll
                for (int t = 0; t < DT; ++t) { acc[i][t][0] *= alpha; acc[i][t][1] *= alpha; acc[i][t][2] *= alpha; acc[i][t][3] *= alpha; }
            }
            const float sub = S == 0 ? -m_new : bhv - m_new;
            const f32x2 sub2 = {sub, sub};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x2 a = st[i][t].lo, b = st[i][t].hi;
                if constexpr (S == 0) { a = pk_fma(a, sc2, sub2); b = pk_fma(b, sc2, sub2); }
                else { a += sub2; b += sub2; }
                a.x = fast_exp2(a.x); a.y = fast_exp2(a.y); b.x = fast_exp2(b.x); b.y = fast_exp2(b.y);
                st[i][t].lo = a; st[i][t].hi = b;
                ps2 += a;
                ps2 += b;
            }
            float ps = ps2.x + ps2.y;
            ps += lane_xor16(ps);
            ps += lane_xor32(ps);
            l_run[i] += ps;
            // P^T as B operand of each 32-key half: slots e<4 <-> key 4g+e of its first 16-tile, e>=4 <-> second
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
                pf[i][hb] = mk8(cvt_pk_bf16(st[i][2 * hb][0], st[i][2 * hb][1]), cvt_pk_bf16(st[i][2 * hb][2], st[i][2 * hb][3]),
                                cvt_pk_bf16(st[i][2 * hb + 1][0], st[i][2 * hb + 1][1]), cvt_pk_bf16(st[i][2 * hb + 1][2], st[i][2 * hb + 1][3]));
        }
        // O^T += V^T P^T: V^T fragment of d-tile t = column c16 of the [4 keys][16 d] blocks at keys 4g.. and 16+4g..
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const uint16_t* vp = vt + (32 * hb + 4 * g + (c16 >> 2)) * VS + 16 * t + 4 * (c16 & 3);
                const uint2 v0 = lds_read_tr16(vp), v1 = lds_read_tr16(vp + 16 * VS);
                const bf16x8 vf = mk8(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
                for (int i = 0; i < QT; ++i) acc[i][t] = mfma_16x16x32_bf16(vf, pf[i][hb], acc[i][t]);
            }
        if (j + 1 < ntiles) {
            stage(buf ^ 1);                                   // last read by tile j-1; everyone is past that barrier
            if (j + 2 < ntiles) fetch_tile(j + 2);
        }
        block_sync();
    };
    for (int j = 0; j < nfull; ++j) tile(std::false_type{}, j);
    if (nfull < ntiles) tile(std::true_type{}, nfull);

    // O[q][d]: lane (g, c16 = q) holds d = 16t + 4g + r
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int qrow = q0 + 16 * i + c16;
        if (qrow < L) {
            const float inv = 1.f / l_run[i];
            uint16_t* orow = p.o + (((long)b * L + qrow) * p.H + h) * D;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int d0 = 16 * t + 4 * g;
                if (d0 < D)
                    *reinterpret_cast<uint2*>(orow + d0) = make_uint2(cvt_pk_bf16(acc[i][t][0] * inv, acc[i][t][1] * inv),
                                                                      cvt_pk_bf16(acc[i][t][2] * inv, acc[i][t][3] * inv));
            }
        }
    }
}

}  // namespace vattn
