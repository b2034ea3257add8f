// WKV7 forward, chunked MFMA form, full-row memory traffic -- gfx950.
//
// Same algorithm, numerics class and producer / consumer wave split as wkv7_fwd_v3.h (reference:
// VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-52); what changed is how the data crosses the chip boundary.  The v3 kernel moves
// "one token per lane": 8-byte loads (a wave touches 16 token rows x 32 B), 4-byte checkpoint / sa stores, 2-byte y stores -- 24
// store instructions per consumer wave and chunk.  benchmarks/mem_role_probe.hip runs that traffic alone (same occupancy, same
// barrier per chunk; profiles/r4_mem_role_probe_fwd.jsonl): 0.656 ms at B = 16, which is what the v3 kernel takes (0.60-0.65 ms):
// it is bound by its memory role.  The same bytes as full rows -- LDS-DMA in, 16 bytes per lane out -- take 0.583 ms.  Here:
//   * inputs arrive by LDS-DMA (global_load_lds_dwordx4, 8 lanes = one 128-byte token row of a head): w q k z a into a staging
//     image the producer waves read their own 8-byte pieces from, v straight into the [t][i] image the consumers read (ring of 3).
//     All images use the backward kernels' format (wkv7_bwd_v5.h: 128-byte rows, 16-byte slots XOR-swizzled with the row, the
//     swizzle applied on the DMA's source address), so the padded copies of v3 (abn / vn: 13 KB per buffer) are gone and every
//     operand whose k index is the token comes through ds_read_b64_tr_b16 from the one image;
//   * outputs leave through LDS images: the checkpoint as the 16 KB it is in memory (S^T[j][i], 4-byte LDS writes from the
//     accumulator tiles, no register transposes), sa as [t][i] fp32, y as [t][i] bf16 from products issued with swapped operands
//     (token = lane, 4 channels = registers); one chunk later each consumer wave sends 1 KB per instruction: 4 + 1 (+ 1) stores
//     instead of 24;
//   * the state tiles keep their rows in the interleaved order tix() of the backward kernels, so the operands against them are one
//     ds_read_b128 (v3: two 8-byte reads 32 bytes apart).
// LDS 79.5 KB: two workgroups per CU, as before.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>     // SF, SplitF, splitf, regmm_pre, mm_f32_image helpers
#include <wkv7_bwd_rows.h>     // DmaLane / dma_lane and (through it) the v5 image helpers

namespace wkv7f4 {

using wkv7::FwdArgs;
using namespace wkv7c;       // N, L, mk8, split4, unpack4, ld8, st8, zero4, SF, WKV_STAMP*
using namespace wkv7v5;      // IMG, HLI, img_off, hl_off, tix, LaneAddr, lane_addr, ld16, st16, mfma32, dot64, tiles_op
using wkv7v7::DmaLane;
using wkv7v7::dma_lane;

struct BufF4 {                       // per chunk, double buffered (producers run one chunk ahead)
    uint16_t opnd[8][IMG];           // Zt_h Zt_l Qt_h Qt_l Ah_h Ah_l Kh_h Kh_l      [t][j]
    uint16_t sc[2][HLI];             // M_zk, M_qk: image[t][s], [hi4 lo4] per 16 bytes
    float scf[2][L][SF];             // M_qa, T: fp32 [t][s]
    float cl[N];                     // c_L[j]
};
struct LdsF4 {
    BufF4 b[2];
    uint16_t vimg[3][IMG];           // V [t][i] of chunk c in slot c % 3, written by LDS-DMA
    uint16_t stg[5][IMG];            // w q k z a of the chunk the producers prepare next (LDS-DMA; read by the producers only)
    float out_s[N * N];              // checkpoint S^T[j][i] of the chunk the consumers finished last, as it lies in memory
    float out_sa[IMG];               // sa [t][i]
    uint16_t out_y[IMG];             // y [t][i], swizzled like every bf16 image
    unsigned cnt[4];                 // 0: operand images of a chunk written (4 per chunk)  1: producers hold their staging pieces (4)
                                     // 2: consumers have sent the previous chunk's output images (4)
};
static_assert(sizeof(LdsF4) <= 80 * 1024, "two workgroups per CU");

// rows of one chunk: 12 requests of 1 KB -- i = 2 arr + half for w q k z a (staging) and v (ring slot) -- 3 per producer wave
DEVFN void dma_rows(LdsF4& lds, const FwdArgs& p, size_t chunk_base, int c, int pw, unsigned ts, const DmaLane& dl) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = pw + 4 * k;                         // wave-uniform
        const int arr = i >> 1, half = i & 1;
        const uint16_t* src = arr == 0 ? p.w : arr == 1 ? p.q : arr == 2 ? p.k : arr == 3 ? p.z : arr == 4 ? p.a : p.v;
        uint16_t* dst = (arr < 5 ? lds.stg[arr] : lds.vimg[c % 3]) + half * 8 * N;
        lds_dma16_sbase(src + chunk_base + (size_t)half * 8 * ts, dl.b16, dst);
    }
}

// decay scan + scaling + hi/lo split of one lane's 4 channels of one token (as wkv7v7::prep7 without sa)
DEVFN void prep4(BufF4& B, uint2 rw, uint2 rq, uint2 rk, uint2 rz, uint2 ra, int c16, int j0, const LaneAddr& la) {
    float wr[4], q[4], k[4], z[4], a[4];
    unpack4(rw, wr); unpack4(rq, q); unpack4(rk, k); unpack4(rz, z); unpack4(ra, a);
    float zt[4], qt[4], ah[4], kh[4], cend[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lw = -fast_exp2(wr[e] * LOG2E) * LOG2E;          // log2 w_t   (w_t = exp(-exp(w_raw)), wkv7_cuda.cu:21)
        float x = lw;
        x += dpp_shr<1>(x); x += dpp_shr<2>(x); x += dpp_shr<4>(x); x += dpp_shr<8>(x);
        const float cc = fast_exp2(x), ic = fast_exp2(-x), cp = dpp_shr1_fill(cc, 1.f);
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; ah[e] = a[e] * ic; kh[e] = k[e] * ic; cend[e] = cc;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&B.opnd[0][la.own], hh); st8(&B.opnd[1][la.own], ll);
    split4(qt, hh, ll); st8(&B.opnd[2][la.own], hh); st8(&B.opnd[3][la.own], ll);
    split4(ah, hh, ll); st8(&B.opnd[4][la.own], hh); st8(&B.opnd[5][la.own], ll);
    split4(kh, hh, ll); st8(&B.opnd[6][la.own], hh); st8(&B.opnd[7][la.own], ll);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(cend[0], cend[1], cend[2], cend[3]);
}

// one score matrix per producer wave.  D = dot64(X, Y): D[m = 4g+r][n = c16] = X_m . Y_n
DEVFN void scores4(BufF4& B, int pw, int c16, int g, const LaneAddr& la) {
    if (pw == 1 || pw == 3) {                 // image[t][s] = M[t][s] = (Zt | Qt)_t . Kh_s : X = Kh (m = s), Y = Zt | Qt (n = t)
        f32x4 d = dot64<true, true>(B.opnd[6], B.opnd[7], B.opnd[pw == 1 ? 0 : 2], B.opnd[pw == 1 ? 1 : 3], la);
        uint2 hh, ll;
        if (pw == 1) mask_split<false, false>(d, c16, g, hh, ll);       // M_zk: s <  t
        else mask_split<true, false>(d, c16, g, hh, ll);                // M_qk: s <= t
        st16(B.sc[pw == 1 ? 0 : 1] + la.hl, hh, ll);
    } else if (pw == 2) {                     // M_qa[t][s] = Qt_t . Ah_s , s <= t, kept in fp32: X = Ah (m = s), Y = Qt (n = t)
        f32x4 d = dot64<true, true>(B.opnd[4], B.opnd[5], B.opnd[2], B.opnd[3], la);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = (4 * g + r <= c16) ? d[r] : 0.f;
        *reinterpret_cast<float4*>(&B.scf[0][c16][4 * g]) = make_float4(d[0], d[1], d[2], d[3]);
    } else {                                  // T = (I - M_za)^-1 by nilpotent doubling on the bf16 matrix core (wkv7_fwd_v3.h::scores_v3)
        // X[r] = M_za[4g+r][c16] = Zt_{4g+r} . Ah_{c16} (strictly lower); its transpose through this wave's own output slot
        f32x4 X = dot64<true, true>(B.opnd[0], B.opnd[1], B.opnd[4], B.opnd[5], la), XT, TT;
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
        *reinterpret_cast<float4*>(&B.scf[1][c16][4 * g]) = make_float4(TT[0], TT[1], TT[2], TT[3]);   // T[c16][4g+r]
    }
}

// D^T: result lane (c16 = n), registers m = 4g + r of P Q with pt = P^T in C layout (A operand) and qc = Q in C layout; see regmm_f32
DEVFN f32x4 mm_f32_regs_image(f32x4 acc, f32x4 pt, const float (*M)[SF], int c16, int g) {
    const float4 m = *reinterpret_cast<const float4*>(&M[c16][4 * g]);      // Q[k = 4g+r][n = c16] := M[c16][4g+r]
    acc = mfma_16x16x4_f32(pt[0], m.x, acc);
    acc = mfma_16x16x4_f32(pt[1], m.y, acc);
    acc = mfma_16x16x4_f32(pt[2], m.z, acc);
    acc = mfma_16x16x4_f32(pt[3], m.w, acc);
    return acc;
}

template <bool PROF, int PRIO = 1>
__global__ __launch_bounds__(512) void fwd_kernel_v4(FwdArgs p) {
    LdsF4& lds = *reinterpret_cast<LdsF4*>(dyn_lds());
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);                              // token stride (elements)
    const unsigned bh = blockIdx.x;
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    WKV_STAMP_DECL

    if (tid < 4) lds.cnt[tid] = 0u;
    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (one chunk ahead; requests only)
        const int pw = wave - 4;
        if (PRIO > 0) wave_priority<PRIO>();             // producers are the younger half: without this they lose VALU arbitration
        const LaneAddr la = lane_addr(c16, g, pw);
        const DmaLane dl = dma_lane(lane, ts);
        dma_rows(lds, p, head_base, 0, pw, ts, dl);
        vmem_drain();
        block_sync_lds();                               // counters zeroed, rows of chunk 0 landed
        unsigned n_rd = 0;
        for (int c = 0; c <= nchunk; ++c) {            // iteration c produces chunk c (one ahead of the consumers)
            if (c < nchunk) {
                BufF4& B = lds.b[c & 1];
                const uint2 rw = ld8(&lds.stg[0][la.own]), rq = ld8(&lds.stg[1][la.own]), rk = ld8(&lds.stg[2][la.own]);
                const uint2 rz = ld8(&lds.stg[3][la.own]), ra = ld8(&lds.stg[4][la.own]);
                lds_flag_add(&lds.cnt[1]);              // (waits for the reads) ...
                n_rd += 4u;
                if (c + 1 < nchunk) {
                    lds_flag_wait(&lds.cnt[1], n_rd);   // ... all four producers hold their pieces: the staging bytes are free
                    dma_rows(lds, p, head_base + (size_t)(c + 1) * L * ts, c + 1, pw, ts, dl);
                }
                prep4(B, rw, rq, rk, rz, ra, c16, 16 * pw + 4 * g, la);
                lds_flag_add(&lds.cnt[0]);
            }
            WKV_STAMP(0)
            // A: the scores need the operand images of all four producer waves.  Only the producers wait (a counter,
            // not s_barrier): the consumers' second half does not depend on anything produced in this iteration.
            if (c < nchunk) lds_flag_wait(&lds.cnt[0], 4u * (unsigned)(c + 1));
            WKV_STAMP(1)
            if (c < nchunk) scores4(lds.b[c & 1], pw, c16, g, la);
            WKV_STAMP(2)
            vmem_drain();                               // the rows of chunk c + 1 have landed (staging, V slot)
            block_sync_lds();                           // B
            WKV_STAMP(3)
        }
        WKV_STAMP_FLUSH(256, 8, 4)
        return;
    }

    // ---------------------------------------------------------------------- consumers (stores only)
    const int wt = wave;                                                  // which 16 value rows i this wave owns
    const LaneAddr la = lane_addr(c16, g, wt);
    f32x4 S[4];                                                           // S[jb][r] = S^T[j = tix(jb, 4g+r)][i = 16wt + c16]
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) S[jb] = zero4();
    if (p.s0) {                                      // s0[i][j]: 4 consecutive j = tix(jb, 4g) .. +3 per load
        const float* sp = p.s0 + ((size_t)bh * N + 16 * wt + c16) * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 x = *reinterpret_cast<const float4*>(sp + tix(jb, 4 * g));
            S[jb][0] = x.x; S[jb][1] = x.y; S[jb][2] = x.z; S[jb][3] = x.w;
        }
    }
    float* psa = p.sa ? p.sa + head_base : nullptr;
    uint16_t* py = p.y + head_base;
    float* ps = p.s ? p.s + (size_t)bh * nchunk * N * N : nullptr;
    // output images -> memory, 1 KB per instruction: checkpoint rows 16wt .. 16wt+15 (4 x 4 rows of 256 B), sa tokens 4wt .. 4wt+3,
    // y tokens 8wt .. 8wt+7 (waves 0, 1)
    const int sa_t = 4 * wt + (lane >> 4), y_t = 8 * (wt & 1) + (lane >> 3);
    auto send = [&](int c) {
        if (ps) {
            float* s_c = ps + (size_t)c * N * N + 16 * wt * N + 4 * lane;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<float4*>(s_c + k * 256) = *reinterpret_cast<const float4*>(&lds.out_s[(16 * wt + 4 * k) * N + 4 * lane]);
        }
        if (psa) *reinterpret_cast<float4*>(psa + (size_t)c * L * ts + (size_t)sa_t * ts + 4 * (lane & 15)) =
                     *reinterpret_cast<const float4*>(&lds.out_sa[sa_t * N + 4 * (lane & 15)]);
        if (wt < 2) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4*>(py + (size_t)c * L * ts + (size_t)y_t * ts + 8 * (lane & 7)) =
                *reinterpret_cast<const u4*>(&lds.out_y[y_t * N + 8 * ((lane & 7) ^ (y_t & 7))]);
        }
    };

    block_sync_lds();      // counters zeroed, rows of chunk 0 landed
    block_sync_lds();      // B  (producers have filled buffer 0)
    for (int c = 0; c < nchunk; ++c) {
        const BufF4& B = lds.b[c & 1];
        const uint16_t* vimg = lds.vimg[c % 3];
        WKV_STAMP(0)
        if (c > 0) {
            send(c - 1);
            lds_flag_add(&lds.cnt[2]);                  // (waits for the reads) the output images may be overwritten
        }
        bf16x8 bsh[2], bsl[2];
        tiles_op(S, bsh, bsl);
        const uint2 vv = lds_read_tr16(&vimg[la.trc]);                   // V[t = 4g+e][i = 16wt + c16]
        const bf16x8 bvv = mk8(vv, vv);

        // R = M_zk V + Zt S0^T  (three independent accumulator chains)
        f32x4 R = mfma32(ld16(&B.sc[0][la.hl]), bvv, zero4());
        f32x4 Ra = zero4(), Rb = zero4();
        {
            const bf16x8 zh = ld16(&B.opnd[0][la.row[0]]), zl = ld16(&B.opnd[1][la.row[0]]);
            Ra = mfma32(zh, bsh[0], Ra); Ra = mfma32(zh, bsl[0], Ra); Ra = mfma32(zl, bsh[0], Ra);
        }
        {
            const bf16x8 zh = ld16(&B.opnd[0][la.row[1]]), zl = ld16(&B.opnd[1][la.row[1]]);
            Rb = mfma32(zh, bsh[1], Rb); Rb = mfma32(zh, bsl[1], Rb); Rb = mfma32(zl, bsh[1], Rb);
        }
        // Y^T partials that do not need SA: operands swapped, so the result is "token = lane, 4 channels = registers"
        f32x4 Y = mfma32(bvv, ld16(&B.sc[1][la.hl]), zero4());
        f32x4 Ya = zero4(), Yb = zero4();
        {
            const bf16x8 qh = ld16(&B.opnd[2][la.row[0]]), ql = ld16(&B.opnd[3][la.row[0]]);
            Ya = mfma32(bsh[0], qh, Ya); Ya = mfma32(bsl[0], qh, Ya); Ya = mfma32(bsh[0], ql, Ya);
        }
        {
            const bf16x8 qh = ld16(&B.opnd[2][la.row[1]]), ql = ld16(&B.opnd[3][la.row[1]]);
            Yb = mfma32(bsh[1], qh, Yb); Yb = mfma32(bsl[1], qh, Yb); Yb = mfma32(bsh[1], ql, Yb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r] += Ra[r] + Rb[r];
        // SA = T R ,  Y^T += SA^T M_qa^T   (f32 matrix core, operands straight from the accumulators)
        const f32x4 SA = mm_f32_image(zero4(), B.scf[1], c16, g, R);
        const f32x4 Yc = mm_f32_regs_image(zero4(), SA, B.scf[0], c16, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[r] += (Ya[r] + Yb[r]) + Yc[r];
        WKV_STAMP(1)
        if (c > 0) lds_flag_wait(&lds.cnt[2], 4u * (unsigned)c);        // every consumer wave has sent chunk c - 1
        WKV_STAMP(2)
        // sa: lane = i, registers = tokens 4g + r (2-way conflicts: free on 4-byte LDS writes); y: lane = token, 4 channels
#pragma unroll
        for (int r = 0; r < 4; ++r) lds.out_sa[(4 * g + r) * N + 16 * wt + c16] = SA[r];
        st8(&lds.out_y[la.own], make_uint2(cvt_pk_bf16(Y[0], Y[1]), cvt_pk_bf16(Y[2], Y[3])));
        WKV_STAMP(3)
        // S_L^T = diag(c_L) (S0^T + [Ah^T | Kh^T] [SA ; V])
        uint2 sah, sal;
        split4(SA, sah, sal);
        const bf16x8 b1 = mk8(sah, vv), b2 = mk8(sal.x, sal.y, 0u, 0u);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
            f32x4 acc = S[jb];
            const int o = la.tri[jb >> 1] + 4 * (jb & 1);
            const bf16x8 ah = mk8(lds_read_tr16(&B.opnd[4][o]), lds_read_tr16(&B.opnd[6][o]));
            const bf16x8 al = mk8(lds_read_tr16(&B.opnd[5][o]), lds_read_tr16(&B.opnd[7][o]));
            acc = mfma32(ah, b1, acc);
            acc = mfma32(ah, b2, acc);
            acc = mfma32(al, b1, acc);
            acc[0] *= cl.x; acc[1] *= cl.y; acc[2] *= cl.z; acc[3] *= cl.w;
            S[jb] = acc;
            if (ps) {
#pragma unroll
                for (int r = 0; r < 4; ++r) lds.out_s[tix(jb, 4 * g + r) * N + 16 * wt + c16] = acc[r];
            }
        }
        WKV_STAMP(4)
        block_sync_lds();                                        // B
        WKV_STAMP(5)
    }
    send(nchunk - 1);
    if (p.s_final) {
        float* sp = p.s_final + ((size_t)bh * N + 16 * wt + c16) * N;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) *reinterpret_cast<float4*>(sp + tix(jb, 4 * g)) = make_float4(S[jb][0], S[jb][1], S[jb][2], S[jb][3]);
    }
    WKV_STAMP_FLUSH(0, 0, 6)
}

}  // namespace wkv7f4
