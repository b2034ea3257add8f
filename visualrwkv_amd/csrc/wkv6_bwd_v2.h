// WKV6 backward, chunked MFMA form, three-role pipeline -- gfx950.
//
// Same math as bwd6_kernel (wkv6_chunked.h; reference: kernel_backward_111 / kernel_backward_222 of
// VisualRWKV-v6/v6.0/cuda/wkv6_cuda.cu:64-227), rescheduled like the WKV7 backward (wkv7_bwd_v6.h / v8.h).  bwd6_kernel runs a chunk
// as prepare -> barrier -> scores, i-split, j-split, tail -> barrier on FOUR waves per (b, h): at B x H = 256 that is one wave
// per SIMD walking ~1100 dependent instructions per chunk (8.2k cycles, 0.29 of the HBM roofline at B = 4, 0.44 at B = 8 with two
// workgroups per CU; profiles/r4_wkv6_micro.jsonl), with 36 two-byte LDS scatter stores per lane for the transposed operand copies
// and four two-byte global stores per lane for dV.  Here twelve waves per (b, h) work on three consecutive chunks at once:
//
//   P (waves 8-11)  step n, cp = nchunk-1-n: takes the rows of chunk cp from their LDS staging image; requests the rows of chunk
//                   cp-2 and S0 of chunk cp-1 (LDS-DMA, full 128 / 256-byte rows at 16 B per lane, two steps ahead of their
//                   readers; v and dy land directly in the images the I / J waves read); sends the results of step n-1 (gr gk gw
//                   gv images -> full-row 16-byte stores); finishes chunk cp+2 (element-wise tail from the J waves' fp32 results);
//                   prepares the operand images of chunk cp (decay scan, hi/lo images [16][64] bf16, XOR-swizzled: wkv7_bwd_v5.h)
//   I (waves 0-3)   chunk cp+1: A = tril(Rt Kh^T) + diag, dV = A^T dY + Kb dS^T (products whose result leaves the chip are issued
//                   with swapped operands so that lane = token, registers = 4 consecutive channels), dS^T update
//   J (waves 4-7)   chunk cp+1: dA = dY V^T in both orientations, dRe = dY S0, dKb = V dS, dRa = dAl Kh, dKh = dAl^T Rt, the
//                   decay-gradient term sum_i dS S0 c_L, its own copy of dS ([i][j] tiles) and its update
//
// The I and J waves share nothing but the chunk's images (each keeps its own orientation of dL/dS, as bwd6_kernel does), and the
// tail runs a full step behind them: ONE workgroup barrier per step plus one LDS counter (the J waves write their results into the
// tail's single fp32 image right after the barrier; the tail, which comes after the P waves' requests and stores, checks it).
// Operands whose contraction index is the token are fetched with ds_read_b64_tr_b16 from the row-major images: no transposed
// copies.  What the memory side costs was measured by leaving it out (profiles/r4_wkv6_memops.jsonl, B = 4): with 8-byte
// token-per-lane loads and stores one step ahead 0.385 ms, of which the four stores 0.14 (32-byte pieces of a row from four
// different waves) and the loads 0.08; with full rows two steps ahead 0.38 -> the remaining 0.29 ms is instruction issue
// (P ~600, J 280, I 170 instructions per step and SIMD).
// LDS: 2 x 16.5 KB operand images + 4 x 4 KB v / dy + 3 x 8 KB staging + 3 x 16 KB S0 + 16 KB results + 2 x 8 KB output images
// = 154 KB, one workgroup per CU.  Byte offsets of the requests are 32-bit: the launcher sends tensors of 4 GiB and more to bwd6_kernel.
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
using wkv7c::ld8;

enum { RT_H, RT_L, KH_H, KH_L, KB_H, KB_L, RE_H, RE_L, NIMG };
struct Chunk6 {
    uint16_t img[NIMG][IMG];         // Rt Kh Kb Re hi, lo [t][j]
    float cl[N];                     // c_L[j]
    float dpart[4][L];               // per P wave: sum over its 16 key columns of r u k
};
struct Stage6 { uint16_t r[IMG], k[IMG]; float ew[IMG]; };       // rows of a chunk as they lie in memory (swizzled like the images)
struct Lds6V2 {
    Chunk6 b[2];                     // by chunk parity: written by P in step n, read by I / J in step n + 1
    uint16_t vdy[4][2][IMG];         // v, dy [t][i] of chunk c in slot c & 3: LDS-DMA straight into the image the I / J waves read
    Stage6 stg[3];                   // r, k, ew of chunk c in slot c % 3: requested two steps before P prepares it
    float s0[3][N * N];              // chunk-start state s[c] ([j][i] fp32, f32_off swizzle) in slot c % 3, requested two steps before J reads it
    float res[4][IMG];               // dRe dRa dKh dKb of the J waves' chunk, token-per-lane fp32 (f32_off): read by the tail of the NEXT step,
                                     // which hands the image back through flag[0] before the J waves write the next one
    uint16_t out[2][4][IMG];         // gr gk gw (P tail) gv (I waves) of step n in slot n & 1, [t][64] bf16 images: stored as full rows in step n + 1
    unsigned flag[2];                // 0: P waves have read `res` (4 per tail)
    float dd[2][L];                  // diagonal of dA
    float glast[2][N];               // c_L[j] sum_i dS[i][j] S0[i][j]
};
static_assert(sizeof(Lds6V2) <= 160 * 1024, "LDS budget");

#ifndef W6_NOMEM
#define W6_NOMEM 0          // timing experiments: 1 no S0 requests, 2 no row requests, 4 no global stores (results are garbage)
#endif
struct Tail6 { uint2 r, k; float ew[4], e_re[4], e_r[4], e_h[4], e_b[4]; };     // what the tail of a chunk needs from its prepare, two steps later

// Requests of one chunk's rows, full rows at 16 bytes per lane (profiles/r4_mem_role_probe.jsonl: 8-byte token-per-lane loads reach
// 3.7 TB/s on their own, full rows 5+).  A bf16 image [16][64] is two 1 KB instructions (8 token rows x 128 B: lane l = row l >> 3,
// physical slot l & 7 <- logical slot (l & 7) ^ (row & 7): the image's swizzle goes on the SOURCE address), the fp32 rows of ew four
// (4 rows x 256 B, slot ^ (row & 15)).  12 instructions per chunk, wave wq issues 3 wq .. 3 wq + 2.  Rows at and beyond `valid`
// (the last chunk of a sequence whose length is not a multiple of 16) are not requested: their lanes are masked and the caller has
// zero-filled the destination.
DEVFN void dma_rows6(Lds6V2& lds, const Bwd6Args& p, size_t chunk_base, int c, int wq, size_t ts, int lane, int valid) {
    Stage6& S = lds.stg[c % 3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int i = 3 * wq + q;                          // uniform
        if (i < 8) {
            const int arr = i >> 1, half = i & 1;
            const int row = 8 * half + (lane >> 3), slot = (lane & 7) ^ (row & 7);
            const size_t o = chunk_base + (size_t)row * ts + 8 * slot;
            if (row < valid) {
                if (arr == 0) lds_dma16(p.r + o, S.r + half * 8 * N); else if (arr == 1) lds_dma16(p.k + o, S.k + half * 8 * N);
                else if (arr == 2) lds_dma16(p.v + o, lds.vdy[c & 3][0] + half * 8 * N); else lds_dma16(p.gy + o, lds.vdy[c & 3][1] + half * 8 * N);
            }
        } else {
            const int k4 = i - 8;
            const int row = 4 * k4 + (lane >> 4), slot = (lane & 15) ^ (row & 15);
            if (row < valid) lds_dma16(p.ew + chunk_base + (size_t)row * ts + 4 * slot, S.ew + k4 * 4 * N);
        }
    }
}

// The steady-state request path (as in wkv7_bwd_v8.h): the array pointers of the kernel arguments are the scalar bases, the chunk's
// byte offset is added once per step to two per-lane offset registers, no per-request address arithmetic, no exec masks (only the
// last chunk of a sequence can be ragged and it is requested before the loop).  Offsets are 32-bit: the launcher sends tensors of
// 4 GiB and more to bwd6_kernel.  Wave wq of the role: r r k | k v v | dy dy ew | ew ew ew, and S0 rows 16 wq .. 16 wq + 15.
struct Lean6 { unsigned b16a, b16b, eo[3], vs[4]; };
DEVFN Lean6 lean6(int lane, int wq, unsigned ts) {
    Lean6 q;
    const unsigned row = (unsigned)lane >> 3, slot = ((unsigned)lane & 7u) ^ (row & 7u);
    q.b16a = row * ts * 2u + slot * 16u;                 // rows l >> 3 of a bf16 [16][64] chunk: also the address of the full-row stores
    q.b16b = q.b16a + 8u * ts * 2u;                      // rows 8 + (l >> 3)
    const unsigned r4 = (unsigned)lane >> 4, l15 = (unsigned)lane & 15u;
    const unsigned eb = r4 * ts * 4u + ((l15 ^ r4) * 16u);        // fp32 rows: row 4 k4 + r4, slot l15 ^ r4 ^ 4 k4 (ts * 4 is a multiple of 256: the XOR below is safe)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned k4 = wq == 2 ? 0u : (unsigned)i + 1u;
        q.eo[i] = (eb ^ (64u * k4)) + 16u * k4 * ts;
    }
    const unsigned base = r4 * 256u + 16u * (l15 ^ r4);
#pragma unroll
    for (int m = 0; m < 4; ++m) q.vs[m] = base ^ (64u * (unsigned)m);
    return q;
}
struct Bases6 { const void *r, *k, *v, *gy, *ew; };           // the array pointers as SGPR pairs, formed once
template <int WQ>
DEVFN void rows6_lean(Lds6V2& lds, const Bases6& bs, int c3, int c4, unsigned cb16, const Lean6& ll) {
    const unsigned va = ll.b16a + cb16, vb = ll.b16b + cb16;
    Stage6& S = lds.stg[c3];
    const void *pr = bs.r, *pk = bs.k, *pv = bs.v, *pg = bs.gy, *pe = bs.ew;
    if (WQ == 0) {
        const unsigned sr = lds_addr_u32(S.r);
        lds_dma16_lean<0>(pr, va, sr); lds_dma16_lean<0>(pr, vb, sr + 1024u); lds_dma16_lean<0>(pk, va, lds_addr_u32(S.k));
    } else if (WQ == 1) {
        const unsigned sv = lds_addr_u32(lds.vdy[c4][0]);
        lds_dma16_lean<0>(pk, vb, lds_addr_u32(S.k) + 1024u); lds_dma16_lean<0>(pv, va, sv); lds_dma16_lean<0>(pv, vb, sv + 1024u);
    } else if (WQ == 2) {
        const unsigned sd = lds_addr_u32(lds.vdy[c4][1]);
        lds_dma16_lean<0>(pg, va, sd); lds_dma16_lean<0>(pg, vb, sd + 1024u); lds_dma16_lean<0>(pe, ll.eo[0] + 2u * cb16, lds_addr_u32(S.ew));
    } else {
        const unsigned se = lds_addr_u32(S.ew);
#pragma unroll
        for (int i = 0; i < 3; ++i) lds_dma16_lean<0>(pe, ll.eo[i] + 2u * cb16, se + 1024u * (unsigned)(i + 1));
    }
}
template <int WQ>
DEVFN void s0_lean6(float* img, const float* s_chunk, const Lean6& ll) {
    const char* base = reinterpret_cast<const char*>(uniform_ptr(s_chunk)) + WQ * 4096;
    const unsigned d = lds_addr_u32(img) + (unsigned)WQ * 4096u;            // the instruction's immediate offset moves the LDS address as well
    lds_dma16_lean<0>(base, ll.vs[0], d); lds_dma16_lean<1024>(base, ll.vs[1], d);
    lds_dma16_lean<2048>(base, ll.vs[2], d); lds_dma16_lean<3072>(base, ll.vs[3], d);
}

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
    if (tid < 2) lds.flag[tid] = 0u;

    if (role == 2) {
        // ================================================================== P
        // before the first step: the rows of the last two chunks and S0 of the last one (everything later is requested two steps
        // ahead); the last chunk may be ragged -> its destinations are zero-filled first
        {
            const int cl = nchunk - 1, valid = T - cl * L;
            if (valid < L) {                               // only the rows the requests below leave out: nothing orders a ds_write against an LDS-DMA write
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4* a = reinterpret_cast<float4*>(&lds.stg[cl % 3]);
                float4* bq = reinterpret_cast<float4*>(&lds.vdy[cl & 3][0][0]);
                for (int i = (tid & 255); i < 512; i += 256) {            // Stage6: r (128 pieces of 16 B, 8 per row), k (128), ew (256, 16 per row)
                    const int row = i < 256 ? (i & 127) >> 3 : (i - 256) >> 4;
                    if (row >= valid) a[i] = z;
                }
                {
                    const int i = tid & 255;                                // v, dy: 128 pieces each
                    if (((i & 127) >> 3) >= valid) bq[i] = z;
                }
            }
            dma_rows6(lds, p, head_base + (size_t)cl * L * ts, cl, w, ts, lane, valid);
            if (cl >= 1) dma_rows6(lds, p, head_base + (size_t)(cl - 1) * L * ts, cl - 1, w, ts, lane, L);
            dma_state(lds.s0[cl % 3], sbase + (size_t)cl * N * N, 4 * w, 4 * w + 4, lane);
            vmem_drain();
        }
        block_sync_lds();
        float uu[4];
        unpack4(*reinterpret_cast<const uint2*>(p.u + (size_t)hh * N + j0), uu);
        Tail6 qa{}, qb{};                                   // even | odd steps: written by the prepare, read by the tail two steps later
        float gu_acc[4] = {0.f, 0.f, 0.f, 0.f};
        // The results of step n - 1 (gr gk gw of its tail's chunk, gv of the I waves' chunk) leave as FULL 128-byte rows, 16 bytes per
        // lane: wave w sends array w, two instructions of 8 token rows each.  (8-byte token-per-lane stores -- 32-byte pieces of a row
        // from four different waves -- cost 0.14 of this kernel's 0.385 ms at B = 4: profiles/r4_wkv6_memops.jsonl.)
        const Lean6 ll = lean6(lane, w, (unsigned)ts);
        const Bases6 bs{uniform_ptr(p.r), uniform_ptr(p.k), uniform_ptr(p.v), uniform_ptr(p.gy), uniform_ptr(p.ew)};
        // ring slots of chunk cp: cp % 3 kept incrementally (cp falls by one per step)
        int cp3 = (nchunk - 1) % 3;
        auto send_to = [&](uint16_t* arr, int np, int chunk) {
            char* dst = reinterpret_cast<char*>(arr);
            const unsigned cb16 = (unsigned)((head_base + (size_t)chunk * L * ts) * 2u);
            const uint16_t* img = lds.out[np & 1][w] + 8 * lane;                    // row l >> 3 (+ 8), physical slot l & 7: 16 B per lane, linear
            if (!(W6_NOMEM & 4) && chunk * L + (lane >> 3) < T)
                *reinterpret_cast<wkv7v5::u32x4v*>(dst + (ll.b16a + cb16)) = *reinterpret_cast<const wkv7v5::u32x4v*>(img);
            if (!(W6_NOMEM & 4) && chunk * L + 8 + (lane >> 3) < T)
                *reinterpret_cast<wkv7v5::u32x4v*>(dst + (ll.b16b + cb16)) = *reinterpret_cast<const wkv7v5::u32x4v*>(img + 8 * N);
        };
        auto send = [&](int n) {
            const int np = n - 1, chunk = nchunk - np + (w < 3 ? 1 : 0);            // tail: cp + 2, I waves: cp + 1 of step np
            if (np < 1 || chunk < 0 || chunk > nchunk - 1) return;
            // four uniform branches, not a select: the compiler turns a four-way pointer select into a table in scratch memory, and the
            // wait for that load drains every request in flight
            if (w == 0) send_to(p.gr, np, chunk); else if (w == 1) send_to(p.gk, np, chunk);
            else if (w == 2) send_to(p.gw, np, chunk); else send_to(p.gv, np, chunk);
        };
        // one step; the caller alternates qt = qa, qb: no queue shifts
        auto pstep = [&](int n, Tail6& qt) {
            const int cp = nchunk - 1 - n, ct = cp + 2;
            // the rows of chunk cp (landed before the last barrier) -> registers
            uint2 rr = make_uint2(0, 0), kk = rr;
            float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const int m1 = cp3 == 0 ? 2 : cp3 - 1, m2 = m1 == 0 ? 2 : m1 - 1;        // (cp - 1) % 3, (cp - 2) % 3
            if (cp >= 0) {
                const Stage6& S = lds.stg[cp3];
                rr = ld8(&S.r[la.own]); kk = ld8(&S.k[la.own]);
                e4 = *reinterpret_cast<const float4*>(&S.ew[la.f32]);
            }
            // requests: rows of chunk cp - 2, S0 of chunk cp - 1 (both two steps ahead of their readers), the tail's stores after them
            if (!(W6_NOMEM & 2) && cp >= 2) {
                const unsigned cb16 = (unsigned)((head_base + (size_t)(cp - 2) * L * ts) * 2u);
                const int c4 = (cp - 2) & 3;
                if (w == 0) rows6_lean<0>(lds, bs, m2, c4, cb16, ll); else if (w == 1) rows6_lean<1>(lds, bs, m2, c4, cb16, ll);
                else if (w == 2) rows6_lean<2>(lds, bs, m2, c4, cb16, ll); else rows6_lean<3>(lds, bs, m2, c4, cb16, ll);
            }
            if (!(W6_NOMEM & 1) && cp >= 1) {
                float* img = lds.s0[m1];
                const float* sc = sbase + (size_t)(cp - 1) * N * N;
                if (w == 0) s0_lean6<0>(img, sc, ll); else if (w == 1) s0_lean6<1>(img, sc, ll);
                else if (w == 2) s0_lean6<2>(img, sc, ll); else s0_lean6<3>(img, sc, ll);
            }
            send(n);
            // ---------------------------------------------------------------- tail of chunk ct (token c16, channels j0..j0+3)
            if (ct <= nchunk - 1) {
                const int pb = ct & 1;
                if (!(W6_NOMEM & 8)) lds_flag_wait(&lds.flag[0], 4u * (unsigned)(n - 1));       // the J waves wrote their results of the last step right after the barrier
                const float4 x0 = *reinterpret_cast<const float4*>(&lds.res[0][la.f32]);
                const float4 x1 = *reinterpret_cast<const float4*>(&lds.res[1][la.f32]);
                const float4 x2 = *reinterpret_cast<const float4*>(&lds.res[2][la.f32]);
                const float4 x3 = *reinterpret_cast<const float4*>(&lds.res[3][la.f32]);
                const float4 g4 = *reinterpret_cast<const float4*>(&lds.glast[pb][j0]);
                const float dd = lds.dd[pb][c16];
                const float vre[4] = {x0.x, x0.y, x0.z, x0.w}, vra[4] = {x1.x, x1.y, x1.z, x1.w};
                const float vkh[4] = {x2.x, x2.y, x2.z, x2.w}, vkb[4] = {x3.x, x3.y, x3.z, x3.w};
                const float gls[4] = {g4.x, g4.y, g4.z, g4.w};
                float r[4], k[4], gr[4], gk[4], gw[4];
                unpack4(qt.r, r); unpack4(qt.k, k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ddu = dd * uu[e];
                    gr[e] = vre[e] * qt.e_re[e] + vra[e] * qt.e_r[e] + ddu * k[e];
                    gk[e] = vkh[e] * qt.e_h[e] + vkb[e] * qt.e_b[e] + ddu * r[e];
                    gu_acc[e] = fmaf(dd * r[e], k[e], gu_acc[e]);
                    const float pa = vra[e] * (r[e] * qt.e_r[e]);                      // dRa Rt
                    const float pr = vre[e] * (r[e] * qt.e_re[e]) + pa;                // dRe Re + dRa Rt
                    const float ph = vkh[e] * (k[e] * qt.e_h[e]);
                    const float pbb = vkb[e] * (k[e] * qt.e_b[e]);
                    float gx = dpp_shl<1>(pr) - ph - pbb;
                    const float sum_b = group_sum<4>(pbb), sum_m = group_sum<4>(ph - pa);
                    if (c16 == 15) gx += sum_b + gls[e];
                    if (c16 == 7) gx += sum_m;
                    gx += dpp_shl<1>(gx); gx += dpp_shl<2>(gx); gx += dpp_shl<4>(gx); gx += dpp_shl<8>(gx);   // suffix sum over t
                    gw[e] = gx * qt.ew[e];
                }
                st8(&lds.out[n & 1][0][la.own], make_uint2(cvt_pk_bf16(gr[0], gr[1]), cvt_pk_bf16(gr[2], gr[3])));
                st8(&lds.out[n & 1][1][la.own], make_uint2(cvt_pk_bf16(gk[0], gk[1]), cvt_pk_bf16(gk[2], gk[3])));
                st8(&lds.out[n & 1][2][la.own], make_uint2(cvt_pk_bf16(gw[0], gw[1]), cvt_pk_bf16(gw[2], gw[3])));
            }
            // ---------------------------------------------------------------- images of chunk cp
            if (cp >= 0) {
                Chunk6& B = lds.b[cp & 1];
                float r[4], k[4];
                unpack4(rr, r); unpack4(kk, k);
                const float ew[4] = {e4.x, e4.y, e4.z, e4.w};
                const Decay6 d = decay_factors(ew, lane);
                float re[4], rt[4], kh[4], kb[4], dp = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[e] = r[e] * d.e_re[e]; rt[e] = r[e] * d.e_r[e]; kh[e] = k[e] * d.e_h[e]; kb[e] = k[e] * d.e_b[e];
                    dp = fmaf(r[e] * uu[e], k[e], dp);
                    qt.ew[e] = ew[e]; qt.e_re[e] = d.e_re[e]; qt.e_r[e] = d.e_r[e]; qt.e_h[e] = d.e_h[e]; qt.e_b[e] = d.e_b[e];
                }
                qt.r = rr; qt.k = kk;
                dp += lane_xor16(dp);
                dp += lane_xor32(dp);
                if (g == 0) B.dpart[w][c16] = dp;
                uint2 h, l;
                split4(rt, h, l); st8(&B.img[RT_H][la.own], h); st8(&B.img[RT_L][la.own], l);
                split4(kh, h, l); st8(&B.img[KH_H][la.own], h); st8(&B.img[KH_L][la.own], l);
                split4(kb, h, l); st8(&B.img[KB_H][la.own], h); st8(&B.img[KB_L][la.own], l);
                split4(re, h, l); st8(&B.img[RE_H][la.own], h); st8(&B.img[RE_L][la.own], l);
                if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(d.c_l[0], d.c_l[1], d.c_l[2], d.c_l[3]);
            }
            // Everything requested a step ago must have landed; what was issued since then may stay in flight.  In the steady state
            // that is exactly 2 stores (last step) + 7 requests + 2 stores of this step; at the ends of the sequence the counts
            // differ and the wave simply drains.
            if (W6_NOMEM & 16) {} else if (n >= 4 && cp >= 2) vmem_wait<11>(); else vmem_drain();
            cp3 = m1;
            block_sync_lds();
        };
        int n = 0;
        for (; n + 1 < nsteps; n += 2) { pstep(n, qa); pstep(n + 1, qb); }
        if (n < nsteps) pstep(n, qa);
        send(nsteps);                                       // chunk 0's gr gk gw
        // gu[b, h, j] = sum over the tokens of this sample
#pragma unroll
        for (int e = 0; e < 4; ++e) gu_acc[e] = group_sum<4>(gu_acc[e]);
        if (c16 == 0)
            *reinterpret_cast<uint2*>(p.gu + (size_t)blockIdx.x * N + j0) =
                make_uint2(cvt_pk_bf16(gu_acc[0], gu_acc[1]), cvt_pk_bf16(gu_acc[2], gu_acc[3]));
        return;
    }

    block_sync_lds();                                       // the P waves' first requests have landed
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
                const uint2 dyv = lds_read_tr16(&lds.vdy[ci & 3][1][la.trc]);             // dY[4g+e][i]
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
                st8(&lds.out[n & 1][3][la.own], make_uint2(cvt_pk_bf16(dV[0], dV[1]), cvt_pk_bf16(dV[2], dV[3])));
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
    f32x4 dRe = zero4(), dKb = zero4(), dRa = zero4(), dKh = zero4();      // results of a step: written to `res` at the top of the next one
    for (int n = 0; n < nsteps; ++n) {
        const int ci = nchunk - n;
        // Last step's results -> the tail's fp32 image, first thing after the barrier (the tail that read the previous contents ran before
        // it): ONE image, and neither role waits in practice -- the P waves reach their tail after their requests and stores.
        if (n >= 2 && n <= nchunk + 1) {
            *reinterpret_cast<float4*>(&lds.res[0][la.f32]) = make_float4(dRe[0], dRe[1], dRe[2], dRe[3]);
            *reinterpret_cast<float4*>(&lds.res[1][la.f32]) = make_float4(dRa[0], dRa[1], dRa[2], dRa[3]);
            *reinterpret_cast<float4*>(&lds.res[2][la.f32]) = make_float4(dKh[0], dKh[1], dKh[2], dKh[3]);
            *reinterpret_cast<float4*>(&lds.res[3][la.f32]) = make_float4(dKb[0], dKb[1], dKb[2], dKb[3]);
            lds_flag_add(&lds.flag[0]);
        }
        if (ci >= 0 && ci <= nchunk - 1) {
            const Chunk6& B = lds.b[ci & 1];
            const int pb = ci & 1;
            const float clj = B.cl[j];
            f32x4 S0[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float4 x = *reinterpret_cast<const float4*>(&lds.s0[ci % 3][f32_off(j, tix(ib, 4 * g))]);
                S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
            }
            // dA[t][s] = sum_i dY[t][i] V[s][i] in both orientations (exact bf16 operands)
            f32x4 da = dot64<false, false>(lds.vdy[ci & 3][1], nullptr, lds.vdy[ci & 3][0], nullptr, la);        // lane s = c16, registers t = 4g + r
            f32x4 dat = dot64<false, false>(lds.vdy[ci & 3][0], nullptr, lds.vdy[ci & 3][1], nullptr, la);       // lane t = c16, registers s = 4g + r
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
            dRe = zero4(); dKb = zero4();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 dyr = ld16(&lds.vdy[ci & 3][1][la.row[kb]]), vr = ld16(&lds.vdy[ci & 3][0][la.row[kb]]);
                dRe = mfma32(s0h[kb], dyr, dRe);            // dY S0
                dRe = mfma32(s0l[kb], dyr, dRe);
                dKb = mfma32(d2h[kb], vr, dKb);             // V dS
                dKb = mfma32(d2l[kb], vr, dKb);
            }
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
                const uint2 dyi = lds_read_tr16(&lds.vdy[ci & 3][1][la.tri[ib >> 1] + 4 * (ib & 1)]);                         // dY[4g+e][i = tix(ib, c16)]
                dS2[ib] = mfma32(mk8(dyi, dyi), bre, acc);
            }
        }
        block_sync_lds();
    }
}

}  // namespace wkv6v2
