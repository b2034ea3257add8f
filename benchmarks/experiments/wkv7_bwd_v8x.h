// EXPERIMENT COPY of csrc/wkv7_bwd_v8.h (namespace wkv7v8x): the same kernel WITH every timing / layout knob rounds 4 - 6 used -- role skips, the tail
// on the J waves (JTAIL), accumulation chains, the issue-cost probe, and round 6's OPT bits (dealt tile-pair reads, swizzled dS image, S0 prefetched
// into the J waves' registers, cache policies of the requests, non-temporal tail stores, no-store / no-S0 timing builds).  The product header carries
// none of them.  Built only by benchmarks/build_alt.sh (through wkv7_experiments.h) and by the host emulator's tests.
//
// WKV7 backward, chunked MFMA form, fifth-generation schedule -- gfx950.
//
// Same math as wkv7_bwd_v5.h (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:54-130), the three roles and the full-row
// LDS-DMA memory role of wkv7_bwd_v7.h.  The phase stamps of v6 / v7 (profiles/r4_wkv7_phases_b16.json) show every role stretched
// by the other two on its SIMD -- the J waves' 130-instruction operand split takes 1.2-2.2k cycles beside the P waves, their 20
// output MFMAs 1.2-2k -- with the VALU port ~60 % busy (profiles/r3_wkv7_pmc_b16.txt: 273 M VALU instructions, 4.1 cycles
// each) and the J role and I wave 0 (scores + T chain, THEN its i-split) last at the barrier.  Three changes take work off
// those two:
//   * ONE copy of dL/dS.  v5 - v7 keep dS twice (i-split tiles in the I waves, j-split tiles in the J waves: the tile-level
//     analogue of the reference's dstate / dstateT pair) and update both: 2 x 12 MFMAs, 2 x 64 VALU of scaling and hi/lo splits
//     per wave pair and step.  Here only the I waves hold dS.  They already form hi/lo operands of diag(c_L) dS^T for dSA / dV;
//     those 8 registers x 2 go to a [i][j] bf16 image (4 x ds_write_b128 per lane), and the J waves -- one step later --
//     fetch THEIR operands of the same matrix (k = i along the image's rows) with ds_read_b64_tr_b16: no second accumulator,
//     no second split, no second update.  The decay-gradient term sum_i dS[i][j] S_L[i][j] becomes the diagonal of
//     (c_L dS)^T S_L / c_L: 6 MFMAs on operands the J waves hold anyway (the S0 operands of the previous step).
//   * The T chain ((I - M_za)^-1 by nilpotent doubling: 28 dependent MFMA / split stages) runs on P wave 0, which has >2k
//     cycles of slack per step in v7 and whose registers the DMA memory role freed; it only needs the images the P waves built
//     a step earlier.  The four I waves share the seven score / score-gradient pieces and all start their i-split at once.
//   * S0 is single buffered (the P waves request the next one when the J waves have lifted theirs into registers: the same
//     counter that releases the tail), which pays for the dS image: LDS 157 KB.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_chunked.h>
#ifndef VRWKV_EMULATED_PRIMS
// 16 bytes per lane from uniform_base + lane_byte_off + IMM INTO the registers of `dst` (a read-write operand: the value stays in one physical register
// tuple from the request to its use iterations later -- no copies the compiler would have to wait for).  The compiler does not track this load:
// the consumer calls vmem_wait_for before it touches `dst`.
// NOPS: wait states in front of the request -- the hazard recogniser does not look into inline asm, and a scalar base that a VALU instruction
// (v_readfirstlane) has just written needs 5 wait states before a vector-memory instruction reads it
template <int IMM, int NOPS = 0> DEVFN void global_load16_inplace(f32x4& dst, const void* uniform_base, unsigned lane_byte_off) {
    if (NOPS > 0) asm volatile("s_nop %4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(lane_byte_off), "s"(uniform_base), "n"(IMM), "n"(NOPS - 1) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(lane_byte_off), "s"(uniform_base), "n"(IMM) : "memory");
}
// an empty statement that inline-asm requests issued after it cannot be moved above, and that needs a .. d computed: keeps the LAST USE of a register
// tuple (whatever a .. d were computed from) in front of the request that refills it
DEVFN void order_after(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
// s_waitcnt vmcnt(N) that the uses of a .. d cannot be scheduled above
template <int N_> DEVFN void vmem_wait_for(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N_) : "memory");
}
template <int IMM, int CPOL = 0> DEVFN void lds_dma16_lean_cp(const void* uniform_base, unsigned lane_byte_off, unsigned lds_dst_uniform) {
    if (CPOL == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 nt"
                                :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform), "n"(IMM) : "memory", "m0");
    else if (CPOL == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 sc1"
                                :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform), "n"(IMM) : "memory", "m0");
    else if (CPOL == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 sc0 sc1 nt"
                                :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform), "n"(IMM) : "memory", "m0");
    else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform), "n"(IMM) : "memory", "m0");
}
// one dword per lane into a scratch line of LDS: a request whose only purpose is to bring the lane's 128-byte line into L2 (OPT & 65536)
DEVFN void lds_dma4_touch(const void* uniform_base, unsigned lane_byte_off, unsigned lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform) : "memory", "m0");
}
DEVFN void sleep_640_cycles() { __builtin_amdgcn_s_sleep(10); }
#else
DEVFN void lds_dma4_touch(const void*, unsigned, unsigned) {}
DEVFN void sleep_640_cycles() {}
#endif
#include <wkv7_bwd_rows.h>     // ChunkImg7, RawP, DmaLane, dma_lane, prep7, dscores-style helpers; through it v6 / v5 building blocks

#ifndef VRWKV_V8_OPT
#define VRWKV_V8_OPT 0
#endif
#ifndef VRWKV_V8_ROTATE
#define VRWKV_V8_ROTATE 1
#endif
#ifndef VRWKV_V8_DM_FIRST
#define VRWKV_V8_DM_FIRST 1
#endif
#ifndef VRWKV_V8_PI
#define VRWKV_V8_PI 0
#endif
#ifndef VRWKV_V8_PJ
#define VRWKV_V8_PJ 0
#endif
#ifndef VRWKV_V8_PP
#define VRWKV_V8_PP 1
#endif
#ifndef VRWKV_PROF_WAVE
#define VRWKV_PROF_WAVE 0       // which wave of each role the PROF instantiation stamps (0 .. 3)
#endif
#ifndef VRWKV_V8_SCORES_ON_J
#define VRWKV_V8_SCORES_ON_J 0
#endif
#ifndef VRWKV_V8_CHAINS
#define VRWKV_V8_CHAINS 0
#endif
// issue-cost probe (experiment builds only): N extra scalar / vector / wait instructions per step in the J waves (role 1) or the P waves (role 2)
#ifndef VRWKV_V8_DUMMY_N
#define VRWKV_V8_DUMMY_N 0
#endif
#ifndef VRWKV_V8_DUMMY_KIND
#define VRWKV_V8_DUMMY_KIND 0       // 0: s_mov_b32   1: v_mov_b32   2: s_nop 0
#endif
#ifndef VRWKV_V8_DUMMY_ROLE
#define VRWKV_V8_DUMMY_ROLE 1
#endif

namespace wkv7v8x {

template <int ROLE>
DEVFN void dummy_issue() {
#if VRWKV_V8_DUMMY_N > 0
    if (ROLE != VRWKV_V8_DUMMY_ROLE) return;
#pragma unroll
    for (int i = 0; i < VRWKV_V8_DUMMY_N; ++i) {
        if (VRWKV_V8_DUMMY_KIND == 0) { unsigned t; asm volatile("s_mov_b32 %0, 0" : "=s"(t)); }
        else if (VRWKV_V8_DUMMY_KIND == 1) { unsigned t; asm volatile("v_mov_b32 %0, 0" : "=v"(t)); }
        else asm volatile("s_nop 0");
    }
#endif
}

using wkv7::BwdArgs;
using namespace wkv7c;
using namespace wkv7v5;      // IMG, HLI, img_off, f32_off, LaneAddr, lane_addr, ld16, st16, mfma32, dot64, mask_split, tiles_op, dma_state
using wkv7v6::Decay;
using wkv7v6::decay_scan;
using wkv7v6::TailRaw;
using wkv7v6::BoolTag;

using wkv7v7::ChunkImg7;
using wkv7v7::RawP;
using wkv7v7::DmaLane;
using wkv7v7::dma_lane;
using wkv7v7::prep7;
using wkv7v7::dma_chunk;
using wkv7v7::read_stage;
using wkv7v7::tail7;
using wkv7v7::dscores7;
constexpr int SIMG = N * N;           // elements of a [64][64] image
// element offset in the [64][64] bf16 dS image: img_off, or (SWZ) with bit 3 of the row folded into the slot swizzle
template <bool SWZ> DEVFN int dsi_off(int row, int col) {
    return SWZ ? row * 64 + ((((col >> 3) ^ (row & 7) ^ ((row >> 1) & 4)) << 3) | (col & 7)) : img_off(row, col);
}
struct LdsV8 {
    ChunkImg7 b[3];
    uint16_t vdy[4][2][IMG];         // V, dY [t][i] of chunk c in slot c & 3, written by LDS-DMA (swizzled like every image)
    uint16_t stg[5][IMG];            // w q k z a of the chunk the P waves prepare next (LDS-DMA; read by the P waves only)
    float stg_sa[IMG];               // sa of that chunk, fp32, f32_off swizzle
    uint16_t dz[2][IMG];             // "DZ" images of M_zk and T^T (I waves, same step)
    uint16_t sc[2][HLI];             // M_qa, M_qk pair images (I waves, same step)
    uint16_t dsc[4][HLI];            // score gradients of the J waves' chunk (I waves 1-3 -> J waves, same step)
    uint16_t dr[2][2][IMG];          // dR hi, lo [t][i] by chunk parity (I waves -> next step's dM and j-split)
    float s0[N * N];                 // S0 of the J waves' next chunk (P waves' LDS-DMA, requested when flag 3 says the current one is in registers)
    uint16_t dsi[2][SIMG];           // hi, lo of diag(c_L) dS as the I waves hold it at the start of their step: [i][j] bf16, swizzled like
                                     // every image (I waves -> the J waves' operands one step later; flag 3 hands it back)
    union {
        float res[4][IMG];           // J -> P: dZt dQt dAh dKh before the decay factors, fp32 (single: flag 4 hands it back)
        float x2r[3][IMG];           // JTAIL: log2 c_t [t][j] of chunk c in slot c % 3 (P waves -> the J waves' tail two steps later)
    };
    float glast[2][N];               // sum_i dS_L[i][j] S_L[i][j] at the chunk's last token, by chunk parity
    unsigned touch[2][64];           // OPT & 65536: landing scratch of the L2 touches (never read)
    unsigned flag[8];                // 0: M_qa, M_qk, M_zk written (3 per step)  1: dM written (3)  2: T written (1)  3: J operands split (4)
                                     // 4: tail has read `res` (4)  5: P waves hold their staging pieces (4)      (flag 1: four I waves since the score-gradient pieces were re-dealt)
};
static_assert(sizeof(LdsV8) <= 160 * 1024, "LDS budget");

// ------------------------------------------------------------------------------------------ P waves 1-3: lean request issue
// The steady-state steps issue their requests with the scalar-base form and NO per-request address arithmetic: the array base
// pointers (kernel arguments) are the scalar bases, the chunk's byte offset is added once per step to three or four per-lane
// offset registers, and the 16 x 1 KB of S0 use four loop-invariant lane offsets + the instruction's immediate.  (The generic
// dma_chunk / dma_state cost ~8-10 scalar and vector instructions per request -- pointer selects, 64-bit adds, M0 save / restore --
// and every instruction of any class takes an issue slot: profiles/r4_wkv7_pmc_v6_v8.txt.)  Offsets are 32-bit: the launcher
// sends tensors of 4 GiB and more to wkv7_bwd_v6.h.
struct LeanLane { unsigned b16, b16h, sa0, sa1, vs[4]; };
DEVFN LeanLane lean_lane(int lane, int wi, unsigned ts) {
    const DmaLane dl = dma_lane(lane, ts);
    LeanLane ll;
    ll.b16 = dl.b16; ll.b16h = dl.b16 + 8u * ts * 2u;
    const unsigned qa = wi == 0 ? 1u : wi == 1 ? 2u : 0u;             // the sa quarters of this wave: i = wi + 3k in {14 .. 17}
    ll.sa0 = (dl.f32 ^ (64u * qa)) + qa * 4u * ts * 4u;
    ll.sa1 = (dl.f32 ^ (64u * 3u)) + 3u * 4u * ts * 4u;               // wave 3 (wi = 2) also has quarter 3
    const unsigned r4 = (unsigned)lane >> 4, base = r4 * 256u + 16u * (((unsigned)lane & 15u) ^ r4);
#pragma unroll
    for (int m = 0; m < 4; ++m) ll.vs[m] = base ^ (64u * (unsigned)m);          // row 4k + r4 of the S0 image: source slot ^ (row & 15)
    return ll;
}
template <int WI, int CP = 0>
DEVFN void rows_lean(LdsV8& lds, const BwdArgs& p, int c, unsigned cb16, const LeanLane& ll) {
    const unsigned v16 = ll.b16 + cb16, v16h = ll.b16h + cb16, vsa0 = ll.sa0 + 2u * cb16, vsa1 = ll.sa1 + 2u * cb16;
    const unsigned vd = lds_addr_u32(lds.vdy[c & 3][0]);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int i = WI + 3 * k;                          // compile-time after unrolling
        if (i < 14) {
            const int arr = i >> 1, half = i & 1;
            const uint16_t* src = arr == 0 ? p.w : arr == 1 ? p.q : arr == 2 ? p.k : arr == 3 ? p.z : arr == 4 ? p.a : arr == 5 ? p.v : p.dy;
            const unsigned dst = (arr < 5 ? lds_addr_u32(lds.stg[arr]) : vd + (unsigned)(arr - 5) * IMG * 2u) + (unsigned)half * 8u * N * 2u;
            lds_dma16_lean_cp<0, CP>(src, half ? v16h : v16, dst);
        } else {
            const int qd = i - 14;
            lds_dma16_lean_cp<0, CP>(p.sa, (WI == 2 && qd == 3) ? vsa1 : vsa0, lds_addr_u32(lds.stg_sa) + (unsigned)qd * 4u * N * 4u);
        }
    }
}
template <int WI, int CP = 0>
DEVFN void s0_lean(LdsV8& lds, const float* s_chunk, const LeanLane& ll) {
    constexpr int K0 = WI == 0 ? 0 : WI == 1 ? 5 : 10, K1 = WI == 0 ? 5 : WI == 1 ? 10 : 16;
    const unsigned dst = lds_addr_u32(lds.s0);
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const char* base = reinterpret_cast<const char*>(s_chunk) + (k >> 2) * 4096;
        const unsigned d = dst + (unsigned)(k >> 2) * 4096u;      // the instruction's immediate offset moves the LDS address as well
        switch (k & 3) {
            case 0: lds_dma16_lean_cp<0, CP>(base, ll.vs[0], d); break;
            case 1: lds_dma16_lean_cp<1024, CP>(base, ll.vs[1], d); break;
            case 2: lds_dma16_lean_cp<2048, CP>(base, ll.vs[2], d); break;
            default: lds_dma16_lean_cp<3072, CP>(base, ll.vs[3], d); break;
        }
    }
}

// ------------------------------------------------------------------------------------------ tail with the prepare's decay factors
// VRWKV_V8_TAILQ: the queue entry of a chunk also carries c_t and 1 / c_t (8 more registers per entry, three entries) so that the tail,
// three steps later, does not form them again (8 v_exp_f32 -- quarter rate -- and their DPP moves per lane and step)
#ifndef VRWKV_V8_TAILQ
#define VRWKV_V8_TAILQ 0
#endif
struct TailQ { uint2 q, k, z, a; float x2[4]; float cc[VRWKV_V8_TAILQ ? 4 : 1], ic[VRWKV_V8_TAILQ ? 4 : 1]; };
struct Prep8 { float x2[4], cc[4], ic[4]; };
DEVFN Prep8 prep8(ChunkImg7& B, const RawP& raw, int c16, int j0, const LaneAddr& la) {
    float q[4], k[4], z[4], a[4];
    unpack4(raw.q, q); unpack4(raw.k, k); unpack4(raw.z, z); unpack4(raw.a, a);
    const Decay d = decay_scan(raw.w);
    Prep8 o;
    float zt[4], qt[4], ah[4], kh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float cc = fast_exp2(d.x2[e]), ic = fast_exp2(-d.x2[e]);
        const float cp = dpp_shr1_fill(cc, 1.f);
        zt[e] = z[e] * cp; qt[e] = q[e] * cc; ah[e] = a[e] * ic; kh[e] = k[e] * ic;
        o.x2[e] = d.x2[e]; o.cc[e] = cc; o.ic[e] = ic;
    }
    uint2 hh, ll;
    split4(zt, hh, ll); st8(&B.opnd[0][la.own], hh); st8(&B.opnd[1][la.own], ll);
    split4(qt, hh, ll); st8(&B.opnd[2][la.own], hh); st8(&B.opnd[3][la.own], ll);
    split4(ah, hh, ll); st8(&B.opnd[4][la.own], hh); st8(&B.opnd[5][la.own], ll);
    split4(kh, hh, ll); st8(&B.opnd[6][la.own], hh); st8(&B.opnd[7][la.own], ll);
    const float sav[4] = {raw.sa.x, raw.sa.y, raw.sa.z, raw.sa.w};
    split4(sav, hh, ll); st8(&B.sa[0][la.own], hh); st8(&B.sa[1][la.own], ll);
    if (c16 == 15) *reinterpret_cast<float4*>(&B.cl[j0]) = make_float4(o.cc[0], o.cc[1], o.cc[2], o.cc[3]);
    return o;
}
template <bool NOSTORE = false, bool NTST = false, bool WIDE = false>
DEVFN void tail8(LdsV8& lds, int par, const TailQ& tr, const BwdArgs& p, size_t u, unsigned lane_boff, int c16, int pw, int g, const LaneAddr& la) {
    const float4 zt4 = *reinterpret_cast<const float4*>(&lds.res[0][la.f32]);
    const float4 qt4 = *reinterpret_cast<const float4*>(&lds.res[1][la.f32]);
    const float4 ah4 = *reinterpret_cast<const float4*>(&lds.res[2][la.f32]);
    const float4 kh4 = *reinterpret_cast<const float4*>(&lds.res[3][la.f32]);
    const float4 gl4 = *reinterpret_cast<const float4*>(&lds.glast[par][16 * pw + 4 * g]);
    if (!WIDE) lds_flag_add(&lds.flag[4]);                // (waits for the reads above) the J waves may overwrite `res`  (WIDE: after the read-back)
    const float dZt[4] = {zt4.x, zt4.y, zt4.z, zt4.w}, dQt[4] = {qt4.x, qt4.y, qt4.z, qt4.w};
    const float dAh[4] = {ah4.x, ah4.y, ah4.z, ah4.w}, dKh[4] = {kh4.x, kh4.y, kh4.z, kh4.w};
    const float glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
    float q[4], k[4], z[4], a[4];
    unpack4(tr.q, q); unpack4(tr.k, k); unpack4(tr.z, z); unpack4(tr.a, a);
    float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x2 = tr.x2[e], l2 = x2 - dpp_shr1_fill(x2, 0.f);      // log2 c_t from the queue; log2 w_t = its difference along t
        const float cc = VRWKV_V8_TAILQ ? tr.cc[VRWKV_V8_TAILQ ? e : 0] : fast_exp2(x2), ic = VRWKV_V8_TAILQ ? tr.ic[VRWKV_V8_TAILQ ? e : 0] : fast_exp2(-x2);
        const float cp = dpp_shr1_fill(cc, 1.f);
        dz[e] = dZt[e] * cp; dq[e] = dQt[e] * cc; da[e] = dAh[e] * ic; dk[e] = dKh[e] * ic;
        float gt = dq[e] * q[e] - da[e] * a[e] - dk[e] * k[e] + dpp_shl<1>(dz[e] * z[e]);
        if (c16 == 15) gt += glv[e];
        gt += dpp_shl<1>(gt); gt += dpp_shl<2>(gt); gt += dpp_shl<4>(gt); gt += dpp_shl<8>(gt);   // suffix sum over t
        dw[e] = gt * (l2 * LN2);
    }
    auto out = [&](uint16_t* base) { return reinterpret_cast<uint2*>(reinterpret_cast<char*>(base + u) + lane_boff); };   // uniform base + lane offset
    if (NOSTORE) {          // timing experiment: the results stay "used" (an LDS write nobody reads) but never leave the chip
        lds.glast[par][16 * pw + 4 * g] = dw[0] + dq[1] + dk[2] + dz[3] + da[0];
        return;
    }
    if (WIDE) {
        // OPT & 8192, full-row stores: the five 8-byte results go back INTO the 16-byte slots of `res` this lane has just read (its own bytes: no hazard
        // with the other P waves' reads) -- array k = 0 .. 3 in half (c16 & 1) of the slot in res[k], the fifth in the other half of the slot in res[0] --
        // and wide_store() sends them out as whole 128-byte token rows once all four P waves have written (flag 7).  Token parity picks the half so that
        // the read-back's 32-lane groups (4 tokens x 8 octets) spread over both halves.
        const int h8 = 2 * (c16 & 1);                                   // in floats
        *reinterpret_cast<uint2*>(&lds.res[0][la.f32 + h8]) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
        *reinterpret_cast<uint2*>(&lds.res[1][la.f32 + h8]) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
        *reinterpret_cast<uint2*>(&lds.res[2][la.f32 + h8]) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
        *reinterpret_cast<uint2*>(&lds.res[3][la.f32 + h8]) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
        *reinterpret_cast<uint2*>(&lds.res[0][la.f32 + 2 - h8]) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
        lds_flag_add(&lds.flag[7]);
        return;
    }
    if (NTST) {
        typedef unsigned long long u64_;
        auto pk = [](uint32_t a, uint32_t b) { return (u64_)a | ((u64_)b << 32); };
        __builtin_nontemporal_store(pk(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3])), reinterpret_cast<u64_*>(out(p.dw)));
        __builtin_nontemporal_store(pk(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3])), reinterpret_cast<u64_*>(out(p.dq)));
        __builtin_nontemporal_store(pk(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3])), reinterpret_cast<u64_*>(out(p.dk)));
        __builtin_nontemporal_store(pk(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3])), reinterpret_cast<u64_*>(out(p.dz)));
        __builtin_nontemporal_store(pk(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3])), reinterpret_cast<u64_*>(out(p.da)));
        return;
    }
    *out(p.dw) = make_uint2(cvt_pk_bf16(dw[0], dw[1]), cvt_pk_bf16(dw[2], dw[3]));
    *out(p.dq) = make_uint2(cvt_pk_bf16(dq[0], dq[1]), cvt_pk_bf16(dq[2], dq[3]));
    *out(p.dk) = make_uint2(cvt_pk_bf16(dk[0], dk[1]), cvt_pk_bf16(dk[2], dk[3]));
    *out(p.dz) = make_uint2(cvt_pk_bf16(dz[0], dz[1]), cvt_pk_bf16(dz[2], dz[3]));
    *out(p.da) = make_uint2(cvt_pk_bf16(da[0], da[1]), cvt_pk_bf16(da[2], da[3]));
}

// OPT & 8192: the read-back of tail8<WIDE>: 10 units of 8 token rows x 128 B (array k = unit >> 1, token half = unit & 1), units w, w + 4, w + 8 of P wave w;
// lane = (token 8 half + (lane >> 3), octet lane & 7): two 8-byte pieces (4-channel groups 2o, 2o + 1) from the slots their owners wrote, one 16-byte store
DEVFN void wide_store(LdsV8& lds, const BwdArgs& p, size_t u, unsigned ts, int lane, int w) {
    uint16_t* const outs[5] = {p.dw, p.dq, p.dk, p.dz, p.da};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int unit = w + 4 * i;
        if (unit < 10) {
            const int k = unit >> 1, t = 8 * (unit & 1) + (lane >> 3), o = lane & 7;
            const int half = k < 4 ? (t & 1) : 1 - (t & 1);
            const float* img = lds.res[k < 4 ? k : 0];
            const uint2 a = *reinterpret_cast<const uint2*>(&img[f32_off(t, 8 * o) + 2 * half]);
            const uint2 b = *reinterpret_cast<const uint2*>(&img[f32_off(t, 8 * o + 4) + 2 * half]);
            typedef uint32_t u4_ __attribute__((ext_vector_type(4)));
            const u4_ v = {a.x, a.y, b.x, b.y};
            *reinterpret_cast<u4_*>(outs[k] + u + (size_t)t * ts + 8 * o) = v;
        }
    }
}

// The same tail on the J waves (JTAIL): dZt dQt dAh dKh stay in registers (lane = token c16, 4 channels 16w + 4g ..), the operands of
// the decay-gradient integrand come from the chunk's operand images (hi + lo; this lane's own 8-byte pieces, the layout the P waves wrote),
// log2 c_t from the ring.  The tail of a chunk runs ONE STEP LATER, between the J waves' own matrix-core phase and their wait for the score
// gradients (where they stood 0.6-0.9k cycles per step): run right after the chunk's products it lengthened the J waves' dependent chain
// and the step by 9 % (profiles/r5a_wkv7_ab.jsonl).  The images of a chunk are overwritten in the next step, so the inputs are lifted
// into registers (JTailIn, 36 registers with the results) at the end of the chunk's own step.
struct JTailIn { uint2 o[8]; float4 x2; f32x4 dZt, dQt, dAh, dKh; };
DEVFN void jtail_lift(JTailIn& t, const LdsV8& lds, const ChunkImg7& B, int cj, const f32x4& dZt, const f32x4& dQt, const f32x4& dAh, const f32x4& dKh, const LaneAddr& la) {
#pragma unroll
    for (int i = 0; i < 8; ++i) t.o[i] = ld8(&B.opnd[i][la.own]);
    t.x2 = *reinterpret_cast<const float4*>(&lds.x2r[cj % 3][la.f32]);
    t.dZt = dZt; t.dQt = dQt; t.dAh = dAh; t.dKh = dKh;
}
DEVFN void jtail_run(const JTailIn& t, const LdsV8& lds, int cj, const BwdArgs& p, size_t u, unsigned lane_boff, int c16, int w, int g) {
    float zh[4], zl[4], qh[4], ql[4], ah[4], al[4], kh[4], kl[4];
    unpack4(t.o[0], zh); unpack4(t.o[1], zl); unpack4(t.o[2], qh); unpack4(t.o[3], ql);
    unpack4(t.o[4], ah); unpack4(t.o[5], al); unpack4(t.o[6], kh); unpack4(t.o[7], kl);
    const float4 gl4 = *reinterpret_cast<const float4*>(&lds.glast[cj & 1][16 * w + 4 * g]);      // written by this wave a step ago
    const float x2v[4] = {t.x2.x, t.x2.y, t.x2.z, t.x2.w}, glv[4] = {gl4.x, gl4.y, gl4.z, gl4.w};
    float dz[4], dq[4], da[4], dk[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x2 = x2v[e], l2 = x2 - dpp_shr1_fill(x2, 0.f);        // log2 w_t = the difference of log2 c_t along t
        const float cc = fast_exp2(x2), ic = fast_exp2(-x2);
        const float cp = dpp_shr1_fill(cc, 1.f);
        dz[e] = t.dZt[e] * cp; dq[e] = t.dQt[e] * cc; da[e] = t.dAh[e] * ic; dk[e] = t.dKh[e] * ic;
        // dq q - da a - dk k + (dz z)(t+1) with q = Qt / c_t etc.: the decay factors cancel
        float gt = t.dQt[e] * (qh[e] + ql[e]) - t.dAh[e] * (ah[e] + al[e]) - t.dKh[e] * (kh[e] + kl[e]) + dpp_shl<1>(t.dZt[e] * (zh[e] + zl[e]));
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

// ------------------------------------------------------------------------------------------ kernel
// dbg (PROF): as wkv7_bwd_v6.h.  SKIP (timing experiments only, results are garbage): bit 0 P does nothing, bit 1 I only raises
// its flags, bit 2 J does nothing.
// TBF16: the doubling chain of T on the bf16 matrix core with split operands (2 MFMAs + 2 splits per product) or on the f32 one (4 MFMAs of
// twice the pipe time, no VALU work)
// PT: priority of P wave 0 while it runs the T chain (back to PP afterwards)
// AHEAD: the three score pieces (M_qa, M_qk, M_zk) of a chunk are formed by P waves 1-3 at the END of the step in which the chunk's
// images are built -- where those waves stood ~1.9k cycles at the barrier (profiles/r4b_wkv7_phases_waves_v8.jsonl) -- instead of by I
// waves 1-3 at the start of the next step, where they delayed the i-split's chain by 1.6-1.8k cycles.  The P waves prepare the
// images BEFORE their tail for that (the queue entry of the prepare is assigned after the tail has consumed the old one).
// JTAIL (variant 10): the element-wise tail and the five gradient stores run on the J waves, in the step in which they form dZt dQt dAh dKh
// -- the results never leave their registers -- instead of on the P waves a step later.  One wave issues an instruction every ~5 cycles
// whatever the other two waves of its SIMD do (profiles/r3_valu_rate.json: 5.5 cycles per VALU instruction at one wave per SIMD, 2.1 at
// three), so a step lasts as long as its LONGEST wave: per step the P waves issued ~660 (wave 0, with the T chain) / ~545 instructions,
// the I waves ~445, the J waves ~225 (ISA of the AHEAD instantiation), and the P waves reached the barrier last with 0.2-0.3k cycles of
// slack (profiles/r4b_wkv7_phases_waves_v8_ahead.jsonl).  The tail is ~165 of them.  On the J waves it needs Zt Qt Ah Kh of the chunk
// (hi + lo from the operand images: dq q = dQt Qt etc., so the raw inputs are not needed) and log2 c_t (the P waves leave it in a
// ring of three fp32 images that takes the place of `res`): no three-deep register queue on the P waves, no `res` round trip, no flag 4.
// OPT (bit mask): 1 = the dS update's transposing reads of tile pairs are dealt so that an instruction touches both halves of its 16-byte slots
// (rows 4g, 4g+1 of one tile and rows 4g+2, 4g+3 of its neighbour: each of the two reads of a pair returns half of either tile, re-paired in
// registers for free) -- as two reads of the SAME half they were 2-way bank conflicts by construction (benchmarks/lds_conflicts.py: 128 of the
// workgroup's 536 conflict cycles per step); 2 = the dS image for the J waves is swizzled with (row & 7) ^ ((row >> 1) & 4), which keeps the
// I waves' 16-byte writes conflict-free and makes the J waves' transposing reads of rows r and r + 8 land on different banks (64 cycles per step);
// 64 / 128 (timing experiments only, results are garbage): no tail stores / no S0 requests
template <bool PROF, int PI = VRWKV_V8_PI, int PJ = VRWKV_V8_PJ, int PP = VRWKV_V8_PP, int SKIP = 0, bool TBF16 = true, int PT = PP, bool AHEAD = false, bool JTAIL = false, int OPT = VRWKV_V8_OPT>
__global__ __launch_bounds__(768) void bwd_kernel_v8(BwdArgs p) {
    LdsV8& lds = *reinterpret_cast<LdsV8*>(dyn_lds());
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
    const int nsteps = nchunk + (JTAIL ? 2 : 3);       // the last step of the P-tail schedule holds only the tail of chunk 0
    const LaneAddr la = lane_addr(c16, g, w);
    const unsigned out_off = (unsigned)c16 * ts + 16u * w + 4u * g;        // token c16, channels 16w+4g..+3
    WKV_STAMP_DECL
    const unsigned long long rt0_ = PROF ? realtime64_() : 0ull;

    if (OPT & 131072) {                                 // stagger the workgroups: (bh & 7) eighths of a step (~0.33 us each) before anything is requested
        for (unsigned i = 0; i < (bh & 7u); ++i) sleep_640_cycles();
    }
    if (tid < 8) lds.flag[tid] = 0u;
    if (role == 2 && !(SKIP & 1)) {                     // rows of the last chunk: staging + its V / dY slot
        const DmaLane dl = dma_lane(lane, ts);
        dma_chunk(lds, p, head_base + (size_t)(nchunk - 1) * L * ts, nchunk - 1, w, ts, dl);
        vmem_drain();
    }
    block_sync_lds();

    if (role == 2) {
        // ================================================================== P: images of chunk cp, T of chunk cp + 1, tail of chunk cp + 3
        wave_priority<PP>();
        TailQ q0{}, q1{}, q2{};                             // inputs of chunks cp+1, cp+2, cp+3 at the top of a step
        const unsigned lane_boff = out_off * 2u;
        const DmaLane dl = dma_lane(lane, ts);
        const LeanLane ll = lean_lane(lane, w > 0 ? w - 1 : 0, ts);
        unsigned n_ps = 0, n_wide = 0;
        // One step.  FULL (steps 3 .. nchunk-2: a tail, a prep, a T, a non-empty S0 and a next chunk every time) has no
        // conditions: every path issues [6 row DMAs, 5-6 S0 DMAs (waves 1-3), 5 tail stores] in this order, so the wait before the
        // barrier is vmcnt(5): everything the other roles will read has landed, the stores stay in flight.
        // qt: the queue entry the tail consumes.  With SHIFT the three entries move up by one afterwards (q2 <- q1 <- q0 <- new);
        // without, the new entry replaces the consumed one in place and the CALLER rotates the names (steady state, in threes:
        // the 24 register moves of the shift are a twentieth of this role's instructions)
        auto pstep = [&](int n, auto full_tag, TailQ& qt, auto shift_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr bool SHIFT = decltype(shift_tag)::value;
            const int cp = nchunk - 1 - n, cd = cp + 1, ct = cp + 3;      // images | the I waves' chunk: T now, S0 for the J waves' next step | tail
            WKV_STAMP(4)
            if (!(SKIP & 1)) {
                dummy_issue<2>();
                RawP raw;
                const bool do_prep = FULL || cp >= 0;
                if (do_prep) {
                    raw = read_stage(lds, la);
                    lds_flag_add(&lds.flag[5]);             // (waits for the reads) ...
                    n_ps += 4u;
                    if (w > 0) lds_flag_wait(&lds.flag[5], n_ps);      // ... all four P waves hold their pieces: the staging bytes are free
                }
                // OPT & 65536: S0 of the NEXT step's chunk is touched now (one dword of each of its 128 lines, waves 1 and 2: one request each), so that
                // the 16 KB requested after flag 3 of the next step come from L2 instead of HBM.  First requests of the step: the counted wait at its
                // end is unchanged.
                if ((OPT & 65536) && FULL && (w == 1 || w == 2) && cd >= 2)
                    lds_dma4_touch(sbase + (size_t)(cd - 2) * N * N, (unsigned)(lane + 64 * (w - 1)) * 128u, lds_addr_u32(lds.touch[w - 1]));
                // waves 1-3 request the rows of the next chunk (6 instructions each); wave 0 has the T chain instead
                if (FULL) {
                    const unsigned cb16 = (unsigned)((head_base + (size_t)(cp - 1) * L * ts) * 2u);
                    constexpr int CPR = (OPT >> 8) & 3;      // cache policy of the row requests (experiment): 0 default, 1 nt, 2 sc1, 3 sc0 sc1 nt
                    if (w == 1) rows_lean<0, CPR>(lds, p, cp - 1, cb16, ll); else if (w == 2) rows_lean<1, CPR>(lds, p, cp - 1, cb16, ll);
                    else if (w == 3) rows_lean<2, CPR>(lds, p, cp - 1, cb16, ll);
                } else if (w > 0 && cp >= 1) dma_chunk<LdsV8, 3>(lds, p, head_base + (size_t)(cp - 1) * L * ts, cp - 1, w - 1, ts, dl);
                Prep8 dd{};
                if ((OPT & 32768) && FULL && w > 0) {           // S0 requested BEFORE the prepare: ~700 cycles more to land (the wait for it ends the step)
                    lds_flag_wait(&lds.flag[3], 4u * (unsigned)(n - 1));
                    const float* sc = sbase + (size_t)(cd - 1) * N * N;
                    if (w == 1) s0_lean<0>(lds, sc, ll); else if (w == 2) s0_lean<1>(lds, sc, ll); else s0_lean<2>(lds, sc, ll);
                }
                if (AHEAD && w > 0 && do_prep) { dd = prep8(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la); lds_flag_add(&lds.flag[6]); }
                // T = (I - M_za)^-1 of the I waves' chunk, from the images this role built a step ago: the doubling chain is 28
                // dependent MFMA / split stages and nobody needs T before the I waves have formed dSA
                if (w == 0 && (FULL || (cd >= 0 && cd <= nchunk - 1))) {
                    if (PT != PP) wave_priority<PT>();
                    if (!(SKIP & 8)) wkv7v6::scores6<TBF16, LdsV8, ChunkImg7, true>(lds, lds.b[cd % 3], 0, c16, g, la);      // SKIP bit 3 (timing experiment): no T chain
                    lds_flag_add(&lds.flag[2]);
                    if (PT != PP) wave_priority<PP>();
                }
                if (AHEAD && w == 0 && do_prep) { dd = prep8(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la); lds_flag_add(&lds.flag[6]); }
                // the J waves have lifted S0 and their dS operands into registers (and split them: VALU only, like the tail, which
                // therefore runs beside their matrix-core phase); J is active in steps 2 .. nchunk + 1 and counts 4 per step
                if (!(OPT & 4) && !(SKIP & 4) && (FULL || (n >= 2 && n <= nchunk + 1))) lds_flag_wait(&lds.flag[3], 4u * (unsigned)(n - 1));
                // S0 of chunk cd = s[cd-1] for the j-split of the next step, into the image the J waves have just left
                {
                    const int k0 = w == 1 ? 0 : w == 2 ? 5 : 10, k1 = w == 0 ? 0 : w == 1 ? 5 : w == 2 ? 10 : 16;      // waves 1-3: 5 5 6 KB
                    if ((OPT & (128 | 4)) || ((OPT & 32768) && FULL)) {
                    } else if (FULL) {
                        const float* sc = sbase + (size_t)(cd - 1) * N * N;
                        constexpr int CPS = (OPT >> 10) & 3;     // cache policy of the S0 requests (experiment)
                        if (w == 1) s0_lean<0, CPS>(lds, sc, ll); else if (w == 2) s0_lean<1, CPS>(lds, sc, ll); else if (w == 3) s0_lean<2, CPS>(lds, sc, ll);
                    } else if (cd >= 0 && cd <= nchunk - 1) dma_state(lds.s0, cd > 0 ? sbase + (size_t)(cd - 1) * N * N : nullptr, k0, k1, lane);
                }
                constexpr bool WIDE = (OPT & 8192) != 0;
                const bool do_tail = !JTAIL && (FULL || (ct >= 0 && ct <= nchunk - 1));
                if (do_tail) tail8<(OPT & 64) != 0, (OPT & 4096) != 0, WIDE>(lds, ct & 1, qt, p, head_base + (size_t)ct * L * ts, lane_boff, c16, w, g, la);
                WKV_STAMP(0)
                if (SHIFT && !JTAIL) { q2 = q1; q1 = q0; }
                TailQ& qn = SHIFT ? q0 : qt;
                if (do_prep && JTAIL) {
                    if (!AHEAD) dd = prep8(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la);
                    *reinterpret_cast<float4*>(&lds.x2r[cp % 3][la.f32]) = make_float4(dd.x2[0], dd.x2[1], dd.x2[2], dd.x2[3]);
                }
                if (do_prep && !JTAIL) {
                    if (!AHEAD) dd = prep8(lds.b[cp % 3], raw, c16, 16 * w + 4 * g, la);
                    qn.q = raw.q; qn.k = raw.k; qn.z = raw.z; qn.a = raw.a;
                    qn.x2[0] = dd.x2[0]; qn.x2[1] = dd.x2[1]; qn.x2[2] = dd.x2[2]; qn.x2[3] = dd.x2[3];
                    if (VRWKV_V8_TAILQ) {
#pragma unroll
                        for (int e = 0; e < (VRWKV_V8_TAILQ ? 4 : 1); ++e) { qn.cc[e] = dd.cc[e]; qn.ic[e] = dd.ic[e]; }
                    }
                }
                if (AHEAD && w > 0 && do_prep) {
                    // scores of chunk cp for the I waves' next step: all four P waves' images are written (flag 6), and the I waves are
                    // past the last use of the previous scores (flag 0: four per step in which they are active, steps 1 ..)
                    lds_flag_wait(&lds.flag[6], 4u * (unsigned)(n + 1));
                    if (n >= 1) lds_flag_wait(&lds.flag[0], 4u * (unsigned)n);
                    wkv7v6::scores6<true>(lds, lds.b[cp % 3], w, c16, g, la);
                }
                if (WIDE && do_tail) {                          // after the score pieces: by then wave 0 (T chain first) has written its quarter too
                    n_wide += 4u;
                    lds_flag_wait(&lds.flag[7], n_wide);        // all four P waves' results are in the image
                    wide_store(lds, p, head_base + (size_t)ct * L * ts, ts, lane, w);
                    lds_flag_add(&lds.flag[4]);                 // (waits for the reads) the J waves may overwrite `res`
                }
                WKV_STAMP(1)
                if (FULL && !JTAIL && (OPT & 8192)) { if (w < 2) vmem_wait<3>(); else vmem_wait<2>(); }
                else if (FULL && !JTAIL && !(OPT & 64)) vmem_wait<5>(); else vmem_drain();      // JTAIL: this role issues requests only
            } else if (w == 0 && cd >= 0 && cd <= nchunk - 1) lds_flag_add(&lds.flag[2]);
            WKV_STAMP(2)
            block_sync_lds();
            WKV_STAMP(3)
        };
        int n = 0;
        for (; n < 3 && n < nsteps; ++n) pstep(n, BoolTag<false>{}, q2, BoolTag<true>{});
#if VRWKV_V8_ROTATE
        for (; !JTAIL && n + 2 < nchunk - 1; n += 3) {  // three steps: the entries rotate through the names and are back in place (JTAIL has no queue)
            pstep(n, BoolTag<true>{}, q2, BoolTag<false>{});
            pstep(n + 1, BoolTag<true>{}, q1, BoolTag<false>{});
            pstep(n + 2, BoolTag<true>{}, q0, BoolTag<false>{});
        }
#endif
        for (; n < nchunk - 1; ++n) pstep(n, BoolTag<true>{}, q2, BoolTag<true>{});
        for (; n < nsteps; ++n) pstep(n, BoolTag<false>{}, q2, BoolTag<true>{});
        WKV_STAMP_FLUSH(512 + 64 * VRWKV_PROF_WAVE, 10, 5)
        return;
    }

    if (role == 0) {
        // ================================================================== I: chunk ci = nchunk - n  (steps 1 .. nchunk)
        wave_priority<PI>();
        f32x4 dS1[4];                                       // dS1[jb][r] = dS[i = 16w+c16][j = tix(jb, 4g+r)]: the only copy of dL/dS
#pragma unroll
        for (int x = 0; x < 4; ++x) dS1[x] = zero4();
        unsigned n_sc = 0, n_t = 0;
        const int img_row = dsi_off<(OPT & 2) != 0>(16 * w + c16, 8 * g);   // this lane's 16-byte piece of the dS image, k block 0 (+ 32 columns: block 1)
        const int img_row1 = dsi_off<(OPT & 2) != 0>(16 * w + c16, 32 + 8 * g);
        // OPT & 1: transposing reads of a tile pair, dealt over both halves of the 16-byte slots (see the template comment)
        const int hb4 = 4 * ((c16 >> 3) & 1);
        const int tra[2] = {la.tri[0] + hb4, la.tri[1] + hb4}, trb[2] = {la.tri[0] + 4 - hb4, la.tri[1] + 4 - hb4};
        for (int n = 0; n < nsteps; ++n) {
            const int ci = nchunk - n, cj = ci + 1;         // this role's chunk | the J waves' chunk of this step
            WKV_STAMP(4)
            // the seven score / score-gradient pieces: wave 0 shares its SIMD with the T chain (P wave 0) and takes one score-gradient piece only.
            //   wave 0: dM_za   wave 1: M_qa, dM_qk   wave 2: M_qk, dM_qa   wave 3: M_zk, dM_zk      (6 / 8 / 10 / 10 MFMAs)
            // Score gradients of the J waves' chunk FIRST (their dR is one step old): the J waves need them in the middle of THIS step
            // (their dM products), the scores below are for this role's own i-split.  With the scores first the J waves stood
            // 1.2k cycles per step at flag 1 (profiles/r4_wkv7_phases_b16.json: J_dMwait) and carried the step's critical path.
            auto score_grads = [&]() {
                if (cj >= 0 && cj <= nchunk - 1) {
                    const ChunkImg7& Bj = lds.b[cj % 3];
                    const uint16_t *vi = lds.vdy[cj & 3][0], *dyi = lds.vdy[cj & 3][1], *drh = lds.dr[cj & 1][0], *drl = lds.dr[cj & 1][1];
                    // wave 0: dM_za (6 MFMAs; it has no score piece and would stand at flag 0 meanwhile)  1: dM_qk (2)  2: dM_qa (4)  3: dM_zk (4)
                    if (!(SKIP & 2)) dscores7(lds, Bj.sa[0], Bj.sa[1], vi, dyi, drh, drl, w == 0 ? 0 : w == 1 ? 3 : w == 2 ? 2 : 1, c16, g, la);
                    lds_flag_add(&lds.flag[1]);
                }
            };
            if (VRWKV_V8_DM_FIRST) score_grads();
            if (ci >= 0 && ci <= nchunk - 1) {
                if (w > 0 && !VRWKV_V8_SCORES_ON_J && !AHEAD) {
                    if (!(SKIP & 2)) wkv7v6::scores6<true>(lds, lds.b[ci % 3], w, c16, g, la);
                    lds_flag_add(&lds.flag[0]);
                }
                n_sc += 3; n_t += 1;
            }
            if (!VRWKV_V8_DM_FIRST) score_grads();
            WKV_STAMP(0)
            if (!(SKIP & 2) && ci >= 0 && ci <= nchunk - 1) {
                const ChunkImg7& B = lds.b[ci % 3];
                const uint16_t* dyi = lds.vdy[ci & 3][1];
                const size_t cbase = head_base + (size_t)ci * L * ts;
                // ------------------------------------------------------------ i-split (i = 16w + c16)
                f32x4 dSc[4];                               // diag(c_L) dS^T: operand of dSA / dV, start of the update, and the J waves' dU
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float4 cl = *reinterpret_cast<const float4*>(&B.cl[tix(jb, 4 * g)]);
                    dSc[jb] = dS1[jb];
                    dSc[jb][0] *= cl.x; dSc[jb][1] *= cl.y; dSc[jb][2] *= cl.z; dSc[jb][3] *= cl.w;
                }
                bf16x8 sh[2], sl[2];
                tiles_op(dSc, sh, sl);
                // VRWKV_V8_CHAINS: 0 = one accumulation chain per product (7 / 9 dependent MFMAs), 1 = two chains (k block 0 | k block 1), 2 = two
                // chains with the products that need no other role's results (dS with this chunk's Ah / Kh images) issued before the
                // waits for the score pieces and T
                f32x4 c0 = zero4(), c1 = zero4(), v0 = zero4(), v1 = zero4();
                auto own_products = [&]() {
                    const bf16x8 ah0 = ld16(&B.opnd[4][la.row[0]]), ah1 = ld16(&B.opnd[4][la.row[1]]);
                    const bf16x8 kh0 = ld16(&B.opnd[6][la.row[0]]), kh1 = ld16(&B.opnd[6][la.row[1]]);
                    c0 = mfma32(ah0, sh[0], c0);
                    c1 = mfma32(ah1, sh[1], c1);
                    v0 = mfma32(sh[0], kh0, v0);
                    v1 = mfma32(sh[1], kh1, v1);
                    c0 = mfma32(ah0, sl[0], c0);
                    c1 = mfma32(ah1, sl[1], c1);
                    v0 = mfma32(sl[0], kh0, v0);
                    v1 = mfma32(sl[1], kh1, v1);
                    c0 = mfma32(ld16(&B.opnd[5][la.row[0]]), sh[0], c0);
                    c1 = mfma32(ld16(&B.opnd[5][la.row[1]]), sh[1], c1);
                    v0 = mfma32(sh[0], ld16(&B.opnd[7][la.row[0]]), v0);
                    v1 = mfma32(sh[1], ld16(&B.opnd[7][la.row[1]]), v1);
                };
                if (VRWKV_V8_CHAINS == 2) own_products();
                // the same operands, one step later, for the J waves: lane (i, g) holds columns j = 32 kb + 8g .. +7 of row i.  The J
                // waves took their operands of the previous image at the top of this step (flag 3; J is active in steps 2 .. nchunk+1)
                if (!(SKIP & 4) && n >= 2 && n <= nchunk + 1) lds_flag_wait(&lds.flag[3], 4u * (unsigned)(n - 1));
                *reinterpret_cast<bf16x8*>(&lds.dsi[0][img_row]) = sh[0]; *reinterpret_cast<bf16x8*>(&lds.dsi[0][img_row1]) = sh[1];
                *reinterpret_cast<bf16x8*>(&lds.dsi[1][img_row]) = sl[0]; *reinterpret_cast<bf16x8*>(&lds.dsi[1][img_row1]) = sl[1];
                if (!AHEAD) lds_flag_wait(&lds.flag[0], n_sc);
                WKV_STAMP(1)
                const uint2 dyv = lds_read_tr16(&dyi[la.trc]);                   // dY[4g+e][i]
                const bf16x8 dyd = mk8(dyv, dyv);
                // dSA[t][i] = sum_s M_qa[s][t] dY[s][i] + sum_j Ah[t][j] c_L[j] dS[i][j]
                f32x4 dSA;
                if (VRWKV_V8_CHAINS == 0) {
                    dSA = mfma32(ld16(&lds.sc[0][la.hl]), dyd, zero4());
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 ah = ld16(&B.opnd[4][la.row[kb]]);
                        dSA = mfma32(ah, sh[kb], dSA);
                        dSA = mfma32(ah, sl[kb], dSA);
                        dSA = mfma32(ld16(&B.opnd[5][la.row[kb]]), sh[kb], dSA);
                    }
                } else {
                    if (VRWKV_V8_CHAINS == 1) own_products();
                    c0 = mfma32(ld16(&lds.sc[0][la.hl]), dyd, c0);
                    dSA = c0 + c1;
                }
                uint2 xh, xl, rh, rl;
                split4(dSA, xh, xl);
                const bf16x8 xhl = mk8(xh, xl);
                // dR = T^T dSA in both orientations: [t][i] stays in registers, [i][t] (token per lane) goes to LDS
                lds_flag_wait(&lds.flag[2], n_t);               // T of this chunk (P wave 0's doubling chain) is in LDS
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
                    const bf16x8 rhl = mk8(rh, rl);
                    f32x4 dV;
                    if (VRWKV_V8_CHAINS == 0) {
                        dV = mfma32(dyd, ld16(&lds.sc[1][la.hl]), zero4());
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            const bf16x8 kh = ld16(&B.opnd[6][la.row[kb]]);
                            dV = mfma32(sh[kb], kh, dV);
                            dV = mfma32(sl[kb], kh, dV);
                            dV = mfma32(sh[kb], ld16(&B.opnd[7][la.row[kb]]), dV);
                        }
                        dV = mfma32(rhl, ld16(&lds.dz[0][la.row[0]]), dV);                 // [M_zk_h M_zk_h]
                        dV = mfma32(rhl, ld16(&lds.dz[0][la.row[1]]), dV);                 // [M_zk_l 0]
                    } else {
                        v0 = mfma32(dyd, ld16(&lds.sc[1][la.hl]), v0);
                        v1 = mfma32(rhl, ld16(&lds.dz[0][la.row[0]]), v1);
                        v0 = mfma32(rhl, ld16(&lds.dz[0][la.row[1]]), v0);
                        dV = v0 + v1;
                    }
                    *reinterpret_cast<uint2*>(p.dv + cbase + out_off) = make_uint2(cvt_pk_bf16(dV[0], dV[1]), cvt_pk_bf16(dV[2], dV[3]));
                }
                if (AHEAD) lds_flag_add(&lds.flag[0]);            // (waits for the reads) the scores of this chunk are consumed
                WKV_STAMP(6)
                // dS^T <- diag(c_L) dS^T + [Qt^T | Zt^T] [dY ; dR]
                const bf16x8 y1 = mk8(dyv, rh), y2 = mk8(0u, 0u, rl.x, rl.y);
                if (OPT & 1) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        // read A: rows 4g, 4g+1 of the even tile | rows 4g+2, 4g+3 of the odd one; read B: the complement
                        const uint2 qha = lds_read_tr16(&B.opnd[2][tra[pr]]), qhb = lds_read_tr16(&B.opnd[2][trb[pr]]);
                        const uint2 zha = lds_read_tr16(&B.opnd[0][tra[pr]]), zhb = lds_read_tr16(&B.opnd[0][trb[pr]]);
                        const uint2 qla = lds_read_tr16(&B.opnd[3][tra[pr]]), qlb = lds_read_tr16(&B.opnd[3][trb[pr]]);
                        const uint2 zla = lds_read_tr16(&B.opnd[1][tra[pr]]), zlb = lds_read_tr16(&B.opnd[1][trb[pr]]);
#pragma unroll
                        for (int od = 0; od < 2; ++od) {
                            const int jb = 2 * pr + od;
                            f32x4 acc = dSc[jb];
                            const bf16x8 xh8 = od ? mk8(qhb.x, qha.y, zhb.x, zha.y) : mk8(qha.x, qhb.y, zha.x, zhb.y);
                            const bf16x8 xl8 = od ? mk8(qlb.x, qla.y, zlb.x, zla.y) : mk8(qla.x, qlb.y, zla.x, zlb.y);
                            acc = mfma32(xh8, y1, acc);
                            acc = mfma32(xl8, y1, acc);
                            acc = mfma32(xh8, y2, acc);
                            dS1[jb] = acc;
                        }
                    }
                } else {
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
            }
            WKV_STAMP(2)
            block_sync_lds();
            WKV_STAMP(3)
        }
        WKV_STAMP_FLUSH(64 * VRWKV_PROF_WAVE, 0, 5)
        if (PROF && blockIdx.x == 0 && tid == 64 * VRWKV_PROF_WAVE) { p.dbg[15] = realtime64_() - rt0_; p.dbg[18] = tacc_[5]; p.dbg[19] = tacc_[6]; }   // i-split: dSA + dR | dV
        return;
    }

    // ====================================================================== J: chunk cj = nchunk + 1 - n  (steps 2 .. nchunk + 1)
    wave_priority<PJ>();
    const int j = 16 * w + c16;                         // key column of the j-split tiles
    // transposing reads of the dS image: operand rows j = 16w + c16, k = i = 32 kb + 8g + e: rows 32 kb + 8g + 4h + (c16 >> 2)
    int tro[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) tro[kb][h] = dsi_off<(OPT & 2) != 0>(32 * kb + 8 * g + 4 * h + (c16 >> 2), 16 * w + 4 * (c16 & 3));
    bf16x8 s0h_p[2] = {mk8(0u, 0u, 0u, 0u), mk8(0u, 0u, 0u, 0u)}, s0l_p[2] = {mk8(0u, 0u, 0u, 0u), mk8(0u, 0u, 0u, 0u)};   // S0 operands of the previous step = S_L of this one
    unsigned n_dm = 0;
    JTailIn jt{};                                       // JTAIL: results and tail inputs of the chunk of the previous step
    // OPT & 4: S0 of the J waves' chunk comes straight from memory into THEIR registers, two steps ahead (two register sets of 16, used in
    // turn), instead of through the P waves' LDS-DMA into a 16 KB image one step ahead: the checkpoint is the J waves' private operand
    // (row j = 16w + c16 of S^T, 4 x 16 B per lane), the J waves issue no other memory operation (vmcnt is theirs alone), and a request that
    // has two steps to land is off the step's critical path (profiles/r6b_ab_opt.jsonl: without the S0 requests the step is 12.7 % shorter).
    // sreg: the register set holding S0 of this step's chunk (refilled for the chunk two steps on as soon as it has been split)
    // unconditional (a branch around a load makes the compiler merge the two paths with register copies, which wait for the load on the
    // spot): chunk 0 (S0 = 0) and the requests past the start of the sequence read checkpoint 0 and the consumer zeroes / ignores them
    const unsigned s_lane = (unsigned)(j * N + 8 * g) * 4u;          // byte offset of S^T[j][i], i = tix(ib, 4g) = 32 (ib >> 1) + 8g + 4 (ib & 1); scalar base + lane offset
    auto ld_s0 = [&](f32x4 (&S)[4], int c) {
        const void* sp = uniform_ptr(sbase + (size_t)(c > 1 ? c - 1 : 0) * N * N);
        global_load16_inplace<0, 5>(S[0], sp, s_lane); global_load16_inplace<16>(S[1], sp, s_lane);
        global_load16_inplace<128>(S[2], sp, s_lane); global_load16_inplace<144>(S[3], sp, s_lane);
    };
    auto jstep = [&](int n, f32x4 (&sreg)[4]) {
        const int cj = nchunk + 1 - n;
        WKV_STAMP(4)
        if (VRWKV_V8_SCORES_ON_J && w < 3 && cj - 1 >= 0 && cj - 1 <= nchunk - 1) {      // experiment: the I waves' three score pieces on J waves 0-2
            wkv7v6::scores6<true>(lds, lds.b[(cj - 1) % 3], w + 1, c16, g, la);
            lds_flag_add(&lds.flag[0]);
        }
        if (!(SKIP & 4) && cj >= 0 && cj <= nchunk - 1) {
            dummy_issue<1>();
            const ChunkImg7& B = lds.b[cj % 3];
            const uint16_t* drh = lds.dr[cj & 1][0];
            const uint16_t* drl = lds.dr[cj & 1][1];
            const uint16_t* vi = lds.vdy[cj & 3][0];
            const uint16_t* dyi = lds.vdy[cj & 3][1];
            n_dm += 4;
            // ---------------------------------------------------------------- j-split (j = 16w + c16)
            f32x4 dZt, dQt, dAh, dKh;
            {
                const float clj = B.cl[j];
                f32x4 S0[4];
                if (OPT & 4) { if (OPT & 16) vmem_wait_for<0>(sreg[0], sreg[1], sreg[2], sreg[3]); else vmem_wait_for<4>(sreg[0], sreg[1], sreg[2], sreg[3]); }
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    // [ib][r] = S0[i = tix(ib, 4g+r)][j]  <-  image row j (zeros for the first chunk of the sequence)
                    if (OPT & 4) { S0[ib] = cj > 0 ? sreg[ib] : zero4(); continue; }
                    const float4 x = *reinterpret_cast<const float4*>(&lds.s0[f32_off(j, tix(ib, 4 * g))]);
                    S0[ib][0] = x.x; S0[ib][1] = x.y; S0[ib][2] = x.z; S0[ib][3] = x.w;
                }
                // dU = dS diag(c_L) as the I waves split it a step ago: [duh | dul][kb] = rows j, k = i = 32 kb + 8g + e
                bf16x8 duh[2], dul[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    duh[kb] = mk8(lds_read_tr16(&lds.dsi[0][tro[kb][0]]), lds_read_tr16(&lds.dsi[0][tro[kb][1]]));
                    dul[kb] = mk8(lds_read_tr16(&lds.dsi[1][tro[kb][0]]), lds_read_tr16(&lds.dsi[1][tro[kb][1]]));
                }
                bf16x8 s0h[2], s0l[2];
                if (OPT & 16384) lds_flag_add(&lds.flag[3]);    // the reads have returned (the add waits for them): hand S0 / the dS image back BEFORE the split's ~60 VALU instructions
                tiles_op(S0, s0h, s0l);
                if (OPT & 4) order_after(s0h[0], s0h[1], s0l[0], s0l[1]);      // the split has consumed the old contents
                if (OPT & 4) ld_s0(sreg, (OPT & 16) ? cj - 1 : cj - 2);          // this set's next occupant: two steps to land (OPT & 16: one set, one step)
                if (!(OPT & 16384)) lds_flag_add(&lds.flag[3]);                 // (waits for the reads) S0 and the dS image may be overwritten
                WKV_STAMP(5)
                // decay-gradient term of this chunk: sum_i dS[i][j] S_L[i][j] = diag((dU)^T S_L)[j] / c_L[j], S_L = the S0 of a step ago
                {
                    f32x4 G = mfma32(duh[0], s0h_p[0], zero4());
                    G = mfma32(dul[0], s0h_p[0], G);
                    G = mfma32(duh[0], s0l_p[0], G);
                    G = mfma32(duh[1], s0h_p[1], G);
                    G = mfma32(dul[1], s0h_p[1], G);
                    G = mfma32(duh[1], s0l_p[1], G);
                    // G[r] = (m = 4g + r, n = c16): the diagonal element of column c16 sits in lane group g == c16 >> 2, register c16 & 3
                    const int r = c16 & 3;
                    const float d01 = r & 1 ? G[1] : G[0], d23 = r & 1 ? G[3] : G[2];
                    if (OPT & 4) {          // the address formed from `j` on the spot: hoisted out of the loop it is spilled (168 registers), and its reload waits for every request in flight
                        int jj = j;
                        asm volatile("" : "+v"(jj));
                        if ((c16 >> 2) == g) lds.glast[cj & 1][jj] = (r & 2 ? d23 : d01) * fast_rcp(clj);
                    } else
                    if ((c16 >> 2) == g) lds.glast[cj & 1][j] = (r & 2 ? d23 : d01) * fast_rcp(clj);
                }
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
                s0h_p[0] = s0h[0]; s0h_p[1] = s0h[1]; s0l_p[0] = s0l[0]; s0l_p[1] = s0l[1];
                WKV_STAMP(6)
            }
            WKV_STAMP(0)
            // the tail of the previous step's chunk: VALU + stores beside this chunk's matrix-core phase and the wait for its score gradients
            if (JTAIL && cj + 1 <= nchunk - 1) jtail_run(jt, lds, cj + 1, p, head_base + (size_t)(cj + 1) * L * ts, out_off * 2u, c16, w, g);
            WKV_STAMP(7)
            // ---------------------------------------------------------------- dM products
            const bf16x8 qzh = mk8(lds_read_tr16(&B.opnd[2][la.trc]), lds_read_tr16(&B.opnd[0][la.trc]));      // [Qt^T | Zt^T]
            const bf16x8 qzl = mk8(lds_read_tr16(&B.opnd[3][la.trc]), lds_read_tr16(&B.opnd[1][la.trc]));
            const bf16x8 akh = mk8(lds_read_tr16(&B.opnd[4][la.trc]), lds_read_tr16(&B.opnd[6][la.trc]));      // [Ah^T | Kh^T]
            const bf16x8 akl = mk8(lds_read_tr16(&B.opnd[5][la.trc]), lds_read_tr16(&B.opnd[7][la.trc]));
            lds_flag_wait(&lds.flag[1], n_dm);
            WKV_STAMP(1)
            {
                // dZt += dM_za Ah + dM_zk Kh ; dQt += dM_qa Ah + dM_qk Kh : X = [Ah^T | Kh^T], Y = pair image rows
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
            if (JTAIL) {
                jtail_lift(jt, lds, B, cj, dZt, dQt, dAh, dKh, la);
            } else {
                if (!(SKIP & 1) && n >= 3) lds_flag_wait(&lds.flag[4], 4u * (unsigned)(n - 2));
                *reinterpret_cast<float4*>(&lds.res[0][la.f32]) = make_float4(dZt[0], dZt[1], dZt[2], dZt[3]);
                *reinterpret_cast<float4*>(&lds.res[1][la.f32]) = make_float4(dQt[0], dQt[1], dQt[2], dQt[3]);
                *reinterpret_cast<float4*>(&lds.res[2][la.f32]) = make_float4(dAh[0], dAh[1], dAh[2], dAh[3]);
                *reinterpret_cast<float4*>(&lds.res[3][la.f32]) = make_float4(dKh[0], dKh[1], dKh[2], dKh[3]);
            }
        }
        WKV_STAMP(2)
        block_sync_lds();
        WKV_STAMP(3)
    };
    if ((OPT & 4) && (OPT & 16)) {
        f32x4 sa_[4] = {zero4(), zero4(), zero4(), zero4()};
        if (!(SKIP & 4)) ld_s0(sa_, nchunk - 1);
        for (int n = 0; n < nsteps; ++n) jstep(n, sa_);
    } else if (OPT & 4) {
        f32x4 sa_[4] = {zero4(), zero4(), zero4(), zero4()}, sb_[4] = {zero4(), zero4(), zero4(), zero4()};
        if (!(SKIP & 4)) { ld_s0(sa_, nchunk - 1); ld_s0(sb_, nchunk - 2); }
        int n = 0;
        for (; n + 1 < nsteps; n += 2) { jstep(n, sa_); jstep(n + 1, sb_); }      // the chunks of even steps live in one set, those of odd steps in the other
        if (n < nsteps) jstep(n, sa_);
    } else {
        f32x4 none_[4] = {zero4(), zero4(), zero4(), zero4()};
        for (int n = 0; n < nsteps; ++n) jstep(n, none_);
    }
    if (JTAIL && !(SKIP & 4)) jtail_run(jt, lds, 0, p, head_base, out_off * 2u, c16, w, g);          // the tail of chunk 0
    WKV_STAMP_FLUSH(256 + 64 * VRWKV_PROF_WAVE, 5, 5)
    if (PROF && blockIdx.x == 0 && tid == 256 + 64 * VRWKV_PROF_WAVE) { p.dbg[16] = tacc_[5]; p.dbg[17] = tacc_[6]; p.dbg[20] = tacc_[7]; }   // j-split: operand reads + split | outputs
}

}  // namespace wkv7v8x
