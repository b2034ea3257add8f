// Weight gradient of a square / wide projection: C[N1 x N2] = A^T B with A (M x N1) and B (M x N2) bf16 row-major, fp32
// accumulation -- the "N,T" class (both operands have the contraction index M as their SLOW index) -- gfx950.
//
// These are dW = dy^T x of receptance / key / value / output (N1 = N2 = C), the channel-mix key / value (C x 4C) and the head
// (VisualRWKV-v7/v7.00/src/model.py:150-153, 214-215, 281), M = every token of the micro-batch (41 984).  The library's kernels
// for this class run at 0.49-0.55 matrix-core utilisation inside the training step where its "T,N" kernels of the forward reach
// 0.78-0.87 (profiles/r3_step_mfma_util.json), and the C x C shapes have only 64-128 output tiles for 256 CUs.
//
// Same idea as lora_wgrad.h, scaled up: tiles go global -> LDS as they lie in memory ([m][column] rows, by LDS-DMA, 1 KB per
// instruction, no registers) and BOTH MFMA operands are fetched with ds_read_b64_tr_b16 -- v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k = m per lane for one row i / column j, which is a column of the image: no transposed copy of either activation
// exists anywhere.  One workgroup = 8 waves = a 256 x 256 tile of C (wave (wr, wc): rows 128 wr.., columns 64 wc..: 4 x 2 MFMA
// tiles, 128 accumulator registers) x one contiguous slice of M (split-K: S slices fill the chip for the C x C shapes; fp32
// partials are summed and rounded by wgrad_big_reduce).  K step 32 rows = 16 KB per operand, three LDS stages (96 KB): the
// requests of stage s+2 are issued before the MFMAs of stage s; one workgroup barrier per stage.
// LDS rows are 512 B = two bank periods, so the 16-byte slots of a row are XOR-ed with 4 (row & 3) -- on the DMA's SOURCE
// address: the four rows a transposing read touches then sit in four different bank groups (conflict-free).
// Workgroups that share an A column block are given ids that are equal mod 8 (one XCD: its L2 serves the block to all of them).
#pragma once
#include <gfx950_prims.h>

namespace wgb {

constexpr int TM = 256, TN = 256, KT = 32;          // tile of C, rows of M per stage
#ifndef WGB_STAGES
#define WGB_STAGES 3
#endif
constexpr int STAGES = WGB_STAGES;
constexpr int ROWB = TM * 2;                        // bytes of one LDS row (256 bf16)
constexpr int OPB = KT * ROWB;                      // bytes of one operand tile of a stage (16 KB)

struct Args {
    long M;
    int N1, N2, S;
    const uint16_t* A;              // (M, N1)
    const uint16_t* B;              // (M, N2)
    float* part;                    // [S][N1][N2] fp32 (S > 1)
    uint16_t* out;                  // (N1, N2) bf16 (S == 1)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint2 lo, uint2 hi) {
    u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void wgrad_big_kernel(Args p) {
    char* lds = dyn_lds();                             // [STAGES][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // block id -> (slice, tile column, tile row) with the tile row in the low bits: ids that are equal mod 8 share A's column block
    const int T1 = p.N1 / TM, T2 = p.N2 / TN;
    const int i1 = blockIdx.x % T1, rest = blockIdx.x / T1, i2 = rest % T2, sl = rest / T2;
    const long nst = p.M / KT;
    const long s0 = nst * sl / p.S, s1 = nst * (sl + 1) / p.S;
    // ---- requests: a stage is 16 + 16 instructions of 1 KB (2 rows of 512 B each); wave w issues instructions 2w, 2w+1 of A and of B.
    // lane l of instruction j: row 2j + (l >> 5), LDS slot l & 31 <- source slot (l & 31) ^ 4 (row & 3)
    unsigned offA[2], offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned row = 2u * (2u * wave + q) + ((unsigned)lane >> 5), slot = ((unsigned)lane & 31u) ^ (4u * (row & 3u));
        offA[q] = row * (unsigned)p.N1 * 2u + 16u * slot;
        offB[q] = row * (unsigned)p.N2 * 2u + 16u * slot;
    }
    const char* gA = reinterpret_cast<const char*>(p.A + (size_t)s0 * KT * p.N1 + (size_t)i1 * TM);
    const char* gB = reinterpret_cast<const char*>(p.B + (size_t)s0 * KT * p.N2 + (size_t)i2 * TN);
    const size_t stepA = (size_t)KT * p.N1 * 2, stepB = (size_t)KT * p.N2 * 2;
    const unsigned lds0 = lds_addr_u32(lds);
    auto request = [&](int slot) {                      // the next stage (gA / gB advance) into LDS stage `slot` (compile-time after unrolling)
        const unsigned d = lds0 + (unsigned)slot * (2u * OPB) + (unsigned)(2 * wave) * 1024u;
        lds_dma16_lean<0>(gA, offA[0], d);
        lds_dma16_lean<0>(gA, offA[1], d + 1024u);
        lds_dma16_lean<0>(gB, offB[0], d + OPB);
        lds_dma16_lean<0>(gB, offB[1], d + OPB + 1024u);
        gA += stepA; gB += stepB;
    };
    // ---- operand fetch: ds_read_b64_tr_b16 on a [32][256] image.  v_mfma_f32_32x32x16_bf16: lane l holds row / column (l & 31), k = 8 (l >> 5) + e.
    // 16-lane group (l >> 4): column block 16 ((l >> 4) & 1) of the 32, rows 8 (l >> 5) + 4h + ((l & 15) >> 2), h = 0, 1; the lane
    // points at 4 consecutive columns 4 (l & 3) of its row.  Physical slot = (column / 8) ^ 4 (row & 3).  One byte offset per
    // fragment and lane, computed once: stage, k16 (+ 8 KB) and h (+ 2 KB) are immediates of the read.
    const int rq = (lane & 15) >> 2;                                   // row & 3 of both reads (8 (l>>5) + 4h are multiples of 4)
    const int cl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);            // column inside a 32-column MFMA tile
    const int rbase = 8 * (lane >> 5) + rq;
    auto frag_off = [&](int col0) {
        const int col = col0 + cl;
        return rbase * ROWB + (((col >> 3) ^ (4 * rq)) * 16) + (col & 7) * 2;
    };
    int fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag_off(128 * wr + 32 * i);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = OPB + frag_off(64 * wc + 32 * j);
    auto frag = [&](const char* stage, int off, int k16) -> bf16x8 {
        const char* base = stage + off + k16 * 16 * ROWB;
        return mk8(lds_read_tr16(reinterpret_cast<const uint16_t*>(base)), lds_read_tr16(reinterpret_cast<const uint16_t*>(base + 4 * ROWB)));
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int slot) {
        const char* st = lds + slot * (2 * OPB);
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag(st, fa[i], k16);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag(st, fb[j], k16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
        }
    };
    // One stage: request stage s + STAGES - 1 into the slot stage s - 1 has left, multiply stage s, wait for this wave's requests of
    // stage s + 1 (all but the newest 4 (STAGES - 2)), barrier.  Unrolled by STAGES so that every LDS address is lane offset + immediate.
    const long ns = s1 - s0;
    long issued = 0;
    if (ns > 0) {
#pragma unroll
        for (int q = 0; q < STAGES - 1; ++q) if (q < ns) { request(q); ++issued; }
        if (issued > 1) vmem_wait<4 * (STAGES - 2)>(); else vmem_drain();
        if (ns < STAGES) vmem_drain();
        block_sync_lds();
        long s = 0;
        for (; s + STAGES <= ns - (STAGES - 1); s += STAGES) {        // steady state: every stage of the group requests another
#pragma unroll
            for (int q = 0; q < STAGES; ++q) {
                request((q + STAGES - 1) % STAGES);
                compute(q);
                vmem_wait<4 * (STAGES - 2)>();
                block_sync_lds();
            }
        }
        for (; s < ns; ++s) {                                          // the last stages: nothing left to request beyond ns
            const int q = (int)(s % STAGES);
            const bool more = s + STAGES - 1 < ns;
            if (more) request((q + STAGES - 1) % STAGES);
            compute(q);
            if (more) vmem_wait<4 * (STAGES - 2)>(); else vmem_drain();
            block_sync_lds();
        }
    }
    // ---- epilogue.  C/D of 32x32: register r <-> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    const int rowl = 4 * (lane >> 5), coll = lane & 31;
    if (p.S > 1) {
        float* out = p.part + ((size_t)sl * p.N1 + (size_t)i1 * TM + 128 * wr) * p.N2 + (size_t)i2 * TN + 64 * wc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(size_t)(32 * i + (r & 3) + 8 * (r >> 2) + rowl) * p.N2 + 32 * j + coll] = acc[i][j][r];
    } else {
        uint16_t* out = p.out + ((size_t)i1 * TM + 128 * wr) * p.N2 + (size_t)i2 * TN + 64 * wc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(size_t)(32 * i + (r & 3) + 8 * (r >> 2) + rowl) * p.N2 + 32 * j + coll] = (uint16_t)f32_to_bf16_bits(acc[i][j][r]);
    }
}

// out (bf16, N1 x N2) = sum over the S slices of part, in a fixed order; one thread per 4 consecutive elements
__global__ __launch_bounds__(256) void wgrad_big_reduce(const float* __restrict__ part, int S, long n, uint16_t* __restrict__ out) {
    const long n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = p4[i];
        for (int k = 1; k < S; ++k) {
            const float4 v = p4[(size_t)k * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<uint2*>(out + 4 * i) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
    }
}

}  // namespace wgb
