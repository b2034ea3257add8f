// C (M x N) = A (M x K) B^T for B (N x K): F.linear's "T,N" class (both operands contraction-contiguous) with an element-wise
// epilogue -- the channel-mix of RWKV-7 without its two streaming passes -- gfx950.
//
// RWKV_CMix_x070 (VisualRWKV-v7/v7.00/src/model.py:221-227) is value(relu(key(x_k))^2).  As library GEMMs + glue the forward writes
// key's output, re-reads it to square it (relusq_fwd: 0.26 ms per layer at the benchmark shape) and the backward re-reads the
// gradient of the square and the key output to form the gradient of key's output (relusq_bwd: 0.41 ms): 16 ms of a 446 ms step that
// only move bytes.  With the epilogue inside the GEMM:
//   EPI_RELUSQ   C = relu(acc)^2                       the key projection: only the square is ever written
//   EPI_DRELUSQ  C = bf16(acc) * 2 sqrt(aux)            the input gradient of value: aux = the saved square, relu(k) = sqrt(relu(k)^2)
//   EPI_NONE     C = acc
// (the backward needs relu(k), not k: its square root is what is kept anyway.)
//
// One workgroup = 8 waves = a 256 (m) x 256 (n) tile; wave (wr, wc): m in [128 wr, +128), n in [64 wc, +64) as 2 x 4 tiles of
// v_mfma_f32_32x32x16_bf16 issued as W x^T (rows = n, columns = m), so that a lane ends up with 4 consecutive n of one m: the
// accumulators go to LDS as 8-byte pieces of a [256][256] bf16 image and leave as full 512-byte rows, 16 bytes per lane.
// K step 64 (128-byte rows, two LDS stages of 64 KB, tiles by LDS-DMA with the 16-byte slots of a row XOR-ed with (row >> 1) & 7 on
// the source side: ds_read_b128 of 32 consecutive rows is conflict-free).  Workgroup ids equal mod 8 (one XCD) share four B tiles.
#pragma once
#include <gfx950_prims.h>

namespace gtn {

constexpr int TM = 256, TN = 256, KT = 64;
constexpr int ROWB = KT * 2;                        // 128 B
constexpr int OPB = TM * ROWB;                      // 32 KB per operand tile
constexpr int STAGEB = 2 * OPB;                     // 64 KB
constexpr int EROWB = TN * 2 + 16;                  // epilogue image row: 512 B + 16 B (consecutive rows 4 banks apart)
constexpr int LDS_BYTES = TM * EROWB > 2 * STAGEB ? TM * EROWB : 2 * STAGEB;      // 135 168

enum { EPI_NONE = 0, EPI_RELUSQ = 1, EPI_DRELUSQ = 2 };

struct Args {
    long M;
    int N, K;
    const uint16_t* A;              // (M, K)
    const uint16_t* B;              // (N, K)
    uint16_t* C;                    // (M, N)
    const uint16_t* aux;            // (M, N) for EPI_DRELUSQ
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn_kernel(Args p) {
    char* lds = dyn_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // tile order: XCD x = id % 8 owns the n tiles x, x + 8, ... (its L2 keeps those B tiles) and walks the m tiles
    const int Tn = p.N / TN;
    const int nper = (Tn + 7) / 8;                                      // n tiles per XCD
    const int x = blockIdx.x % 8, lid = blockIdx.x / 8;
    const int tn = x + 8 * (lid % nper), tm = lid / nper;
    if (tn >= Tn || (long)tm * TM >= p.M) return;
    // ---- requests: an operand tile is 32 instructions of 1 KB (8 rows of 128 B); wave w issues instructions w, w+8, w+16, w+24 of each
    // operand.  lane l of instruction j: row 8j + (l >> 3), LDS slot l & 7 <- source slot (l & 7) ^ ((row >> 1) & 7)
    const unsigned rl = (unsigned)lane >> 3;
    const unsigned sw = ((unsigned)lane & 7u) ^ ((4u * (unsigned)wave + (rl >> 1)) & 7u);       // row = 8 (w + 8q) + rl: (row >> 1) & 7 = (4w + (rl >> 1)) & 7
    unsigned offA[4], offB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned row = 8u * ((unsigned)wave + 8u * q) + rl;
        offA[q] = row * (unsigned)p.K * 2u + 16u * sw;
        offB[q] = offA[q];
    }
    const char* gA = reinterpret_cast<const char*>(p.A + (size_t)tm * TM * p.K);
    const char* gB = reinterpret_cast<const char*>(p.B + (size_t)tn * TN * p.K);
    const unsigned lds0 = lds_addr_u32(lds);
    auto request = [&](int slot) {
        const unsigned d = lds0 + (unsigned)slot * STAGEB + (unsigned)wave * 1024u;
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma16_lean<0>(gA, offA[q], d + (unsigned)q * 8192u);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_dma16_lean<0>(gB, offB[q], d + OPB + (unsigned)q * 8192u);
        gA += ROWB; gB += ROWB;                         // the next 64 columns of K
    };
    // ---- operand fetch: lane l holds row (l & 31) of a 32-row block, k = 16 k16 + 8 (l >> 5) .. +7: ONE 16-byte read at slot (2 k16 + (l >> 5)) ^ swz(row)
    const int r32 = lane & 31, kh = lane >> 5;
    auto frag_off = [&](int row0) {                     // byte offset inside an operand tile, k16 = 0
        const int row = row0 + r32;
        return row * ROWB + ((kh ^ ((row >> 1) & 7)) * 16);
    };
    int fw[2], fx[4];                                   // W (n rows) fragments: the MFMA's first operand; x (m rows): the second
#pragma unroll
    for (int i = 0; i < 2; ++i) fw[i] = OPB + frag_off(64 * wc + 32 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) fx[j] = frag_off(128 * wr + 32 * j);
    auto frag = [&](const char* stage, int off, int k16) -> bf16x8 {
        // slot (2 k16 + kh) ^ s = (kh ^ s) ^ (2 k16): the k16 part flips bits 1-2 of the slot index = bytes 32 k16 XOR-ed in
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(stage + (off ^ (32 * k16))));
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the fragments of step k16 + 1 are requested from LDS before the 8 MFMAs of step k16 are issued (two register sets)
    auto compute = [&](int slot) {
        const char* st = lds + slot * STAGEB;
        bf16x8 w[2][2], xx[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) w[0][i] = frag(st, fw[i], 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) xx[0][j] = frag(st, fx[j], 0);
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            const int cur = k16 & 1, nxt = cur ^ 1;
            if (k16 + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) w[nxt][i] = frag(st, fw[i], k16 + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) xx[nxt][j] = frag(st, fx[j], k16 + 1);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x16_bf16(w[cur][i], xx[cur][j], acc[i][j]);
        }
    };
    const int ns = p.K / KT;
    request(0);
    vmem_drain();
    block_sync_lds();
    for (int s = 0; s < ns; s += 2) {                   // two stages per trip: the LDS stage is a compile-time constant
        if (s + 1 < ns) request(1);
        compute(0);
        vmem_drain();
        block_sync_lds();
        if (s + 1 < ns) {
            if (s + 2 < ns) request(0);
            compute(1);
            vmem_drain();
            block_sync_lds();
        }
    }
    // ---- epilogue.  C/D of 32x32: register r <-> row (r & 3) + 8 (r >> 2) + 4 (l >> 5) = n, column l & 31 = m: 4 consecutive n per
    // 4 registers.  Accumulators -> bf16 -> image [m][n] (padded rows) -> full rows out.
    {
        const int m_l = lane & 31, nq = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[i][j][4 * rg + e];
                        if (EPI == EPI_RELUSQ) { a = fmaxf(a, 0.f); a = a * a; }
                        v[e] = a;
                    }
                    const int m = 128 * wr + 32 * j + m_l, n = 64 * wc + 32 * i + 8 * rg + nq;
                    *reinterpret_cast<uint2*>(lds + m * EROWB + n * 2) = make_uint2(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]));
                }
    }
    block_sync_lds();
    {
        // wave w sends rows 32w .. 32w+31: 2 rows (2 x 512 B) per instruction
        const size_t crow = (size_t)tm * TM, ccol = (size_t)tn * TN;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int m = 32 * wave + 2 * it + (lane >> 5), c8 = 8 * (lane & 31);
            u32x4 v = *reinterpret_cast<const u32x4*>(lds + m * EROWB + c8 * 2);
            const size_t go = (crow + m) * (size_t)p.N + ccol + c8;
            if (EPI == EPI_DRELUSQ) {
                const u32x4 h = *reinterpret_cast<const u32x4*>(p.aux + go);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = bf16_lo(v[e]) * (2.f * __builtin_sqrtf(bf16_lo(h[e]))), a1 = bf16_hi(v[e]) * (2.f * __builtin_sqrtf(bf16_hi(h[e])));
                    v[e] = cvt_pk_bf16(a0, a1);
                }
            }
            *reinterpret_cast<u32x4*>(p.C + go) = v;
        }
    }
}

}  // namespace gtn
