// WKV7 forward in chunked (matmul) form on bf16 MFMA with split operands -- gfx950.
//
// Same operator as wkv7_kernels.h (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-52) but the
// 16 tokens of a checkpoint chunk are processed together so that all work is dense 16x16 tiles
// (oracle/wkv7_chunked.py is the CPU statement of this algorithm and its validation):
//     c_t = prod_{r<=t} w_r ;  Zt = z*c_{t-1}  Qt = q*c_t  Ah = a/c_t  Kh = k/c_t
//     M_zk = tril_(Zt Kh^T)  M_qa = tril(Qt Ah^T)  M_qk = tril(Qt Kh^T)  T = (I - tril_(Zt Ah^T))^-1
//     SA = T (Zt S0^T + M_zk V) ;  Y = Qt S0^T + M_qa SA + M_qk V
//     S_L = S0 diag(c_L) + SA^T (Ah c_L) + V^T (Kh c_L)
// Precision: every fp32 operand x is split into bf16 hi = rn(x), lo = rn(x - hi) and a product is
// hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation ("bf16x3", ~2^-16 relative);
// plain bf16 products miss the 1e-3 parity bar, bf16x3 lands at ~1.5e-4 (tests/test_chunked_numerics.py).
//
// This header holds what the kernels share (constants, operand loads, the hi/lo split, phase stamps); the kernels are
// wkv7_fwd_v3.h, wkv7_bwd_v5.h, wkv7_bwd_v6.h and wkv6_chunked.h.
#pragma once
#include <gfx950_prims.h>
#include <wkv7_kernels.h>

namespace wkv7c {

using wkv7::FwdArgs;
constexpr int N = 64, L = 16;
constexpr int TJ = 68;   // [t][j] row stride in bf16 elements (136 B: conflict-free 8-byte column reads).  72 (144 B) looks
                         // better on paper for the 16-byte reads but measured 4-12 % slower in all five kernels.
constexpr int JT = 24;   // [j][t] row stride (48 B)
constexpr int SS = 24;   // [t][s] row stride (48 B)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
DEVFN bf16x8 mk8(uint2 lo, uint2 hi) { return mk8(lo.x, lo.y, hi.x, hi.y); }
DEVFN f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

struct RawChunk { uint2 w, q, k, z, a, v; };

DEVFN uint2 ld8(const uint16_t* p) { VRWKV_LDS_TRACE(1, p) return *reinterpret_cast<const uint2*>(p); }
DEVFN void st8(uint16_t* p, uint2 v) { VRWKV_LDS_TRACE(5, p) *reinterpret_cast<uint2*>(p) = v; }

// split 4 consecutive values -> packed hi (2 dwords) and lo (2 dwords)
DEVFN void split4(const float* x, uint2& hi, uint2& lo) {
    split_pk(x[0], x[1], hi.x, lo.x);
    split_pk(x[2], x[3], hi.y, lo.y);
}
DEVFN void split4(f32x4 x, uint2& hi, uint2& lo) {
    split_pk(x[0], x[1], hi.x, lo.x);
    split_pk(x[2], x[3], hi.y, lo.y);
}

DEVFN void st_b16x4_T(uint16_t (*M)[JT], int row0, int col, uint2 v) {   // 4 values -> M[row0+e][col]
    M[row0 + 0][col] = (uint16_t)v.x; M[row0 + 1][col] = (uint16_t)(v.x >> 16);
    M[row0 + 2][col] = (uint16_t)v.y; M[row0 + 3][col] = (uint16_t)(v.y >> 16);
}

DEVFN void unpack4(uint2 u, float* f) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
}

// natural-k operand: row `row`, elements j = 32kb + 8g + e
DEVFN bf16x8 ld_nat(const uint16_t (*M)[TJ], int row, int kb, int g) {
    const uint16_t* p = &M[row][32 * kb + 8 * g];
    return mk8(ld8(p), ld8(p + 4));
}
// permuted-k operand for products against the S accumulators:
// slot (g, e<4) <-> j = 32kb + 4g + e ; slot (g, e>=4) <-> j = 32kb + 16 + 4g + (e-4)
DEVFN bf16x8 ld_perm(const uint16_t (*M)[TJ], int row, int kb, int g) {
    const uint16_t* p = &M[row][32 * kb + 4 * g];
    return mk8(ld8(p), ld8(p + 16));
}

// PROF: per-phase shader-clock deltas are accumulated in registers (s_memtime is scalar) and written to
// p.dbg once at kernel end by WKV_STAMP_FLUSH -- a stamp that touched memory would bill its own latency to the
// next phase.
#define WKV_STAMP_DECL unsigned long long tprev_ = PROF ? clock64_() : 0ull, tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WKV_STAMP(slot)                                                        \
    if (PROF) {                                                                \
        const unsigned long long now_ = clock64_();                            \
        tacc_[slot] += now_ - tprev_;                                          \
        tprev_ = now_;                                                         \
    }
#define WKV_STAMP_FLUSH(who, base, n)                                          \
    if (PROF && blockIdx.x == 0 && tid == (who)) {                             \
        for (int i_ = 0; i_ < (n); ++i_) p.dbg[(base) + i_] = tacc_[i_];       \
    }



}  // namespace wkv7c
