// Weight gradient of a skinny projection: C[Nw x D] = Wide^T Narrow, Wide (M x Nw) and Narrow (M x D) bf16 row-major,
// M = all tokens of the micro-batch (~42 k), Nw = 2048, D = a LoRA rank (64 / 96 / 256) -- gfx950.
//
// These are the 8 weight-gradient products per layer of RWKV_Tmix_x070's LoRA pairs (VisualRWKV-v7/v7.00/src/model.py:
// 176,181-183: w1/w2, a1/a2, v1/v2, g1/g2; dW1 = x^T dH with Wide = x, dW2 = h^T dOut = (dOut^T h)^T with Wide = dOut).
// Both operands have the reduction index M as their slow index and one of them is only D wide: the library's kernels
// for the shape run at 105-150 us (165 us inside the step) against a floor of ~35 us for reading Wide once.
//
// One workgroup = 128 columns of Wide x all D columns of Narrow (two groups of 128 for D = 256: the accumulators of 256
// columns do not fit the register file at two waves per SIMD) x one contiguous slice of M (grid: Nw/128 [x 2] x S).
// Per step of 32 rows: the two tiles go global -> registers -> LDS as they lie (16-byte accesses, the [m][col] images are
// never transposed by software), and both MFMA operands are fetched with ds_read_b64_tr_b16, which hands a lane 4
// consecutive rows (m) of one column -- the k index of v_mfma_f32_16x16x32_bf16 runs along m for A and for B alike
// (operand slot (g, e) <-> m = 4g + e for e < 4, 16 + 4g + e - 4 above; the same permutation on both sides).
// Loads run two steps ahead of the LDS stores (two register stages), three workgroups per CU keep ~80 KB in flight.
// fp32 partial tiles per M-slice are summed (and rounded to bf16, optionally transposed) by reduce_kernel.
#pragma once
#include <gfx950_prims.h>

namespace lwg {

constexpr int CT = 128;             // columns of Wide per workgroup (4 waves x 2 tiles of 16)
constexpr int KS = 32;              // rows of M per step = K of one MFMA
constexpr int WS = CT + 8;          // LDS row strides in elements (272 B / 2 D + 16 B: rows start 4 banks apart)

struct Args {
    long M;
    int Nw, D;
    const uint16_t* wide;
    const uint16_t* narrow;
    float* part;                    // [S][Nw][D] fp32
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint2 lo, uint2 hi) {
    u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}
// staged pieces are the NATIVE vector type: arrays of HIP's uint4 struct were kept in scratch memory by the compiler
DEVFN u32x4 ld16(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
DEVFN void st16(uint16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

template <int ND>
__global__ __launch_bounds__(256, (ND <= 6 ? 3 : 2)) void wgrad_kernel(Args p) {       // 3 (2) workgroups per CU: register cap
    constexpr int D = 16 * ND, NS = D + 8;           // D: Narrow columns of THIS workgroup (p.D / D column groups)
    constexpr int NCH = KS * D / 8;                  // 16-byte chunks of a Narrow tile
    constexpr int NL = (NCH + 255) / 256;            // ... per thread
    uint16_t* lds = reinterpret_cast<uint16_t*>(dyn_lds());
    uint16_t (*wt)[KS][WS] = reinterpret_cast<uint16_t (*)[KS][WS]>(lds);                  // [2]
    uint16_t (*nt)[KS][NS] = reinterpret_cast<uint16_t (*)[KS][NS]>(lds + 2 * KS * WS);    // [2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
    const int ngrp = p.D / D, grp = blockIdx.x % ngrp;           // neighbouring workgroups share a Wide tile (L2 / MALL)
    const int col0 = (blockIdx.x / ngrp) * CT, d0 = grp * D;
    const long nsteps = (p.M + KS - 1) / KS;
    const long t0 = nsteps * blockIdx.y / gridDim.y, t1 = nsteps * (blockIdx.y + 1) / gridDim.y;

    struct Stage { u32x4 w[2]; u32x4 n[NL]; };
    const u32x4 zero = {0u, 0u, 0u, 0u};
    auto fetch = [&](Stage& s, long t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i, row = q >> 4, cc = q & 15;
            const long m = t * KS + row;
            s.w[i] = m < p.M ? ld16(p.wide + m * p.Nw + col0 + 8 * cc) : zero;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i, row = q / (D / 8), cc = q % (D / 8);
            const long m = t * KS + row;
            s.n[i] = (q < NCH && m < p.M) ? ld16(p.narrow + m * p.D + d0 + 8 * cc) : zero;
        }
    };
    auto stash = [&](const Stage& s, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i;
            st16(&wt[buf][q >> 4][8 * (q & 15)], s.w[i]);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = tid + 256 * i;
            if (q < NCH) st16(&nt[buf][q / (D / 8)][8 * (q % (D / 8))], s.n[i]);
        }
    };
    f32x4 acc[2][ND];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int n = 0; n < ND; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[ct][n] = z; }
    const int tr = 4 * g + (c16 >> 2), tc = 4 * (c16 & 3);      // ds_read_b64_tr_b16 addressing inside a 16 x 16 block
    auto compute = [&](int buf) {
        bf16x8 a[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int cb = 16 * (2 * wave + ct) + tc;
            a[ct] = mk8(lds_read_tr16(&wt[buf][tr][cb]), lds_read_tr16(&wt[buf][16 + tr][cb]));
        }
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const bf16x8 b = mk8(lds_read_tr16(&nt[buf][tr][16 * n + tc]), lds_read_tr16(&nt[buf][16 + tr][16 * n + tc]));
            acc[0][n] = mfma_16x16x32_bf16(a[0], b, acc[0][n]);
            acc[1][n] = mfma_16x16x32_bf16(a[1], b, acc[1][n]);
        }
    };

    if (t0 < t1) {
        Stage sa, sb;
        fetch(sa, t0);
        fetch(sb, t0 + 1);                           // steps past the slice are fetched (zeros past M) and stashed but
        stash(sa, 0);                                // never multiplied: see the guards on compute below
        fetch(sa, t0 + 2);
        block_sync_lds();
        for (long t = t0; t < t1; t += 2) {
            compute(0);                              // step t from buffer 0; sb = step t+1, sa = step t+2 (in flight)
            stash(sb, 1);
            fetch(sb, t + 3);
            block_sync_lds();
            if (t + 1 < t1) compute(1);              // step t+1 from buffer 1; sa = step t+2, sb = step t+3 (in flight)
            stash(sa, 0);
            fetch(sa, t + 4);
            block_sync_lds();
        }
    }
    float* out = p.part + (size_t)blockIdx.y * p.Nw * p.D + d0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(size_t)(col0 + 16 * (2 * wave + ct) + 4 * g + r) * p.D + 16 * n + c16] = acc[ct][n][r];
}

// out (bf16) = sum over the S slices; transposed: out[d][nw] instead of out[nw][d].  One thread per 4 consecutive
// elements (D % 16 == 0: they share a row), four independent partial sums so that the S loads overlap.
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int S, int Nw, int D, int transposed,
                                                     uint16_t* __restrict__ out) {
    const long n4 = (long)Nw * D / 4;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= S; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = p4[(size_t)(k + u) * n4 + i];
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
        for (; k < S; ++k) {
            const float4 v = p4[(size_t)k * n4 + i];
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
        const float s[4] = {(acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                            (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)};
        const long e = 4 * i;
        if (!transposed) {
            *reinterpret_cast<uint2*>(out + e) = make_uint2(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]));
        } else {
            const long nw = e / D, d = e % D;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(d + j) * Nw + nw] = (uint16_t)f32_to_bf16_bits(s[j]);
        }
    }
}

}  // namespace lwg
