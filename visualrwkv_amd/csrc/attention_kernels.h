// Softmax attention forward for the frozen ViT towers (SigLIP so400m: head dim 72, DINOv2-L / SAM: 64) on the
// gfx950 matrix cores.  o = softmax(q k^T / sqrt(D)) v, no mask, any sequence length (tail keys masked).
//
// Replaces (forward only -- the towers are frozen, src/model.py:349,368) the attention inside timm's
// VisionTransformer blocks that `SamDinoSigLIPViTBackbone.forward` runs (VisualRWKV-v7/v7.00/src/vision.py:123-134)
// and `Attention.forward` of the SAM encoder without its relative-position bias (src/sam.py:289-305).
//
// One workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 queries.  Keys/values are
// streamed in tiles of 32 through LDS (K row-major, V transposed).  The score tile is computed TRANSPOSED,
// S^T = K Q^T, so a lane holds scores of ONE query (column) for 8 keys: the online-softmax max/sum are
// in-lane reductions plus two cross-row xors, and P^T in C layout is directly the B operand of
// O^T += V^T P^T (k-slots permuted to the accumulator map, as in the WKV7 kernels) -- no LDS round trip for P.
#pragma once
#include <gfx950_prims.h>

namespace vattn {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}

struct Args {
    const uint16_t *q, *k, *v;     // bf16; element (b, l, h, d) at b*sb + l*sl + h*sh + d
    uint16_t* o;                   // bf16 (B, L, H, D) contiguous
    long sb, sl, sh;               // strides of q/k/v in elements (shared by the three: slices of one qkv tensor)
    int L, H;
    float scale_log2e;             // 1/sqrt(D) * log2(e)
};

constexpr int KT = 32;             // keys per tile

template <int D>
__global__ __launch_bounds__(256) void fwd_kernel(Args p) {
    constexpr int DP = (D + 31) / 32 * 32;       // contraction length of QK^T padded to the MFMA K
    constexpr int NKB = DP / 32;
    constexpr int DT = (D + 15) / 16;            // 16-wide tiles of the head dim in O^T
    constexpr int KS = DP + 8;                   // K tile row stride (elements)
    constexpr int VS = KT + 8;                   // V^T tile row stride
    __shared__ __attribute__((aligned(16))) uint16_t kt[KT][KS];
    __shared__ __attribute__((aligned(16))) uint16_t vt[DT * 16][VS];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const long base = (long)b * p.sb + (long)h * p.sh;
    const int L = p.L;

    // Q fragment (B operand: [k = d][n = q]): lane (g, c16 = q) holds d = 32kb + 8g .. +7
    bf16x8 qf[NKB];
    {
        const int qrow = q0 + c16;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int d0 = 32 * kb + 8 * g;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (qrow < L && d0 < D) u = *reinterpret_cast<const uint4*>(p.q + base + (long)qrow * p.sl + d0);
            qf[kb] = mk8(u.x, u.y, u.z, u.w);
        }
    }
    f32x4 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    for (int k0 = 0; k0 < L; k0 += KT) {
        block_sync();                                  // previous tile fully consumed
        // stage K tile (row-major, zero padded) and V tile (transposed)
        for (int idx = tid; idx < KT * (DP / 8); idx += 256) {
            const int key = idx / (DP / 8), d0 = (idx % (DP / 8)) * 8;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (k0 + key < L && d0 < D) u = *reinterpret_cast<const uint4*>(p.k + base + (long)(k0 + key) * p.sl + d0);
            *reinterpret_cast<uint4*>(&kt[key][d0]) = u;
        }
        for (int idx = tid; idx < KT * (DT * 2); idx += 256) {
            const int key = idx % KT, d0 = (idx / KT) * 8;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (k0 + key < L && d0 < D) u = *reinterpret_cast<const uint4*>(p.v + base + (long)(k0 + key) * p.sl + d0);
            vt[d0 + 0][key] = (uint16_t)u.x; vt[d0 + 1][key] = (uint16_t)(u.x >> 16);
            vt[d0 + 2][key] = (uint16_t)u.y; vt[d0 + 3][key] = (uint16_t)(u.y >> 16);
            vt[d0 + 4][key] = (uint16_t)u.z; vt[d0 + 5][key] = (uint16_t)(u.z >> 16);
            vt[d0 + 6][key] = (uint16_t)u.w; vt[d0 + 7][key] = (uint16_t)(u.w >> 16);
        }
        block_sync();

        // S^T tiles: st[tile][r] = score(key = k0 + 16 tile + 4g + r, query = q0 + c16)
        f32x4 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint4 u = *reinterpret_cast<const uint4*>(&kt[16 * t + c16][32 * kb + 8 * g]);
                s = mfma_16x16x32_bf16(mk8(u.x, u.y, u.z, u.w), qf[kb], s);
            }
            st[t] = s;
        }
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = k0 + 16 * t + 4 * g + r < L;
                st[t][r] = valid ? st[t][r] * p.scale_log2e : -1e30f;
                mx = fmaxf(mx, st[t][r]);
            }
        mx = fmaxf(mx, lane_xor16(mx));
        mx = fmaxf(mx, lane_xor32(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[t][r] = exp2f(st[t][r] - m_new); ps += st[t][r]; }
        ps += lane_xor16(ps);
        ps += lane_xor32(ps);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // P^T as B operand: slots e<4 <-> key 4g+e of tile 0, e>=4 <-> tile 1
        const bf16x8 pf = mk8(cvt_pk_bf16(st[0][0], st[0][1]), cvt_pk_bf16(st[0][2], st[0][3]),
                              cvt_pk_bf16(st[1][0], st[1][1]), cvt_pk_bf16(st[1][2], st[1][3]));
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            f32x4 a = acc[t];
            a[0] *= alpha; a[1] *= alpha; a[2] *= alpha; a[3] *= alpha;
            const uint2 v0 = *reinterpret_cast<const uint2*>(&vt[16 * t + c16][4 * g]);
            const uint2 v1 = *reinterpret_cast<const uint2*>(&vt[16 * t + c16][16 + 4 * g]);
            acc[t] = mfma_16x16x32_bf16(mk8(v0.x, v0.y, v1.x, v1.y), pf, a);
        }
    }
    // O[q][d]: lane (g, c16 = q) holds d = 16t + 4g + r
    const int qrow = q0 + c16;
    if (qrow < L) {
        const float inv = 1.f / l_run;
        uint16_t* orow = p.o + (((long)b * L + qrow) * p.H + h) * D;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int d0 = 16 * t + 4 * g;
            if (d0 < D)
                *reinterpret_cast<uint2*>(orow + d0) = make_uint2(cvt_pk_bf16(acc[t][0] * inv, acc[t][1] * inv),
                                                                  cvt_pk_bf16(acc[t][2] * inv, acc[t][3] * inv));
        }
    }
}

}  // namespace vattn
