// Patch embedding of the ViT towers as an implicit GEMM on the gfx950 matrix cores, reading the NCHW pixels directly.
//
// Replaces timm's PatchEmbed (Conv2d with stride = kernel = P, then flatten/transpose) followed by the position
// embedding add that `SamDinoSigLIPViTBackbone.forward` runs through timm (VisualRWKV-v7/v7.00/src/vision.py:123-134)
// and `PatchEmbed.forward` + `x = x + self.pos_embed` of the SAM encoder (src/sam.py:118-121, 468-483):
//     out[b, prefix + m, n] = sum_k pixels[b, c, gy P + py, gx P + px] W[n, k] + bias[n] + pos[m, n],  k = (c, py, px)
// -- no unfolded (im2col) copy of the image, no separate bias / position kernels.
//
// One workgroup = 64 patches of one image x all N output channels.  The 64 x K patch matrix is gathered once into LDS
// (whole pixel rows, coalesced: consecutive lanes walk the patches of one grid row) and each wave lifts its 16 rows into
// registers (all K, as B operands).  The same LDS region then becomes a double buffer for the weights: chunks of 32
// output channels (32 x KP bf16, contiguous in memory) are staged global -> registers -> LDS by all four waves, one
// barrier per chunk, loads of chunk c+1 in flight during the MFMAs of chunk c; every wave reads the chunk's rows as A
// operands.  The product comes out TRANSPOSED, D[n][patch], so a lane owns 4 consecutive channels of one patch and the
// epilogue (bias + position embedding in fp32, one rounding) stores 8 bytes per lane.  Weights are pre-padded to
// KP = ceil32(K) columns of zeros.  (First version: every wave streamed the whole weight matrix from L2 as MFMA
// operands -- 4.8 MB per workgroup, 285 us for 16 x 1024 patches x 1152; staged through LDS it is one read per workgroup.)
#pragma once
#include <gfx950_prims.h>

namespace vpe {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
DEVFN bf16x8 mk8(uint4 u) {
    u32x4 v = {u.x, u.y, u.z, u.w};
    return __builtin_bit_cast(bf16x8, v);
}

struct Args {
    const uint16_t* px;      // (B, 3, Himg, Wimg) bf16
    const uint16_t* w;       // (N, KP) bf16, columns >= 3 P P zero
    const uint16_t* bias;    // (N) bf16 or null
    const uint16_t* pos;     // (Mimg, N) bf16 or null
    uint16_t* out;           // (B, Ltot, N) bf16; patch m of image b at token prefix + m
    int Himg, Wimg, N, gw, Mimg, Ltot, prefix;
};

template <int P> struct Geo {
    static constexpr int K = 3 * P * P, KP = (K + 31) / 32 * 32, NKB = KP / 32;
    static constexpr int AS = KP + 8;                       // LDS row stride (elements): KP/2 + 4 dwords = 4 mod 8
    static constexpr int LDS_BYTES = 64 * AS * 2;
};

template <int P>
__global__ __launch_bounds__(256) void kernel(Args a) {
    using G = Geo<P>;
    constexpr int K = G::K, KP = G::KP, NKB = G::NKB, AS = G::AS, HW = P / 2;
    uint16_t* A = reinterpret_cast<uint16_t*>(dyn_lds());                  // [64][AS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int tiles = a.Mimg / 64, b = blockIdx.x / tiles, m0 = (blockIdx.x % tiles) * 64;

    // gather: a thread owns up to two dwords (patch ml, dword j of the patch's pixel row) and walks the 3 P (channel,
    // pixel row) pairs: every address is a per-thread base plus a wave-uniform offset, one channel's P loads in flight
    constexpr int DPR = 64 * HW;                           // dwords per (channel, pixel row) over the 64 patches
    constexpr int SETS = (DPR + 255) / 256;
    const uint16_t* gsrc[SETS];
    uint16_t* ldst[SETS];
    bool own[SETS];
#pragma unroll
    for (int u = 0; u < SETS; ++u) {
        const int d = tid + 256 * u;
        own[u] = d < DPR;
        const int ml = own[u] ? d / HW : 0, j = own[u] ? d % HW : 0;
        const int m = m0 + ml, gy = m / a.gw, gx = m - gy * a.gw;
        gsrc[u] = a.px + ((long)b * 3 * a.Himg + gy * P) * a.Wimg + gx * P + 2 * j;
        ldst[u] = A + ml * AS + 2 * j;
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        uint32_t v[SETS][P];
#pragma unroll
        for (int py = 0; py < P; ++py)
#pragma unroll
            for (int u = 0; u < SETS; ++u)
                v[u][py] = (DPR % 256 == 0 || own[u]) ? *reinterpret_cast<const uint32_t*>(gsrc[u] + ((long)c * a.Himg + py) * a.Wimg) : 0u;
#pragma unroll
        for (int py = 0; py < P; ++py)
#pragma unroll
            for (int u = 0; u < SETS; ++u)
                if (DPR % 256 == 0 || own[u]) *reinterpret_cast<uint32_t*>(ldst[u] + c * P * P + py * P) = v[u][py];
    }
    if constexpr (KP > K)
        for (int e = tid; e < 64 * (KP - K) / 2; e += 256) {
            const int ml = e / ((KP - K) / 2), j = e % ((KP - K) / 2);
            *reinterpret_cast<uint32_t*>(A + ml * AS + K + 2 * j) = 0u;
        }
    block_sync();

    bf16x8 af[NKB];                                        // B operand: lane (g, c16 = patch) holds k = 32kb + 8g .. +7
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) af[kb] = mk8(*reinterpret_cast<const uint4*>(A + (16 * wave + c16) * AS + 32 * kb + 8 * g));
    block_sync();                                          // every wave holds its rows: the image region is free

    // ---- weights: chunks of NC output channels through the double buffer W[2][NC][AS] (same bytes as the patch image)
    constexpr int NC = 32, PPR = KP / 8, PIECES = NC * PPR, NLD = (PIECES + 255) / 256;   // 16-byte pieces per chunk / thread
    static_assert(2 * NC * AS <= 64 * AS, "weight double buffer must fit the patch image");
    uint16_t* Wb = A;
    u32x4 wr[NLD];                                        // native vector type: HIP's uint4 struct kept this array in scratch
    int wrow_[NLD], wcol_[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        wrow_[i] = idx / PPR;
        wcol_[i] = (idx % PPR) * 8;
    }
    const bool tail_ok = PIECES % 256 == 0 || tid + 256 * (NLD - 1) < PIECES;        // last piece of this thread exists
#define VPE_FETCH(c_)                                                                                        \
    {                                                                                                        \
        const uint16_t* src_ = a.w + (long)(c_) * NC * KP + 8 * tid;                                         \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i)   /* a missing last piece re-reads piece 0 (never staged) */ \
            wr[i] = *reinterpret_cast<const u32x4*>(src_ + ((i + 1 < NLD || tail_ok) ? 2048 * i : 0));      \
    }
#define VPE_STAGE(buf_)                                                                                      \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i)                                                      \
            if (i + 1 < NLD || tail_ok) *reinterpret_cast<u32x4*>(Wb + ((buf_) * NC + wrow_[i]) * AS + wcol_[i]) = wr[i]; \
    }
    const int m = m0 + 16 * wave + c16;
    uint16_t* orow = a.out + ((long)b * a.Ltot + a.prefix + m) * a.N;
    const uint16_t* prow = a.pos ? a.pos + (long)m * a.N : nullptr;
    const int nchunks = a.N / NC;
    VPE_FETCH(0)
    VPE_STAGE(0)
    if (nchunks > 1) VPE_FETCH(1)
    block_sync();
    for (int c = 0; c < nchunks; ++c) {
        const uint16_t* wc = Wb + (c & 1) * NC * AS;
#pragma unroll
        for (int nt = 0; nt < NC / 16; ++nt) {
            const uint16_t* wrow = wc + (16 * nt + c16) * AS + 8 * g;      // A operand: lane (g, c16 = channel)
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const bf16x8 wf = mk8(*reinterpret_cast<const uint4*>(wrow + 32 * kb));
                if (kb & 1) acc1 = mfma_16x16x32_bf16(wf, af[kb], acc1);
                else acc0 = mfma_16x16x32_bf16(wf, af[kb], acc0);
            }
            const int n0 = NC * c + 16 * nt + 4 * g;       // lane (g, c16 = patch) holds channels n0 .. n0+3
            float y[4] = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
            if (a.bias) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.bias + n0);
                y[0] += bf16_lo(u.x); y[1] += bf16_hi(u.x); y[2] += bf16_lo(u.y); y[3] += bf16_hi(u.y);
            }
            if (prow) {
                const uint2 u = *reinterpret_cast<const uint2*>(prow + n0);
                y[0] += bf16_lo(u.x); y[1] += bf16_hi(u.x); y[2] += bf16_lo(u.y); y[3] += bf16_hi(u.y);
            }
            *reinterpret_cast<uint2*>(orow + n0) = make_uint2(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]));
        }
        if (c + 1 < nchunks) {
            VPE_STAGE((c + 1) & 1)                         // last read in iteration c-1: everyone is past that barrier
            if (c + 2 < nchunks) VPE_FETCH(c + 2)
        }
        block_sync();
    }
}

#undef VPE_FETCH
#undef VPE_STAGE

}  // namespace vpe
