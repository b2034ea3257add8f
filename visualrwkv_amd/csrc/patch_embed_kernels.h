// Patch embedding of the ViT towers as an implicit GEMM on the gfx950 matrix cores, reading the NCHW pixels directly.
//
// Replaces timm's PatchEmbed (Conv2d with stride = kernel = P, then flatten/transpose) followed by the position
// embedding add that `SamDinoSigLIPViTBackbone.forward` runs through timm (VisualRWKV-v7/v7.00/src/vision.py:123-134)
// and `PatchEmbed.forward` + `x = x + self.pos_embed` of the SAM encoder (src/sam.py:118-121, 468-483):
//     out[b, prefix + m, n] = sum_k pixels[b, c, gy P + py, gx P + px] W[n, k] + bias[n] + pos[m, n],  k = (c, py, px)
// -- no unfolded (im2col) copy of the image, no separate bias / position kernels.
//
// One workgroup = 64 patches of one image x all N output channels.  The 64 x K patch matrix is gathered once into LDS
// (whole pixel rows, coalesced: consecutive lanes walk the patches of one grid row), each wave lifts its 16 rows into
// registers (all K, as B operands) and then streams the weight rows straight from L2 as A operands: the product comes
// out TRANSPOSED, D[n][patch], so a lane owns 4 consecutive channels of one patch and the epilogue (bias + position
// embedding in fp32, one rounding) stores 8 bytes per lane.  Weights are pre-padded to KP = ceil32(K) columns of zeros.
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

    // gather: dword e -> (channel, pixel row py, patch ml, dword j of that patch's pixel row)
    for (int e = tid; e < 3 * P * 64 * HW; e += 256) {
        const int j = e % HW, ml = (e / HW) % 64, rest = e / (HW * 64), py = rest % P, c = rest / P;
        const int m = m0 + ml, gy = m / a.gw, gx = m - gy * a.gw;
        const uint32_t v = *reinterpret_cast<const uint32_t*>(a.px + (((long)b * 3 + c) * a.Himg + gy * P + py) * a.Wimg + gx * P + 2 * j);
        *reinterpret_cast<uint32_t*>(A + ml * AS + c * P * P + py * P + 2 * j) = v;
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

    const int m = m0 + 16 * wave + c16;
    uint16_t* orow = a.out + ((long)b * a.Ltot + a.prefix + m) * a.N;
    const uint16_t* prow = a.pos ? a.pos + (long)m * a.N : nullptr;
    for (int nt = 0; nt < a.N / 16; ++nt) {
        const uint16_t* wrow = a.w + (long)(16 * nt + c16) * KP + 8 * g;   // A operand: lane (g, c16 = channel)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const bf16x8 wf = mk8(*reinterpret_cast<const uint4*>(wrow + 32 * kb));
            if (kb & 1) acc1 = mfma_16x16x32_bf16(wf, af[kb], acc1);
            else acc0 = mfma_16x16x32_bf16(wf, af[kb], acc0);
        }
        const int n0 = 16 * nt + 4 * g;                    // lane (g, c16 = patch) holds channels n0 .. n0+3
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
}

}  // namespace vpe
