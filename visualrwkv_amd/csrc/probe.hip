// Hardware probes used by the GPU tests: check the MFMA lane->element maps documented in
// gfx950_prims.h (and modelled by tests/emu/gfx950_prims.h) on the real chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

// D[M][N] = A[M][K] * B[K][N], one wave, operands row-major f32 in global memory.
__global__ void probe_16x16x4_f32(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma_16x16x4_f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void probe_32x32x2_f32(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = mfma_32x32x2_f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void probe_16x16x32_bf16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l >> 4) * 8 + e;
        a[e] = (short)f32_to_bf16_bits(A[(l & 15) * 32 + k]);
        b[e] = (short)f32_to_bf16_bits(B[k * 16 + (l & 15)]);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma_16x16x32_bf16(a, b, c);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void probe_32x32x16_bf16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l >> 5) * 8 + e;
        a[e] = (short)f32_to_bf16_bits(A[(l & 31) * 16 + k]);
        b[e] = (short)f32_to_bf16_bits(B[k * 32 + (l & 31)]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = mfma_32x32x16_bf16(a, b, c);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// cross-lane primitives: out[0..63] = group_sum<4>(in), out[64..] = half_mirror, mirror, xor16, xor32
__global__ void probe_lanes(const float* in, float* out) {
    const int l = threadIdx.x;
    const float x = in[l];
    out[l] = group_sum<4>(x);
    out[64 + l] = lane_half_mirror(x);
    out[128 + l] = lane_mirror(x);
    out[192 + l] = lane_xor(x, 16);
    out[256 + l] = lane_xor(x, 32);
    out[320 + l] = group_sum<6>(x);
    out[384 + l] = lane_xor16(x);
    out[448 + l] = lane_xor32(x);
    out[512 + l] = quad_perm<0x90>(x);
    out[576 + l] = quad_perm<0xF9>(x);
    { f32x4 v = {x, x + 100.f, x + 200.f, x + 300.f}; v = quad_transpose(v); out[640 + l] = v[0]; out[704 + l] = v[1]; out[768 + l] = v[2]; out[832 + l] = v[3]; }
}

// D[16][16] = A[16][16] * B[16][16] with the K=16 bf16 MFMA
__global__ void probe_16x16x16_bf16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    bf16x4 a, b;
    for (int e = 0; e < 4; ++e) {
        const int k = (l >> 4) * 4 + e;
        a[e] = (short)f32_to_bf16_bits(A[(l & 15) * 16 + k]);
        b[e] = (short)f32_to_bf16_bits(B[k * 16 + (l & 15)]);
    }
    f32x4 d = mfma_16x16x16_bf16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = d[r];
}

// Issue rate of the bf16 MFMAs on one SIMD: NCH independent accumulator chains, 512 rounds; cycles per MFMA -> D[0..5]
template <int K, int NCH>
DEVFN float mfma_rate(int l) {
    f32x4 acc[NCH];
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8; bf16x4 a4;
    for (int e = 0; e < 8; ++e) a8[e] = (short)(0x3f80 + l + e);
    for (int e = 0; e < 4; ++e) a4[e] = (short)(0x3f80 + l + e);
    const long t0 = clock64_();
    for (int it = 0; it < 512; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (K == 32) acc[c] = mfma_16x16x32_bf16(a8, a8, acc[c]);
            else acc[c] = mfma_16x16x16_bf16(a4, a4, acc[c]);
        }
    }
    float sink = 0.f;
    for (int c = 0; c < NCH; ++c) sink += acc[c][0];
    const long t1 = clock64_();
    return (float)(t1 - t0) / (512.f * NCH) + (sink == 12345.f ? 1.f : 0.f);
}
__global__ void probe_mfma_rate(float* D) {
    const int l = threadIdx.x;
    const float r0 = mfma_rate<32, 1>(l), r1 = mfma_rate<32, 4>(l), r2 = mfma_rate<16, 1>(l), r3 = mfma_rate<16, 4>(l);
    const float r4 = mfma_rate<16, 8>(l), r5 = mfma_rate<32, 8>(l);
    if (l == 0) { D[0] = r0; D[1] = r1; D[2] = r2; D[3] = r3; D[4] = r4; D[5] = r5; }
}

// ds_read_b64_tr_b16: in = 64 x 80 floats (integers < 32768) copied to a [64][80] u16 LDS image; lane l points at row
// 4 (l>>4) + ((l&15)>>2), columns 4 (l&3)..+3; out[l*4 + e] = what the lane received.
__global__ void probe_tr16(const float* in, float* out) {
    __shared__ __attribute__((aligned(16))) uint16_t img[64][80];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 80; i += 64) img[i / 80][i % 80] = (uint16_t)in[i];
    block_sync();
    const uint2 v = lds_read_tr16(&img[4 * (l >> 4) + ((l & 15) >> 2)][4 * (l & 3)]);
    out[l * 4 + 0] = (float)(v.x & 0xffff); out[l * 4 + 1] = (float)(v.x >> 16);
    out[l * 4 + 2] = (float)(v.y & 0xffff); out[l * 4 + 3] = (float)(v.y >> 16);
}

// LDS-DMA with an immediate offset (lds_dma16_lean<1024>): in = 1024 floats; every lane requests in[4 l .. 4 l + 3] (+ the immediate)
// into an LDS image poisoned with -1; out = the first 1024 floats of the image.  The immediate moves the global AND the LDS address.
__global__ void probe_dma_imm(const float* in, float* out) {
    __shared__ __attribute__((aligned(16))) float img[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) img[i] = -1.f;
    block_sync();
    lds_dma16_lean<1024>(in, (unsigned)l * 16u, lds_addr_u32(img));
    vmem_drain();
    block_sync();
    for (int i = l; i < 1024; i += 64) out[i] = img[i];
}

}  // namespace

// Streaming-copy ceiling of the box (the WKV roofline fraction is reported against the vendor HBM peak and
// against what a plain copy reaches).  One 16-byte vector per thread, plain loads and stores, as many workgroups as vectors / 256
// -- the form the MI355X guide quotes 6.29 TB/s for.  On the boxes of this pool it reaches 6.2 TB/s where the tiled form used
// until round 3 (32 KB per workgroup, 8 non-temporal loads in flight per lane, 2048 workgroups) reached 5.3-5.7 and a grid-stride
// loop over 1024-16384 workgroups 4.6-5.7 (benchmarks/mem_role_probe.hip, profiles/r4_mem_role_probe.jsonl): the "copy ceiling" of
// rounds 1-3 measured that kernel, not the box.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) dst[i] = src[i];
}

// Streaming probes with other read : write mixes than the copy's 1 : 1 (the WKV7 forward writes 22 of its 34 B/element, the
// backward reads 34 of its 46): mode 1 = fill (0 : 1), 2 = one read, two writes (1 : 2), 3 = read only (1 : 0; a value that
// cannot occur keeps the loads alive), 4 = two reads, one write (2 : 1).  Same tiling as the copy kernel.
template <int MODE>
__global__ __launch_bounds__(256) void stream_mix_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, u32x4_t* __restrict__ d0,
                                                         u32x4_t* __restrict__ d1, long nvec) {
    constexpr int U = 8;
    const long ntiles = (nvec + 256 * U - 1) / (256 * U);
    u32x4_t keep = {0u, 0u, 0u, 0u};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long base = t * 256 * U + threadIdx.x;
        u32x4_t v[U], w[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (base + q * 256 >= nvec) continue;
            if (MODE != 1) v[q] = __builtin_nontemporal_load(a + base + q * 256); else v[q] = keep;
            if (MODE == 4) w[q] = __builtin_nontemporal_load(b + base + q * 256);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (base + q * 256 >= nvec) continue;
            if (MODE == 4) v[q] ^= w[q];
            if (MODE == 3) { keep |= v[q]; continue; }
            __builtin_nontemporal_store(v[q], d0 + base + q * 256);
            if (MODE == 2) __builtin_nontemporal_store(v[q], d1 + base + q * 256);
        }
    }
    if (MODE == 3 && keep[0] == 0x12345678u && keep[1] == 0x9abcdef0u) d0[0] = keep;
}

// bytes = size of ONE array; returns 0 / error.  a, b: sources; d0, d1: destinations (unused ones may alias).
extern "C" int vrwkv_stream_probe(int mode, const void* a, const void* b, void* d0, void* d1, long bytes, void* stream) {
    if (!a || !d0 || bytes <= 0 || bytes % 16 != 0) return VRWKV_EINVAL;
    const dim3 grid(256 * 8), block(256);
    const long nvec = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 1: hipLaunchKernelGGL(stream_mix_kernel<1>, grid, block, 0, st, (const u32x4_t*)a, (const u32x4_t*)b, (u32x4_t*)d0, (u32x4_t*)d1, nvec); break;
        case 2: hipLaunchKernelGGL(stream_mix_kernel<2>, grid, block, 0, st, (const u32x4_t*)a, (const u32x4_t*)b, (u32x4_t*)d0, (u32x4_t*)d1, nvec); break;
        case 3: hipLaunchKernelGGL(stream_mix_kernel<3>, grid, block, 0, st, (const u32x4_t*)a, (const u32x4_t*)b, (u32x4_t*)d0, (u32x4_t*)d1, nvec); break;
        case 4: hipLaunchKernelGGL(stream_mix_kernel<4>, grid, block, 0, st, (const u32x4_t*)a, (const u32x4_t*)b, (u32x4_t*)d0, (u32x4_t*)d1, nvec); break;
        default: return VRWKV_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

extern "C" int vrwkv_stream_copy(const void* src, void* dst, long bytes, void* stream) {
    if (!src || !dst || bytes <= 0) return VRWKV_EINVAL;
    if (bytes % 16 != 0) return VRWKV_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) return VRWKV_EALIGN;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (u32x4_t*)dst, bytes / 16);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

extern "C" int vrwkv_debug_probe(int which, const float* a, const float* b, float* d, void* stream) {
    if (!a || !d) return VRWKV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(probe_16x16x4_f32, dim3(1), dim3(64), 0, st, a, b, d); break;
        case 1: hipLaunchKernelGGL(probe_32x32x2_f32, dim3(1), dim3(64), 0, st, a, b, d); break;
        case 2: hipLaunchKernelGGL(probe_16x16x32_bf16, dim3(1), dim3(64), 0, st, a, b, d); break;
        case 3: hipLaunchKernelGGL(probe_32x32x16_bf16, dim3(1), dim3(64), 0, st, a, b, d); break;
        case 4: hipLaunchKernelGGL(probe_lanes, dim3(1), dim3(64), 0, st, a, d); break;
        case 5: hipLaunchKernelGGL(probe_16x16x16_bf16, dim3(1), dim3(64), 0, st, a, b, d); break;
        case 6: hipLaunchKernelGGL(probe_mfma_rate, dim3(1), dim3(64), 0, st, d); break;
        case 7: hipLaunchKernelGGL(probe_tr16, dim3(1), dim3(64), 0, st, a, d); break;
        case 8: hipLaunchKernelGGL(probe_dma_imm, dim3(1), dim3(64), 0, st, a, d); break;
        default: return VRWKV_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}
