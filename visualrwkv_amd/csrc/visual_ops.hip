// Image side of the hot path between the vision towers and the language model (VisualRWKV-v7/v7.00/src/model.py):
//   adaptive average pooling of the ViT patch grid to sqrt(num_token_per_image)^2 tokens   (:354,442-447)
//   x * sigmoid(gate(x)) of MLPWithContextGating                                            (:336-338)
// Streaming, HBM-bound kernels: a thread owns 8 consecutive channels (16-byte accesses), fp32 arithmetic, one rounding.
// (ln_v + the masked scatter into the token embeddings are vrwkv_ln_scatter_* in ln_fused.hip.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/visualrwkv_hip.h"
#include <gfx950_prims.h>

namespace {

DEVFN void unpack8f(uint4 u, float* f) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
DEVFN uint4 pack8f(const float* f) {
    return make_uint4(cvt_pk_bf16(f[0], f[1]), cvt_pk_bf16(f[2], f[3]), cvt_pk_bf16(f[4], f[5]), cvt_pk_bf16(f[6], f[7]));
}

// nn.AdaptiveAvgPool2d on a token-major feature map: x (B, Sin*Sin, D) -> y (B, Sout*Sout, D); output cell (oy, ox)
// averages rows floor(oy Sin / Sout) .. ceil((oy+1) Sin / Sout) - 1 (same for columns), PyTorch's window rule, which
// also covers Sout > Sin (2304 tokens from a 32 x 32 grid: windows of one cell).  One workgroup per output token.
__global__ __launch_bounds__(256) void adaptive_pool_kernel(int Sin, int Sout, int D, const uint16_t* __restrict__ x, uint16_t* __restrict__ y) {
    const int tok = blockIdx.x % (Sout * Sout), b = blockIdx.x / (Sout * Sout);
    const int oy = tok / Sout, ox = tok % Sout;
    const int y0 = (oy * Sin) / Sout, y1 = ((oy + 1) * Sin + Sout - 1) / Sout;
    const int x0 = (ox * Sin) / Sout, x1 = ((ox + 1) * Sin + Sout - 1) / Sout;
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    const uint16_t* xb = x + (size_t)b * Sin * Sin * D;
    uint16_t* yo = y + (size_t)blockIdx.x * D;
    for (int c0 = threadIdx.x * 8; c0 < D; c0 += blockDim.x * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                float f[8];
                unpack8f(*reinterpret_cast<const uint4*>(xb + (size_t)(iy * Sin + ix) * D + c0), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
        *reinterpret_cast<uint4*>(yo + c0) = pack8f(acc);
    }
}

DEVFN float sigmoidf_(float g) { return fast_rcp(1.f + fast_exp(-g)); }

// out = x * sigmoid(g)
__global__ __launch_bounds__(256) void gate_fwd_kernel(long nvec, const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       uint16_t* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float xf[8], gf[8], o[8];
        unpack8f(reinterpret_cast<const uint4*>(x)[i], xf);
        unpack8f(reinterpret_cast<const uint4*>(g)[i], gf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xf[e] * sigmoidf_(gf[e]);
        reinterpret_cast<uint4*>(out)[i] = pack8f(o);
    }
}
// dg = dout * x * s (1 - s), and (optionally) dx = dout * s       (s = sigmoid(g))
__global__ __launch_bounds__(256) void gate_bwd_kernel(long nvec, const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       const uint16_t* __restrict__ dout, uint16_t* __restrict__ dg,
                                                       uint16_t* __restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float xf[8], gf[8], df[8], o[8], ox[8];
        unpack8f(reinterpret_cast<const uint4*>(x)[i], xf);
        unpack8f(reinterpret_cast<const uint4*>(g)[i], gf);
        unpack8f(reinterpret_cast<const uint4*>(dout)[i], df);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = sigmoidf_(gf[e]);
            o[e] = df[e] * xf[e] * s * (1.f - s);
            ox[e] = df[e] * s;
        }
        reinterpret_cast<uint4*>(dg)[i] = pack8f(o);
        if (dx) reinterpret_cast<uint4*>(dx)[i] = pack8f(ox);
    }
}

// nn.GELU of the towers' MLPs (timm Mlp via src/vision.py:123-134; src/sam.py MLPBlock), one 16-byte vector per thread (the launch shape of relusq_*).
// TANH (SigLIP's gelu_tanh): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as x * sigmoid(2u): one v_exp and one v_rcp.
// exact (DINOv2, SAM): 0.5 x (1 + erf(x / sqrt 2)) with erfc(|z|) = t exp(-z^2 + P9(t)), t = 1 / (1 + |z| / 2) (the Chebyshev fit of Numerical Recipes 6.2:
// fractional error < 1.2e-7 for every z, so the negative tail keeps its relative accuracy) -- bf16 keeps 8 bits; PyTorch's kernel (ocml erff, ~45 VALU
// operations per element) is VALU-bound at 2.3 TB/s on this GEMM output.
template <bool TANH>
__global__ __launch_bounds__(256) void gelu_kernel(long nvec, const uint16_t* __restrict__ x, uint16_t* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    float f[8], o[8];
    unpack8f(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = f[e];
        if (TANH) {
            const float u2 = 1.5957691216057308f * v * fmaf(0.044715f * v, v, 1.f);      // 2u
            o[e] = v * fast_rcp(1.f + fast_exp(-u2));
        } else {
            const float z = fabsf(v) * 0.7071067811865476f;
            const float t = fast_rcp(fmaf(0.5f, z, 1.f));
            float pl = fmaf(t, 0.17087277f, -0.82215223f);
            pl = fmaf(t, pl, 1.48851587f); pl = fmaf(t, pl, -1.13520398f); pl = fmaf(t, pl, 0.27886807f); pl = fmaf(t, pl, -0.18628806f);
            pl = fmaf(t, pl, 0.09678418f); pl = fmaf(t, pl, 0.37409196f); pl = fmaf(t, pl, 1.00002368f); pl = fmaf(t, pl, -1.26551223f);
            const float q = t * fast_exp(fmaf(-z, z, pl));                                 // erfc(|z|)
            o[e] = 0.5f * v * (v >= 0.f ? 2.f - q : q);
        }
    }
    reinterpret_cast<uint4*>(y)[i] = pack8f(o);
}

int grid_for(long nvec) {
    long b = (nvec + 255) / 256;
    return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" {

int vrwkv_adaptive_pool_bf16(int B, int side_in, int side_out, int D, const void* x, void* y, void* stream) {
    if (B <= 0 || side_in <= 0 || side_out <= 0 || !x || !y) return VRWKV_EINVAL;
    if (D <= 0 || D % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(adaptive_pool_kernel, dim3((unsigned)(B * side_out * side_out)), dim3(256), 0, (hipStream_t)stream,
                       side_in, side_out, D, (const uint16_t*)x, (uint16_t*)y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_gate_fwd_bf16(long n, const void* x, const void* g, void* out, void* stream) {
    if (n <= 0 || !x || !g || !out) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n >> 3, (const uint16_t*)x,
                       (const uint16_t*)g, (uint16_t*)out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_gate_bwd_bf16(long n, const void* x, const void* g, const void* dout, void* dg, void* dx, void* stream) {
    if (n <= 0 || !x || !g || !dout || !dg) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, n >> 3, (const uint16_t*)x,
                       (const uint16_t*)g, (const uint16_t*)dout, (uint16_t*)dg, (uint16_t*)dx);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

int vrwkv_gelu_bf16(long n, const void* x, void* y, int tanh_approx, void* stream) {
    if (n <= 0 || !x || !y) return VRWKV_EINVAL;
    if (n % 8 != 0) return VRWKV_ESHAPE;
    const long nvec = n / 8, blocks = (nvec + 255) / 256;
    if (blocks > 0x7fffffffL) return VRWKV_ESHAPE;
    if (tanh_approx) hipLaunchKernelGGL(gelu_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nvec, (const uint16_t*)x, (uint16_t*)y);
    else hipLaunchKernelGGL(gelu_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nvec, (const uint16_t*)x, (uint16_t*)y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // extern "C"
