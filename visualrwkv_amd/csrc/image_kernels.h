// Tower image transform on the device: antialiased bicubic resize of one decoded uint8 HWC image to S x S, clip to the
// 8-bit range, normalise with (mean, std), planar CHW output -- in one pass over the source window of every output pixel.
//
// Replaces, per image and tower, what the reference's single DataLoader worker does on the CPU with PIL / torchvision:
// `Resize((S,S), bicubic)`, `ToTensor`, `Normalize(mean, std)` (VisualRWKV-v7/v7.00/src/vision.py:96-121; S = 448, 448, 1024).
// Resampling follows PIL's / torch's antialias convention: support = 2 max(scale, 1) source pixels around the centre
// scale (i + 0.5), Keys cubic (a = -0.5) evaluated at (x - centre + 0.5) / max(scale, 1), window clipped to the image and
// the weights renormalised.  The two weight tables of a workgroup's 32 x 8 output pixels are built once in LDS.
//
// HBM-bound byte work: every source byte is read once from HBM (windows overlap in L2 / L1), 3 S^2 outputs written once.
#pragma once
#include <gfx950_prims.h>

namespace vimg {

struct Args {
    const uint8_t* src;      // (H, W, 3) uint8
    void* dst;               // (3, S, S) bf16 or fp32
    int H, W, S, taps_x, taps_y, out_f32;
    float scale_x, scale_y;  // W / S, H / S
    float mul[3], add[3];    // y = clip(v, 0, 255) * mul[c] + add[c]   (mul = 1 / (255 std), add = -mean / std)
};

DEVFN float keys_cubic(float x) {                    // a = -0.5
    x = fabsf(x);
    if (x < 1.f) return (1.5f * x - 2.5f) * x * x + 1.f;
    if (x < 2.f) return ((-0.5f * x + 2.5f) * x - 4.f) * x + 2.f;
    return 0.f;
}

// window of output index i: first source index and tap count (torch's upsample_bicubic2d_aa / PIL's precompute_coeffs)
DEVFN void window(int i, float scale, int n_in, int& lo, int& cnt, float& centre, float& inv) {
    const float support = scale >= 1.f ? 2.f * scale : 2.f;
    inv = scale >= 1.f ? 1.f / scale : 1.f;
    centre = scale * ((float)i + 0.5f);
    lo = (int)(centre - support + 0.5f);
    if (lo < 0) lo = 0;
    int hi = (int)(centre + support + 0.5f);
    if (hi > n_in) hi = n_in;
    cnt = hi - lo;
}

constexpr int BX = 32, BY = 8;

__global__ __launch_bounds__(256) void resize_normalize_kernel(Args a) {
    float* tab = reinterpret_cast<float*>(dyn_lds());            // wx[BX][taps_x], wy[BY][taps_y]
    int* meta = reinterpret_cast<int*>(tab + BX * a.taps_x + BY * a.taps_y);      // lo/cnt of the BX columns, then the BY rows
    const int tid = threadIdx.x, tx = tid % BX, ty = tid / BX;
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
    if (tid < BX + BY) {                                          // one thread per table row
        const bool isx = tid < BX;
        const int i = isx ? x0 + tid : y0 + (tid - BX);
        const int taps = isx ? a.taps_x : a.taps_y;
        float* w = isx ? tab + tid * a.taps_x : tab + BX * a.taps_x + (tid - BX) * a.taps_y;
        int lo = 0, cnt = 0;
        float centre = 0.f, inv = 1.f;
        if (i < a.S) window(i, isx ? a.scale_x : a.scale_y, isx ? a.W : a.H, lo, cnt, centre, inv);
        float total = 0.f;
        for (int j = 0; j < taps; ++j) {
            const float v = j < cnt ? keys_cubic(((float)(j + lo) - centre + 0.5f) * inv) : 0.f;
            w[j] = v;
            total += v;
        }
        const float r = total != 0.f ? 1.f / total : 0.f;
        for (int j = 0; j < taps; ++j) w[j] *= r;
        meta[2 * tid] = lo;
        meta[2 * tid + 1] = cnt;
    }
    block_sync();
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= a.S || oy >= a.S) return;
    const float* wx = tab + tx * a.taps_x;
    const float* wy = tab + BX * a.taps_x + ty * a.taps_y;
    const int xlo = meta[2 * tx], xcnt = meta[2 * tx + 1], ylo = meta[2 * (BX + ty)], ycnt = meta[2 * (BX + ty) + 1];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int jy = 0; jy < ycnt; ++jy) {
        const uint8_t* row = a.src + ((long)(ylo + jy) * a.W + xlo) * 3;
        float r3[3] = {0.f, 0.f, 0.f};
        for (int jx = 0; jx < xcnt; ++jx) {
            const float w = wx[jx];
            r3[0] = fmaf(w, (float)row[3 * jx], r3[0]);
            r3[1] = fmaf(w, (float)row[3 * jx + 1], r3[1]);
            r3[2] = fmaf(w, (float)row[3 * jx + 2], r3[2]);
        }
        const float w = wy[jy];
        acc[0] = fmaf(w, r3[0], acc[0]); acc[1] = fmaf(w, r3[1], acc[1]); acc[2] = fmaf(w, r3[2], acc[2]);
    }
    const long plane = (long)a.S * a.S, o = (long)oy * a.S + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(acc[c], 0.f), 255.f) * a.mul[c] + a.add[c];
        if (a.out_f32) reinterpret_cast<float*>(a.dst)[c * plane + o] = v;
        else reinterpret_cast<uint16_t*>(a.dst)[c * plane + o] = (uint16_t)f32_to_bf16_bits(v);
    }
}

// taps needed on one axis: ceil(2 max(scale, 1)) * 2 + 1 bounds torch's window size
inline int max_taps(float scale) {
    const float support = scale >= 1.f ? 2.f * scale : 2.f;
    return (int)(support) * 2 + 3;
}

}  // namespace vimg
