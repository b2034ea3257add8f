// Tower image transform under the host emulator.  TEST INFRASTRUCTURE ONLY.
#include <gfx950_prims.h>
#include <image_kernels.h>

extern "C" int emu_resize_normalize_u8(int H, int W, const void* src, int S, const float* mean3, const float* std3, void* dst, int f32) {
    vimg::Args a{};
    a.src = (const uint8_t*)src; a.dst = dst; a.H = H; a.W = W; a.S = S; a.out_f32 = f32;
    a.scale_x = (float)W / (float)S; a.scale_y = (float)H / (float)S;
    a.taps_x = vimg::max_taps(a.scale_x); a.taps_y = vimg::max_taps(a.scale_y);
    for (int c = 0; c < 3; ++c) { a.mul[c] = 1.f / (255.f * std3[c]); a.add[c] = -mean3[c] / std3[c]; }
    emu::launch(dim3((unsigned)((S + vimg::BX - 1) / vimg::BX), (unsigned)((S + vimg::BY - 1) / vimg::BY)), dim3(256),
                [&] { vimg::resize_normalize_kernel(a); });
    return 0;
}
