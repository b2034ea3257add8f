// ViT attention kernel under the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <gfx950_prims.h>
#include <attention_kernels.h>

extern "C" int emu_attention_fwd(int B, int L, int H, int D, const void* q, const void* k, const void* v,
                                 long sb, long sl, long sh, void* o) {
    vattn::Args p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, sb, sl, sh, L, H,
                  (float)(1.4426950408889634 / sqrt((double)D))};
    const dim3 grid((unsigned)((L + 63) / 64), (unsigned)(B * H));
    if (D == 64) emu::launch(grid, dim3(256), [&] { vattn::fwd_kernel<64>(p); });
    else emu::launch(grid, dim3(256), [&] { vattn::fwd_kernel<72>(p); });
    return 0;
}
