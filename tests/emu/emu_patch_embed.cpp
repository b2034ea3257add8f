// Patch-embedding kernel under the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <gfx950_prims.h>
#include <patch_embed_kernels.h>

extern "C" int emu_patch_embed(int B, int Himg, int Wimg, int P, int N, const void* px, const void* w, const void* bias,
                               const void* pos, void* out, int Ltot, int prefix) {
    const int gw = Wimg / P, M = (Himg / P) * gw;
    vpe::Args a{(const uint16_t*)px, (const uint16_t*)w, (const uint16_t*)bias, (const uint16_t*)pos, (uint16_t*)out,
                Himg, Wimg, N, gw, M, Ltot, prefix};
    if (P == 14) emu::launch(dim3((unsigned)(B * (M / 64))), dim3(256), [&] { vpe::kernel<14>(a); });
    else if (P == 16) emu::launch(dim3((unsigned)(B * (M / 64))), dim3(256), [&] { vpe::kernel<16>(a); });
    else return -1;
    return 0;
}
