// ViT attention kernels under the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <gfx950_prims.h>
#include <attention_kernels.h>

template <int D, int QT, int S>
static void run(vattn::Args p, int B) {
    using G = vattn::Geo<D, QT, S>;
    p.nqb = (p.L + G::NQ - 1) / G::NQ;
    p.BH = B * p.H;
    emu::launch(dim3((unsigned)(p.nqb * p.BH)), dim3(256), [&] { vattn::fwd_kernel<D, QT, S>(p); });
}

// S = 0: plain attention over L keys; S = 14 / 64: SAM window of S x S tokens with the decomposed rel-pos bias.
extern "C" int emu_attention_fwd(int B, int L, int H, int D, const void* q, const void* k, const void* v,
                                 long sb, long sl, long sh, void* o, int qt, int S, const void* rel_h, const void* rel_w) {
    vattn::Args p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, sb, sl, sh, L, H,
                  (float)(1.4426950408889634 / sqrt((double)D)), (const uint16_t*)rel_h, (const uint16_t*)rel_w, 0, 0};
    if (S == 0 && D == 64 && qt == 1) run<64, 1, 0>(p, B);
    else if (S == 0 && D == 64 && qt == 2) run<64, 2, 0>(p, B);
    else if (S == 0 && D == 72 && qt == 1) run<72, 1, 0>(p, B);
    else if (S == 0 && D == 72 && qt == 2) run<72, 2, 0>(p, B);
    else if (S == 14 && D == 64 && qt == 1) run<64, 1, 14>(p, B);
    else if (S == 14 && D == 64 && qt == 2) run<64, 2, 14>(p, B);
    else if (S == 64 && D == 64 && qt == 2) run<64, 2, 64>(p, B);
    else return -1;
    return 0;
}
