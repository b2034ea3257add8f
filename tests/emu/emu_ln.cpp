// LayerNorm kernels (csrc/ln_kernels.h) under the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <gfx950_prims.h>
#include <ln_kernels.h>
#include <vector>

using namespace vln;

static int threads_for(int C) { return (C / 8 + 63) / 64 * 64; }

// residual add + LayerNorm + token shift + M lerps; `grid` workgroups share the ntok rows (contiguous ranges)
extern "C" int emu_ln_mix_fwd(long ntok, int T, int C, float eps, int M, const void* x, const void* delta, const void* w, const void* b,
                              const void* const* mu, void* xn, void* const* out, float* mean, float* rstd, int grid) {
    LmPtrs pm{}; LmOuts po{};
    for (int j = 0; j < M; ++j) { pm.p[j] = (const uint16_t*)mu[j]; po.p[j] = (uint16_t*)out[j]; }
    const dim3 g((unsigned)grid), blk((unsigned)threads_for(C));
    if (M == 1) emu::launch(g, blk, [&] { ln_mix_fwd_kernel<1>(ntok, T, C, eps, (const uint16_t*)x, (const uint16_t*)delta, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn, mean, rstd, pm, po); });
    else if (M == 6) emu::launch(g, blk, [&] { ln_mix_fwd_kernel<6>(ntok, T, C, eps, (const uint16_t*)x, (const uint16_t*)delta, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn, mean, rstd, pm, po); });
    else return -1;
    return 0;
}

// its backward for M = 1 (channel-mix): dx, dwb = (dgamma | dbeta) (2C floats), dmu (C floats)
extern "C" int emu_ln_mix_bwd1(long ntok, int T, int C, const void* xn, const float* mean, const float* rstd, const void* w, const void* b,
                               const void* mu, const void* dout, const void* dres, void* dx, float* dwb, float* dmu, int grid) {
    LmPtrs pm{}, pd{};
    pm.p[0] = (const uint16_t*)mu; pd.p[0] = (const uint16_t*)dout;
    std::vector<float> part_ln((size_t)grid * 2 * C), part_mu((size_t)grid * C);
    float* pl = part_ln.data(); float* pmu = part_mu.data();
    emu::launch(dim3((unsigned)grid), dim3((unsigned)threads_for(C)), [&] {
        ln_mix_bwd_kernel<1, false, 256>(ntok, T, C, (const uint16_t*)xn, mean, rstd, (const uint16_t*)w, (const uint16_t*)b, pm, pd,
                                         (const uint16_t*)nullptr, (const uint16_t*)dres, (uint16_t*)dx, pl, pmu); });
    emu::launch(dim3((unsigned)(2L * C / 16)), dim3(256), [&] { ln_colsum_kernel(grid, 2L * C, pl, dwb); });
    emu::launch(dim3((unsigned)(C / 16)), dim3(256), [&] { ln_colsum_kernel(grid, (long)C, pmu, dmu); });
    return 0;
}

// plain add + LayerNorm forward / backward (the two-kernel path's LayerNorm half)
extern "C" int emu_add_ln_fwd(long ntok, int C, float eps, const void* x, const void* delta, const void* w, const void* b, void* xn, void* y,
                              float* mean, float* rstd, int grid) {
    emu::launch(dim3((unsigned)grid), dim3((unsigned)threads_for(C)), [&] {
        add_ln_fwd_kernel(ntok, C, eps, (const uint16_t*)x, (const uint16_t*)delta, (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)xn, (uint16_t*)y,
                          mean, rstd, (const long*)nullptr, (const uint16_t*)nullptr); });
    return 0;
}
extern "C" int emu_add_ln_bwd(long ntok, int C, const void* dy, const void* dres, const void* xn, const float* mean, const float* rstd,
                              const void* w, void* dx, float* dwb, int grid) {
    std::vector<float> part((size_t)grid * 2 * C);
    float* pl = part.data();
    emu::launch(dim3((unsigned)grid), dim3((unsigned)threads_for(C)), [&] {
        add_ln_bwd_kernel(ntok, C, (const uint16_t*)dy, (const uint16_t*)dres, (const uint16_t*)xn, mean, rstd, (const uint16_t*)w, (uint16_t*)dx, pl,
                          (const long*)nullptr); });
    emu::launch(dim3((unsigned)(2L * C / 16)), dim3(256), [&] { ln_colsum_kernel(grid, 2L * C, pl, dwb); });
    return 0;
}
