// Profiling builds of the chunked WKV7 kernels (C-ABI: vrwkv_wkv7_profile_bf16): dbg[0..31] (device, zeroed by the caller) receives the
// shader-clock cycles workgroup 0 spent in each phase (WKV_STAMP in wkv7_chunked.h); benchmarks/wkv7_phases.py names the slots.
#include <wkv7_launch.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>
#include <wkv7_fwd_v4.h>
#include <wkv7_bwd_v5.h>
#include <wkv7_bwd_v6.h>
#include <wkv7_bwd_v8.h>
#ifdef VRWKV_V6_EXPERIMENTS
#include <wkv7_experiments.h>
#endif
#ifndef VRWKV_PROF_AHEAD
#define VRWKV_PROF_AHEAD false      // the v8 entry stamps variant 8; -DVRWKV_PROF_AHEAD=true: variant 9's schedule
#endif

using namespace wkv7launch;

// backward: 0 = forward (the default kernel for the size), 1 = wkv7_bwd_v5.h, 4 = wkv7_bwd_v8.h; experiment builds: 2 = the round-3 kernel, 3 = v7,
// 20 + mask = v6 with roles switched off
extern "C" int vrwkv_wkv7_profile_bf16(int backward, int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                                       const void* z, const void* a, const void* dy, void* y, float* s, float* sa,
                                       void* dw, void* dq, void* dk, void* dv, void* dz, void* da,
                                       unsigned long long* dbg, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!dbg) return VRWKV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long)B * H));
    if (!backward) {
        const wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                              (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa, dbg};
        if (g_fwd_variant == 7 || g_fwd_variant == -1) return launch_lds(&wkv7f4::fwd_kernel_v4<true>, grid, dim3(512), sizeof(wkv7f4::LdsF4), st, p);
        return launch_lds(&wkv7c::fwd_kernel_v3<true, false, 1, 1, false, true, true>, grid, dim3(512), sizeof(wkv7c::LdsF), st, p);
    }
    const wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                          (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                          (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
#ifdef VRWKV_V6_EXPERIMENTS
    if (backward == 3 || (backward >= 20 && backward < 28)) return wkv7exp::launch_profile(backward, grid, st, p);
#endif
    if (backward == 4)       // wkv7_bwd_v8.h: same stamps as v6
        return launch_lds(&wkv7v8::bwd_kernel_v8<true, VRWKV_PROF_AHEAD>, grid, dim3(768), sizeof(wkv7v8::LdsV8), st, p);
#ifdef VRWKV_V6_EXPERIMENTS
    if (backward == 2)       // three-stage pipeline (benchmarks/experiments/wkv7_bwd_v6_kernel.h): I / J / P wave 0, five stamps each
        return launch_lds(&wkv7v6::bwd_kernel_v6<true>, grid, dim3(768), sizeof(wkv7v6::LdsV6), st, p);
#endif
    if (backward == 1) return launch_lds(&wkv7v5::bwd_kernel_v5<true, BWD_V5_MODE>, grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
    return VRWKV_EINVAL;
}
