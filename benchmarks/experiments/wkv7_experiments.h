// Backward variants that are NOT part of the product library: A/B partners and timing builds, compiled into wkv7_capi.hip /
// wkv7_profile.hip only with -DVRWKV_V6_EXPERIMENTS -I benchmarks/experiments (benchmarks/build_alt.sh does that) and selected through
// vrwkv_wkv7_set_backward_variant:
//    7        wkv7_bwd_v7.h (the v6 schedule with the full-row memory role)
//   10 / 11   wkv7_bwd_v8.h variants 9 / 8 with the element-wise tail and the gradient stores on the J waves (JTAIL; round 5: 1 - 9 % slower,
//             profiles/r5_wkv7_jtail_*.json*)
//   61 .. 67, 71 .. 77, 81 .. 88   v6 / v7 / v8 with one or two wave roles switched off (SKIP mask: 1 = no P, 2 = no I, 4 = no J,
//             8 = no T chain; RESULTS ARE GARBAGE, timing only)
#pragma once
#include <wkv7_launch.h>
#include <wkv7_bwd_v6.h>
#include <wkv7_bwd_v6_kernel.h>     // the round-3 kernel (variant 6)
#include <wkv7_bwd_v7.h>
#include <wkv7_bwd_v8x.h>     // the v8 kernel WITH its knobs (namespace wkv7v8x); the product's wkv7_bwd_v8.h has none

namespace wkv7exp {
using namespace wkv7launch;

inline bool is_experiment(int var) { return var == 6 || var == 7 || var == 10 || var == 11 || (var >= 20 && var <= 59) || (var > 60 && var < 68) || (var > 70 && var < 78) || (var > 80 && var < 89); }

inline int launch(int var, dim3 grid, hipStream_t st, const wkv7::BwdArgs& p) {
    void (*kern)(wkv7::BwdArgs) = nullptr;
    size_t lds = sizeof(wkv7v8x::LdsV8);
    switch (var) {
        case 6: kern = &wkv7v6::bwd_kernel_v6<false>; lds = sizeof(wkv7v6::LdsV6); break;
        case 7: kern = &wkv7v7::bwd_kernel_v7<false>; lds = sizeof(wkv7v7::LdsV7); break;
        case 10: kern = &wkv7v8x::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, true, true>; break;
        case 11: kern = &wkv7v8x::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, false, true>; break;
        // round 6: variant 9 with OPT bits (wkv7_bwd_v8.h): 21 dealt tile-pair reads, 22 swizzled dS image, 23 both; timing only (garbage results): 24 no tail
        // stores, 25 no S0 requests, 26 neither; 27 S0 by register prefetch in the J waves, 28 = 27 + 23, 29 = 27 without tail stores (timing only)
#define VRWKV_OPT_CASE(v, o) case v: kern = &wkv7v8x::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, true, false, o>; break;
        VRWKV_OPT_CASE(20, 0) VRWKV_OPT_CASE(21, 1) VRWKV_OPT_CASE(22, 2) VRWKV_OPT_CASE(23, 3) VRWKV_OPT_CASE(24, 64) VRWKV_OPT_CASE(25, 128) VRWKV_OPT_CASE(26, 192) VRWKV_OPT_CASE(27, 4) VRWKV_OPT_CASE(28, 20) VRWKV_OPT_CASE(29, 4 + 64)
        // cache policy of the requests (rows: bits 8-9, S0: bits 10-11; 1 nt, 2 sc1, 3 sc0 sc1 nt) and non-temporal tail stores (4096)
        VRWKV_OPT_CASE(30, 256) VRWKV_OPT_CASE(31, 1024) VRWKV_OPT_CASE(32, 1280) VRWKV_OPT_CASE(33, 2048) VRWKV_OPT_CASE(34, 3072) VRWKV_OPT_CASE(35, 4096) VRWKV_OPT_CASE(36, 4096 + 1280) VRWKV_OPT_CASE(37, 512 + 2048) VRWKV_OPT_CASE(38, 8192) VRWKV_OPT_CASE(39, 16384) VRWKV_OPT_CASE(47, 32768) VRWKV_OPT_CASE(48, 16384 + 32768) VRWKV_OPT_CASE(49, 16384 + 32768 + 3 + 256) VRWKV_OPT_CASE(50, 65536) VRWKV_OPT_CASE(51, 65536 + 3) VRWKV_OPT_CASE(52, 131072)     /* 38: full-row tail stores through the `res` slots */
#undef VRWKV_OPT_CASE
        // static wave priorities of the three roles (I, J, P; the product: 0, 0, 1) on variant 9
#define VRWKV_PRIO_CASE(v, pi, pj, pp) case v: kern = &wkv7v8x::bwd_kernel_v8<false, pi, pj, pp, 0, true, pp, true, false, 0>; break;
        VRWKV_PRIO_CASE(40, 0, 1, 1) VRWKV_PRIO_CASE(41, 1, 0, 1) VRWKV_PRIO_CASE(42, 0, 0, 0) VRWKV_PRIO_CASE(43, 0, 0, 2) VRWKV_PRIO_CASE(44, 1, 1, 2) VRWKV_PRIO_CASE(45, 0, 2, 1) VRWKV_PRIO_CASE(46, 1, 2, 0)
#undef VRWKV_PRIO_CASE
#define VRWKV_ROLE_CASES(base, KERN, LDS, ...)                                                        \
        case base + 1: kern = &KERN<false, __VA_ARGS__ 1>; lds = sizeof(LDS); break;   /* no P */        \
        case base + 2: kern = &KERN<false, __VA_ARGS__ 2>; lds = sizeof(LDS); break;   /* no I */        \
        case base + 3: kern = &KERN<false, __VA_ARGS__ 4>; lds = sizeof(LDS); break;   /* no J */        \
        case base + 4: kern = &KERN<false, __VA_ARGS__ 3>; lds = sizeof(LDS); break;   /* J alone */     \
        case base + 5: kern = &KERN<false, __VA_ARGS__ 5>; lds = sizeof(LDS); break;   /* I alone */     \
        case base + 6: kern = &KERN<false, __VA_ARGS__ 6>; lds = sizeof(LDS); break;   /* P alone */     \
        case base + 7: kern = &KERN<false, __VA_ARGS__ 7>; lds = sizeof(LDS); break;   /* barriers only */
        VRWKV_ROLE_CASES(60, wkv7v6::bwd_kernel_v6, wkv7v6::LdsV6, 0, 0, 1, false, true,)
        VRWKV_ROLE_CASES(70, wkv7v7::bwd_kernel_v7, wkv7v7::LdsV7, 0, 0, 1,)
        VRWKV_ROLE_CASES(80, wkv7v8x::bwd_kernel_v8, wkv7v8x::LdsV8, 0, 0, 1,)
#undef VRWKV_ROLE_CASES
        case 88: kern = &wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 8>; break;     // everything but the T chain (P wave 0 only raises its flag)
        default: return VRWKV_EINVAL;
    }
    return launch_lds(kern, grid, dim3(768), lds, st, p);
}

// profiling entry: 3 = v7 with the v6 stamps; 20 + SKIP mask = the v6 profiling build with roles switched off
inline int launch_profile(int backward, dim3 grid, hipStream_t st, const wkv7::BwdArgs& p) {
    if (backward == 3) return launch_lds(&wkv7v7::bwd_kernel_v7<true>, grid, dim3(768), sizeof(wkv7v7::LdsV7), st, p);
    void (*kern)(wkv7::BwdArgs) = nullptr;
    switch (backward - 20) {
        case 0: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 0>; break;
        case 1: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 1>; break;
        case 2: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 2>; break;
        case 3: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 3>; break;
        case 4: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 4>; break;
        case 5: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 5>; break;
        case 6: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 6>; break;
        default: kern = &wkv7v6::bwd_kernel_v6<true, 0, 0, 1, false, true, 7>; break;
    }
    return launch_lds(kern, grid, dim3(768), sizeof(wkv7v6::LdsV6), st, p);
}

}  // namespace wkv7exp
