// Host launchers + C-ABI for the WKV7 kernels (include/visualrwkv_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/visualrwkv_hip.h"
#include <wkv7_kernels.h>
#include <wkv7_chunked.h>
#include <wkv7_chunked_bwd.h>
#include <wkv7_fwd_v3.h>
#include <wkv7_bwd_v3.h>
#include <wkv7_bwd_v4.h>
#include <wkv7_bwd_v5.h>
#include <wkv7_fwd_v5.h>

namespace {

int g_fwd_variant = -1;
int g_bwd_variant = -1;
constexpr int BWD_V3_DEFAULT_MODE = 2;    // same-process A/B on MI355X (benchmarks/wkv7_ab.py): counters +1..3 % slower, bf16x3 doubling -1 %

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

int check_common(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return VRWKV_EINVAL;
    if (T % VRWKV_CHUNK_LEN != 0) return VRWKV_ESHAPE;   // cuda_backward asserts this, wkv7_cuda.cu:136
    return VRWKV_OK;
}

int finish_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VRWKV_OK : (int)e;
}

}  // namespace

extern "C" {

int vrwkv_abi_version(void) { return 1; }

const char* vrwkv_strerror(int code) {
    switch (code) {
        case VRWKV_OK: return "ok";
        case VRWKV_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case VRWKV_ESHAPE: return "T must be a multiple of 16";
        case VRWKV_EALIGN: return "pointer not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int vrwkv_wkv7_set_forward_variant(int variant) {
    if (variant > 13) return VRWKV_EINVAL;
    g_fwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_set_backward_variant(int variant) {
    if (variant > 12) return VRWKV_EINVAL;
    g_bwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_forward_bf16(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, void* y, float* s, float* sa, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !y || !s || !sa) return VRWKV_EINVAL;
    if (misaligned(w) || misaligned(q) || misaligned(k) || misaligned(v) || misaligned(z) || misaligned(a) ||
        misaligned(y) || misaligned(s) || misaligned(sa))
        return VRWKV_EALIGN;
    wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa};
    hipStream_t st = (hipStream_t)stream;
    const long heads = (long)B * H;
    int variant = g_fwd_variant;
    if (variant < 0) variant = 5;                                           // chunked MFMA, producer/consumer waves
    const dim3 grid((unsigned)heads);
    if (variant == 12 || variant == 13) {           // second-generation schedule (wkv7_fwd_v5.h); 13: T chain on the bf16 matrix core
        auto launch5 = [&](auto kern) -> int {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(wkv7v5::LdsF5));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7v5::LdsF5), st, p);
            return 0;
        };
        const int e = variant == 12 ? launch5(&wkv7v5::fwd_kernel_v5<false, 0>) : launch5(&wkv7v5::fwd_kernel_v5<false, 2>);
        if (e) return e;
    } else if (variant == 0) hipLaunchKernelGGL((wkv7::fwd_kernel<16, 8>), grid, dim3(64), 0, st, p);
    else if (variant == 1) hipLaunchKernelGGL((wkv7::fwd_kernel<8, 16>), grid, dim3(128), 0, st, p);
    else if (variant == 2) hipLaunchKernelGGL((wkv7::fwd_kernel<4, 16>), grid, dim3(256), 0, st, p);
    else if (variant == 3) hipLaunchKernelGGL(wkv7c::fwd_kernel_t<false>, grid, dim3(256), 0, st, p);
    else {
        // variants 4..7: producer/consumer kernel; bit 0 of (variant-4): narrow stores, bit 1: no producer priority
        auto launch = [&](auto kern) -> int {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsF));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7c::LdsF), st, p);
            return 0;
        };
        int e = 0;
        if (variant == 4) e = launch(&wkv7c::fwd_kernel_v3<false, true, 1>);
        else if (variant == 5) e = launch(&wkv7c::fwd_kernel_v3<false, false, 1>);
        else if (variant == 6) e = launch(&wkv7c::fwd_kernel_v3<false, true, 0>);
        else if (variant == 7) e = launch(&wkv7c::fwd_kernel_v3<false, false, 0>);
        else if (variant == 8) e = launch(&wkv7c::fwd_kernel_v3<false, false, 1, 2, false>);     // 5 + two chunks of prefetch
        else if (variant == 9) e = launch(&wkv7c::fwd_kernel_v3<false, false, 1, 1, true>);      // 5 + DPP suffix scan
        else if (variant == 10) e = launch(&wkv7c::fwd_kernel_v3<false, false, 1, 2, true>);     // 5 + both
        else e = launch(&wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true>);                 // 5 + transpose reads (no transposed LDS copies)
        if (e) return e;
    }
    return finish_launch();
}

int vrwkv_wkv7_forward_state_bf16(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                                  const void* z, const void* a, void* y, const float* s0, float* s_final, float* s_ckpt,
                                  float* sa, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !y) return VRWKV_EINVAL;
    if (misaligned(w) || misaligned(q) || misaligned(k) || misaligned(v) || misaligned(z) || misaligned(a) || misaligned(y) ||
        (s0 && misaligned(s0)) || (s_final && misaligned(s_final)) || (s_ckpt && misaligned(s_ckpt)) || (sa && misaligned(sa)))
        return VRWKV_EALIGN;
    wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s_ckpt, sa, nullptr, s0, s_final};
    auto kern = &wkv7c::fwd_kernel_v3<false, false, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(wkv7c::LdsF));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)B * H)), dim3(512), sizeof(wkv7c::LdsF), (hipStream_t)stream, p);
    return finish_launch();
}

int vrwkv_wkv7_backward_bf16(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                             const void* z, const void* a, const void* dy, const float* s, const float* sa,
                             void* dw, void* dq, void* dk, void* dv, void* dz, void* da, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !dy || !s || !sa || !dw || !dq || !dk || !dv || !dz || !da) return VRWKV_EINVAL;
    if (misaligned(w) || misaligned(q) || misaligned(k) || misaligned(v) || misaligned(z) || misaligned(a) ||
        misaligned(dy) || misaligned(s) || misaligned(sa) || misaligned(dw) || misaligned(dq) || misaligned(dk) ||
        misaligned(dv) || misaligned(dz) || misaligned(da))
        return VRWKV_EALIGN;
    wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                    (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long)B * H));
    if (g_bwd_variant < 0 || (g_bwd_variant >= 7 && g_bwd_variant <= 12)) {      // default: second-generation schedule (wkv7_bwd_v5.h); 8: T doubling on the bf16 matrix core
        auto launch5 = [&](auto kern) -> int {
            // the > 64 KB LDS opt-in is per device: set it on every launch (cheap) instead of caching a per-process flag
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(wkv7v5::LdsV5));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
            return 0;
        };
        const int e = g_bwd_variant == 8 ? launch5(&wkv7v5::bwd_kernel_v5<false, 2>) : g_bwd_variant == 9 ? launch5(&wkv7v5::bwd_kernel_v5<false, 4>)
                    : g_bwd_variant == 10 ? launch5(&wkv7v5::bwd_kernel_v5<false, 8>) : g_bwd_variant == 11 ? launch5(&wkv7v5::bwd_kernel_v5<false, 6>)
                    : g_bwd_variant == 12 ? launch5(&wkv7v5::bwd_kernel_v5<false, 10>) : g_bwd_variant == 7 ? launch5(&wkv7v5::bwd_kernel_v5<false, 0>)
                    : launch5(&wkv7v5::bwd_kernel_v5<false, 6>);
        if (e) return e;
    } else if (g_bwd_variant == 0) {
        hipLaunchKernelGGL((wkv7::bwd_kernel<8>), grid, dim3(256), 0, st, p);
    } else if (g_bwd_variant >= 2 && g_bwd_variant <= 5) {
        // 2: barriers + f32 doubling, 3: hand-off counters, 4: barriers + bf16x3 doubling, 5: both
        const int mode = g_bwd_variant - 2;
        auto launch = [&](auto kern) -> int {
            // > 64 KB of LDS needs the opt-in per device: set on every launch (cheap), no per-process flag
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(wkv7c::LdsB3));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7c::LdsB3), st, p);
            return 0;
        };
        int e = mode == 0 ? launch(&wkv7c::bwd_kernel_v3<false, 0>) : mode == 1 ? launch(&wkv7c::bwd_kernel_v3<false, 1>)
              : mode == 2 ? launch(&wkv7c::bwd_kernel_v3<false, 2>) : launch(&wkv7c::bwd_kernel_v3<false, 3>);
        if (e) return e;
    } else if (g_bwd_variant == 6) {      // 12 waves: I / J consumer roles + producers (wkv7_bwd_v4.h)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::bwd_kernel_v4<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsB3));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(wkv7c::bwd_kernel_v4<false>, grid, dim3(768), sizeof(wkv7c::LdsB3), st, p);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::bwd_kernel_t<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsB));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(wkv7c::bwd_kernel_t<false>, grid, dim3(256), sizeof(wkv7c::LdsB), st, p);
    }
    return finish_launch();
}

int vrwkv_wkv7_backward_segments_bf16(int B, int T, int H, int nseg, const void* w, const void* q, const void* k, const void* v,
                                      const void* z, const void* a, const void* dy, const float* s, const float* sa,
                                      const float* ds_in, float* ds_out,
                                      void* dw, void* dq, void* dk, void* dv, void* dz, void* da, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !dy || !s || !sa || !dw || !dq || !dk || !dv || !dz || !da) return VRWKV_EINVAL;
    if (nseg < 1 || nseg > T / VRWKV_CHUNK_LEN) return VRWKV_ESHAPE;
    if (misaligned(w) || misaligned(q) || misaligned(k) || misaligned(v) || misaligned(z) || misaligned(a) ||
        misaligned(dy) || misaligned(s) || misaligned(sa) || misaligned(dw) || misaligned(dq) || misaligned(dk) ||
        misaligned(dv) || misaligned(dz) || misaligned(da) || misaligned(ds_in) || misaligned(ds_out))
        return VRWKV_EALIGN;
    wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                    (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da};
    p.ds_in = ds_in; p.ds_out = ds_out; p.nseg = nseg;
    const dim3 grid((unsigned)((long)B * H * nseg));
    if (g_bwd_variant < 0 || g_bwd_variant == 7 || g_bwd_variant == 8) {
        auto kern = &wkv7v5::bwd_kernel_v5<false, 6, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7v5::LdsV5));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7v5::LdsV5), (hipStream_t)stream, p);
        return finish_launch();
    }
    auto kern = &wkv7c::bwd_kernel_v3<false, BWD_V3_DEFAULT_MODE, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(wkv7c::LdsB3));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7c::LdsB3), (hipStream_t)stream, p);
    return finish_launch();
}

// Profiling builds of the chunked kernels: dbg[0..15] (device, zeroed by the caller) receives the shader-clock
// cycles workgroup 0 spent in each phase (see WKV_STAMP in wkv7_chunked*.h).
int vrwkv_wkv7_profile_bf16(int backward, int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, const void* dy, void* y, float* s, float* sa,
                            void* dw, void* dq, void* dk, void* dv, void* dz, void* da,
                            unsigned long long* dbg, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!dbg) return VRWKV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long)B * H));
    if (!backward) {
        wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa, dbg};
        if (g_fwd_variant == 12) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v5::fwd_kernel_v5<true, 0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v5::LdsF5));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wkv7v5::fwd_kernel_v5<true, 0>), grid, dim3(512), sizeof(wkv7v5::LdsF5), st, p);
        } else if (backward == 0 && g_fwd_variant == 3) {
            hipLaunchKernelGGL(wkv7c::fwd_kernel_t<true>, grid, dim3(256), 0, st, p);
        } else {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::fwd_kernel_v3<true, false, 1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsF));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wkv7c::fwd_kernel_v3<true, false, 1>), grid, dim3(512), sizeof(wkv7c::LdsF), st, p);
        }
    } else {
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
        if (g_bwd_variant == 10) {      // register dump of workgroup 0 (debugging aid): dbg = float[2][24][256][4]
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v5::bwd_kernel_v5<false, 0, false, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v5::LdsV5));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wkv7v5::bwd_kernel_v5<false, 0, false, true>), grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
        } else if (g_bwd_variant < 0 || g_bwd_variant == 7 || g_bwd_variant == 8) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v5::bwd_kernel_v5<true, 6>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v5::LdsV5));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wkv7v5::bwd_kernel_v5<true, 6>), grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
        } else if (g_bwd_variant == 1) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::bwd_kernel_t<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsB));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(wkv7c::bwd_kernel_t<true>, grid, dim3(256), sizeof(wkv7c::LdsB), st, p);
        } else if (g_bwd_variant == 6) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::bwd_kernel_v4<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsB3));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(wkv7c::bwd_kernel_v4<true>, grid, dim3(768), sizeof(wkv7c::LdsB3), st, p);
        } else {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7c::bwd_kernel_v3<true, BWD_V3_DEFAULT_MODE>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsB3));
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wkv7c::bwd_kernel_v3<true, BWD_V3_DEFAULT_MODE>), grid, dim3(512), sizeof(wkv7c::LdsB3), st, p);
        }
    }
    return finish_launch();
}

}  // extern "C"
