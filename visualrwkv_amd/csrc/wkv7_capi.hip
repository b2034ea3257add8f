// Host launchers + C-ABI for the WKV7 kernels (include/visualrwkv_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/visualrwkv_hip.h"
#include <wkv7_kernels.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>
#include <wkv7_fwd_v4.h>
#include <wkv7_bwd_v5.h>
#include <wkv7_bwd_v6.h>
#include <wkv7_bwd_v7.h>
#include <wkv7_bwd_v8.h>
#ifndef VRWKV_PROF_AHEAD
#define VRWKV_PROF_AHEAD false      // the profile entry point stamps the default schedule; -DVRWKV_PROF_AHEAD=true: variant 9's
#endif

namespace {

int g_fwd_variant = -1;
int g_bwd_variant = -1;
int g_last_fwd = 0, g_last_bwd = 0;     // what the last launch of each direction resolved to (vrwkv_wkv7_last_variant)
// T chain on the bf16 matrix core (2) + producer priority 2 (4; same-box A/B: 1.18 -> 1.09 ms) + priorities swapped in
// segment 1, where the producers have ~1.2k cycles of slack per chunk and the consumers none (128; 1.09 -> 1.04 ms)
constexpr int BWD_V5_MODE = 2 + 4 + 128;
// Forward default: no Ab / Kb images (state update from Ah / Kh, scaled by c_L afterwards; T chain splitting every matrix once
// per level) + natural [t][j] images read with ds_read_b64_tr_b16.  Same-box A/B (benchmarks/wkv7_ab.py --fwd 1 2 4): B=8
// 0.358 -> 0.331 -> 0.324 ms, B=16 0.647 -> 0.637 -> 0.627 ms.  Variant 1 = the round-2 instantiation.
#define VRWKV_FWD_DEFAULT wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>
#define VRWKV_FWD_DEFAULT_PROF wkv7c::fwd_kernel_v3<true, false, 1, 1, false, true, true>
// few heads (B*H <= 128: at most half of the 256 CUs would be busy): two workgroups per head, 32 value rows each
#define VRWKV_FWD_ISPLIT wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true, true>
constexpr long FWD_ISPLIT_MAX_HEADS = 128;
constexpr bool FWD_DEFAULT_V4 = true;      // wkv7_fwd_v4.h (full-row memory traffic) for B*H > 128
// same-box A/B on MI355X, B=16 x 2624 x 32 heads: micro-benchmark (random inputs) 1.042 -> 0.993 ms, inside the training step
// (bench.py, VRWKV_BWD_VARIANT=5 / 6) 0.981 -> 0.872 ms
constexpr int BWD_DEFAULT = 9;          // 5: wkv7_bwd_v5.h   6: wkv7_bwd_v6.h   7: wkv7_bwd_v7.h   8: wkv7_bwd_v8.h (one dS copy, T on P wave 0, full-row LDS-DMA)   9: 8 with the score pieces a step ahead on the P waves

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
    if (variant != -1 && !(variant >= 1 && variant <= 5) && variant != 7) return VRWKV_EINVAL;   // 1..5: A/B instantiations of wkv7_fwd_v3.h; 7: wkv7_fwd_v4.h
    g_fwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_set_backward_variant(int variant) {
    if (variant != -1 && !(variant >= 5 && variant <= 11) && !(variant >= 60 && variant < 90)) return VRWKV_EINVAL;   // see include/visualrwkv_hip.h
    g_bwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_last_variant(int backward) { return backward ? g_last_bwd : g_last_fwd; }

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
    const dim3 grid((unsigned)heads);
    {                                               // chunked MFMA, producer / consumer waves (wkv7_fwd_v3.h)
        void (*kern)(wkv7::FwdArgs) = &VRWKV_FWD_DEFAULT;
        if (g_fwd_variant == 1) kern = &wkv7c::fwd_kernel_v3<false, false, 1>;
        if (g_fwd_variant == 2) kern = &wkv7c::fwd_kernel_v3<false, false, 1, 1, false, false, true>;
        if (g_fwd_variant == 3) kern = &wkv7c::fwd_kernel_v3<false, true, 1, 1, false, false, true>;     // + 16-byte transposed stores
        if (g_fwd_variant == 4) kern = &wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>;     // + natural images read with tr16
        if (g_fwd_variant == 5) kern = &wkv7c::fwd_kernel_v3<false, false, 1, 2, false, false, true>;     // + two chunks of prefetch
        dim3 g2 = grid;
        if (g_fwd_variant == -1 && heads <= FWD_ISPLIT_MAX_HEADS) { kern = &VRWKV_FWD_ISPLIT; g2 = dim3((unsigned)(2 * heads)); }
        if (g_fwd_variant == 7 || (g_fwd_variant == -1 && FWD_DEFAULT_V4 && heads > FWD_ISPLIT_MAX_HEADS)) {      // full-row memory traffic (wkv7_fwd_v4.h)
            void (*k4)(wkv7::FwdArgs) = &wkv7f4::fwd_kernel_v4<false>;
            hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7f4::LdsF4));
            if (e4 != hipSuccess) return (int)e4;
            hipLaunchKernelGGL(k4, grid, dim3(512), sizeof(wkv7f4::LdsF4), st, p);
            g_last_fwd = 7;
            return finish_launch();
        }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7c::LdsF));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, g2, dim3(512), sizeof(wkv7c::LdsF), st, p);
        g_last_fwd = g2.x != grid.x ? 6 : g_fwd_variant == -1 ? 4 : g_fwd_variant;      // 6: two workgroups per head; 4: the default instantiation of wkv7_fwd_v3.h
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
    const long heads = (long)B * H;
    const bool split = g_fwd_variant == -1 && heads <= FWD_ISPLIT_MAX_HEADS;
    void (*kern)(wkv7::FwdArgs) = split ? &VRWKV_FWD_ISPLIT : &VRWKV_FWD_DEFAULT;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(wkv7c::LdsF));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)(split ? 2 * heads : heads)), dim3(512), sizeof(wkv7c::LdsF), (hipStream_t)stream, p);
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
    // default: variant 9 when the launch has more workgroups than the chip has CUs (measured -0.4 ... -2.3 % at B x H = 384 ... 1024), variant 8
    // for a single round of workgroups (B x H <= 256: 9 measured +0.3 ... +1.8 % there); profiles/r4_wkv7_ab.jsonl, r4c_wkv7_ab.jsonl
    int var = g_bwd_variant == -1 ? ((long)B * H > 256 ? BWD_DEFAULT : 8) : g_bwd_variant;
    const bool fits32 = (unsigned long long)B * T * H * 64ull * 4ull < (1ull << 32);      // wkv7_bwd_v8.h uses 32-bit byte offsets inside a tensor
    if (!fits32 && (var >= 8 && var <= 11 || var >= 80)) var = 6;
    if ((var >= 8 && var <= 11) || (var >= 80 && var < 90)) {
        // one copy of dL/dS, T chain on P wave 0, full-row memory role, 12 waves (wkv7_bwd_v8.h)
        void (*kern)(wkv7::BwdArgs) = &wkv7v8::bwd_kernel_v8<false>;
        if (var == 9) kern = &wkv7v8::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, true>;      // score pieces a step ahead on the P waves
        if (var == 10) kern = &wkv7v8::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, true, true>;      // 9 with the tail on the J waves
        if (var == 11) kern = &wkv7v8::bwd_kernel_v8<false, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, false, true>;      // 8 with the tail on the J waves
#ifdef VRWKV_V6_EXPERIMENTS   // role-timing builds: one or two roles switched off, results garbage
        switch (g_bwd_variant) {
            case 81: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 1>; break;     // no P
            case 82: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 2>; break;     // no I
            case 83: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 4>; break;     // no J
            case 84: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 3>; break;     // J alone
            case 85: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 5>; break;     // I alone
            case 86: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 6>; break;     // P alone
            case 87: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 7>; break;     // barriers only
            case 88: kern = &wkv7v8::bwd_kernel_v8<false, 0, 0, 1, 8>; break;     // everything but the T chain (P wave 0 only raises its flag)
            default: break;
        }
#endif
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7v8::LdsV8));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(768), sizeof(wkv7v8::LdsV8), st, p);
    } else if (var == 7 || (var >= 70 && var < 78)) {
        // three-stage wave pipeline with a full-row memory role, 12 waves (wkv7_bwd_v7.h)
        void (*kern)(wkv7::BwdArgs) = &wkv7v7::bwd_kernel_v7<false>;
#ifdef VRWKV_V6_EXPERIMENTS   // role-timing builds: one or two roles switched off, results garbage
        switch (g_bwd_variant) {
            case 71: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 1>; break;     // no P
            case 72: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 2>; break;     // no I
            case 73: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 4>; break;     // no J
            case 74: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 3>; break;     // J alone
            case 75: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 5>; break;     // I alone
            case 76: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 6>; break;     // P alone
            case 77: kern = &wkv7v7::bwd_kernel_v7<false, 0, 0, 1, 7>; break;     // barriers only
            default: break;
        }
#endif
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7v7::LdsV7));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(768), sizeof(wkv7v7::LdsV7), st, p);
    } else if (var == 6 || (var >= 60 && var < 68)) {
        // three-stage wave pipeline, 12 waves (wkv7_bwd_v6.h)
        void (*kern)(wkv7::BwdArgs) = &wkv7v6::bwd_kernel_v6<false>;
#ifdef VRWKV_V6_EXPERIMENTS   // role-timing builds (VRWKV_EXTRA_HIPCC_FLAGS=-DVRWKV_V6_EXPERIMENTS): one or two roles switched off, results garbage
        switch (g_bwd_variant) {
            case 61: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 1>; break;     // no P
            case 62: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 2>; break;     // no I
            case 63: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 4>; break;     // no J
            case 64: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 3>; break;     // J alone
            case 65: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 5>; break;     // I alone
            case 66: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 6>; break;     // P alone
            case 67: kern = &wkv7v6::bwd_kernel_v6<false, 0, 0, 1, false, true, 7>; break;     // barriers only
            default: break;
        }
#endif
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7v6::LdsV6));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(768), sizeof(wkv7v6::LdsV6), st, p);
    } else {                                        // default: second-generation schedule (wkv7_bwd_v5.h)
        auto kern = &wkv7v5::bwd_kernel_v5<false, BWD_V5_MODE>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(wkv7v5::LdsV5));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
        var = 5;
    }
    g_last_bwd = var;
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
    auto kern = &wkv7v5::bwd_kernel_v5<false, BWD_V5_MODE, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(wkv7v5::LdsV5));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, dim3(512), sizeof(wkv7v5::LdsV5), (hipStream_t)stream, p);
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
    if (!backward && (g_fwd_variant == 7 || (g_fwd_variant == -1 && FWD_DEFAULT_V4))) {
        wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7f4::fwd_kernel_v4<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7f4::LdsF4));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((wkv7f4::fwd_kernel_v4<true>), grid, dim3(512), sizeof(wkv7f4::LdsF4), st, p);
    } else if (!backward) {
        wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&VRWKV_FWD_DEFAULT_PROF),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7c::LdsF));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((VRWKV_FWD_DEFAULT_PROF), grid, dim3(512), sizeof(wkv7c::LdsF), st, p);
#ifdef VRWKV_V6_EXPERIMENTS
    } else if (backward >= 20 && backward < 28) {   // the profiling build with roles switched off (20 + SKIP mask: 1 = no P, 2 = no I, 4 = no J)
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v6::LdsV6));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(768), sizeof(wkv7v6::LdsV6), st, p);
#endif
    } else if (backward == 4) {                     // wkv7_bwd_v8.h: same stamps as v6
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v8::bwd_kernel_v8<true, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, VRWKV_PROF_AHEAD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v8::LdsV8));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((wkv7v8::bwd_kernel_v8<true, VRWKV_V8_PI, VRWKV_V8_PJ, VRWKV_V8_PP, 0, true, VRWKV_V8_PP, VRWKV_PROF_AHEAD>), grid, dim3(768), sizeof(wkv7v8::LdsV8), st, p);
    } else if (backward == 3) {                     // three-stage pipeline with the full-row memory role (wkv7_bwd_v7.h): same stamps as v6
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v7::bwd_kernel_v7<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v7::LdsV7));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((wkv7v7::bwd_kernel_v7<true>), grid, dim3(768), sizeof(wkv7v7::LdsV7), st, p);
    } else if (backward == 2) {                     // three-stage pipeline (wkv7_bwd_v6.h): I / J / P wave 0, five stamps each
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v6::bwd_kernel_v6<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v6::LdsV6));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((wkv7v6::bwd_kernel_v6<true>), grid, dim3(768), sizeof(wkv7v6::LdsV6), st, p);
    } else {
        wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                        (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                        (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da, dbg};
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wkv7v5::bwd_kernel_v5<true, BWD_V5_MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wkv7v5::LdsV5));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((wkv7v5::bwd_kernel_v5<true, BWD_V5_MODE>), grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
    }
    return finish_launch();
}

}  // extern "C"
