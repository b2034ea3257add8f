// Host launchers + C-ABI for the WKV7 kernels (include/visualrwkv_hip.h).  The profiling entry point is wkv7_profile.hip; A/B partners
// that are not part of the product (wkv7_bwd_v7.h, builds with one or two wave roles switched off, the tail on the J waves) come in
// through benchmarks/experiments/wkv7_experiments.h when the library is built with -DVRWKV_V6_EXPERIMENTS (benchmarks/build_alt.sh).
#include <atomic>
#include <wkv7_launch.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>
#include <wkv7_fwd_v4.h>
#include <wkv7_bwd_v5.h>
#include <wkv7_bwd_v6.h>     // shared building blocks of the backward kernels
#include <wkv7_bwd_v8.h>
#ifdef VRWKV_V6_EXPERIMENTS
#include <wkv7_experiments.h>
#endif

namespace wkv7launch {
std::atomic<int> g_fwd_variant{-1}, g_bwd_variant{-1};
}

namespace {
using namespace wkv7launch;
// what the last launch of each direction resolved to (vrwkv_wkv7_last_variant): a single-threaded test aid -- launches come from the
// Python thread AND from autograd's backward thread, so code that needs to know which kernel a launch of a given shape uses asks
// vrwkv_wkv7_resolve_variant (a pure function of the shape and the override) instead of reading this after the fact
std::atomic<int> g_last_fwd{0}, g_last_bwd{0};
// Forward default: no Ab / Kb images (state update from Ah / Kh, scaled by c_L afterwards; T chain splitting every matrix once
// per level) + natural [t][j] images read with ds_read_b64_tr_b16.  Same-box A/B (benchmarks/wkv7_ab.py --fwd 1 2 4): B=8
// 0.358 -> 0.331 -> 0.324 ms, B=16 0.647 -> 0.637 -> 0.627 ms.  Variant 1 = the round-2 instantiation.
#define VRWKV_FWD_DEFAULT wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>
#define VRWKV_FWD_ISPLIT wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true, true>      // two workgroups per head
// backward: 5 = wkv7_bwd_v5.h (8 waves; also the sequence-parallel kernel), 8 = wkv7_bwd_v8.h (one dS copy, T chain on P wave 0, full-row LDS-DMA),
// 9 = 8 with the score pieces a step ahead on the P waves.  (6 = the round-3 kernel, experiment builds only: benchmarks/experiments/wkv7_bwd_v6_kernel.h)
constexpr int BWD_DEFAULT = 9;
// wkv7_bwd_v8.h forms 32-bit byte offsets inside a tensor (the largest is the fp32 `sa`): a launch whose tensors reach this many bytes is cut into
// batch slices below it (every tensor of the op is batch-major, so a slice is the same launch on offset pointers); a single sample that large goes
// to wkv7_bwd_v5.h, which addresses with 64 bits.  vrwkv_wkv7_set_backward_slice_limit lowers it for tests.
std::atomic<unsigned long long> g_slice_limit{1ull << 32};

// The kernel a forward launch of `heads` = B x H workgroups uses under override `forced` (-1: none): 7 = wkv7_fwd_v4.h (full-row memory traffic)
// when the heads alone give every CU a workgroup pair's worth of work, 6 = wkv7_fwd_v3.h with two workgroups per head below that, else the forced
// instantiation of wkv7_fwd_v3.h.  The stateful entry (vrwkv_wkv7_forward_state_bf16) follows the same rule; its A/B overrides 1..5 all mean 4
// (the default instantiation of wkv7_fwd_v3.h: the others have no state arguments).
int resolve_fwd(long heads, int forced, bool stateful) {
    if (forced == 7 || (forced == -1 && heads > FWD_ISPLIT_MAX_HEADS)) return 7;
    if (forced == -1 || forced == 6) return 6;
    return stateful ? 4 : forced;
}
// backward: variant 9 when the launch has more workgroups than the chip has CUs (measured -0.4 ... -2.3 % at B x H = 384 ... 1024), variant 8
// for a single round of workgroups (B x H <= 256: 9 measured +0.3 ... +1.8 % there); profiles/r4_wkv7_ab.jsonl, r4c_wkv7_ab.jsonl
int resolve_bwd(long heads, int forced) { return forced == -1 ? (heads > 256 ? BWD_DEFAULT : 8) : forced; }
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
    if (variant != -1 && !(variant >= 1 && variant <= 7)) return VRWKV_EINVAL;   // 1..5: A/B instantiations of wkv7_fwd_v3.h; 6: its two-workgroups-per-head form; 7: wkv7_fwd_v4.h
    g_fwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_set_backward_variant(int variant) {
    bool ok = variant == -1 || variant == 5 || variant == 8 || variant == 9;      // see include/visualrwkv_hip.h
#ifdef VRWKV_V6_EXPERIMENTS
    ok = ok || wkv7exp::is_experiment(variant);
#endif
    if (!ok) return VRWKV_EINVAL;
    g_bwd_variant = variant;
    return VRWKV_OK;
}

int vrwkv_wkv7_set_backward_slice_limit(unsigned long long bytes) {
    g_slice_limit = bytes ? bytes : (1ull << 32);
    return VRWKV_OK;
}

int vrwkv_wkv7_last_variant(int backward) { return backward ? g_last_bwd.load() : g_last_fwd.load(); }

int vrwkv_wkv7_resolve_variant(int kind, int B, int T, int H) {
    if (check_common(B, T, H) != VRWKV_OK || kind < 0 || kind > 2) return VRWKV_EINVAL;
    const long heads = (long)B * H;
    if (kind == 1) return resolve_bwd(heads, g_bwd_variant.load());
    return resolve_fwd(heads, g_fwd_variant.load(), kind == 2);
}

int vrwkv_wkv7_forward_bf16(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, void* y, float* s, float* sa, void* stream) {
    int rc = check_common(B, T, H);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !y || !s || !sa) return VRWKV_EINVAL;
    if (misaligned(w) || misaligned(q) || misaligned(k) || misaligned(v) || misaligned(z) || misaligned(a) ||
        misaligned(y) || misaligned(s) || misaligned(sa))
        return VRWKV_EALIGN;
    const wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                          (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa};
    hipStream_t st = (hipStream_t)stream;
    const long heads = (long)B * H;
    const dim3 grid((unsigned)heads);
    const int var = resolve_fwd(heads, g_fwd_variant.load(), false);
    g_last_fwd = var;
    if (var == 7)        // full-row memory traffic (wkv7_fwd_v4.h)
        return launch_lds(&wkv7f4::fwd_kernel_v4<false>, grid, dim3(512), sizeof(wkv7f4::LdsF4), st, p);
    // chunked MFMA, producer / consumer waves (wkv7_fwd_v3.h); 6: two workgroups per head; 4: its default instantiation
    void (*kern)(wkv7::FwdArgs) = &VRWKV_FWD_DEFAULT;
    if (var == 1) kern = &wkv7c::fwd_kernel_v3<false, false, 1>;
    if (var == 2) kern = &wkv7c::fwd_kernel_v3<false, false, 1, 1, false, false, true>;
    if (var == 3) kern = &wkv7c::fwd_kernel_v3<false, true, 1, 1, false, false, true>;     // + 16-byte transposed stores
    if (var == 5) kern = &wkv7c::fwd_kernel_v3<false, false, 1, 2, false, false, true>;     // + two chunks of prefetch
    if (var == 6) kern = &VRWKV_FWD_ISPLIT;
    return launch_lds(kern, var == 6 ? dim3((unsigned)(2 * heads)) : grid, dim3(512), sizeof(wkv7c::LdsF), st, p);
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
    const wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                          (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s_ckpt, sa, nullptr, s0, s_final};
    // same dispatch as the training forward: the full-row kernel when the heads fill the chip (it takes the state arguments and skips the
    // by-products it has no buffers for), two workgroups per head below that
    const long heads = (long)B * H;
    const int var = resolve_fwd(heads, g_fwd_variant.load(), true);
    g_last_fwd = var;
    if (var == 7) return launch_lds(&wkv7f4::fwd_kernel_v4<false>, dim3((unsigned)heads), dim3(512), sizeof(wkv7f4::LdsF4), (hipStream_t)stream, p);
    void (*kern)(wkv7::FwdArgs) = var == 6 ? &VRWKV_FWD_ISPLIT : &VRWKV_FWD_DEFAULT;
    return launch_lds(kern, dim3((unsigned)(var == 6 ? 2 * heads : heads)), dim3(512), sizeof(wkv7c::LdsF), (hipStream_t)stream, p);
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
    hipStream_t st = (hipStream_t)stream;
    int var = resolve_bwd((long)B * H, g_bwd_variant.load());
    // batch slices for the 32-bit offsets of wkv7_bwd_v8.h (see g_slice_limit)
    const unsigned long long per_sample = (unsigned long long)T * H * 64ull * 4ull, limit = g_slice_limit.load();
    int bmax = B;
    if (var >= 8 && var <= 9 && (unsigned long long)B * per_sample >= limit) {
        bmax = (int)((limit - 1) / per_sample);
        if (bmax < 1) { var = 5; bmax = B; }
    }
    g_last_bwd = var;
    const size_t act = (size_t)T * H * 64, ckpt = (size_t)H * (T / VRWKV_CHUNK_LEN) * 64 * 64;      // elements per sample
    for (int b0 = 0; b0 < B; b0 += bmax) {
        const int nb = B - b0 < bmax ? B - b0 : bmax;
        const size_t o = (size_t)b0 * act;
        const wkv7::BwdArgs p{T, H, (const uint16_t*)w + o, (const uint16_t*)q + o, (const uint16_t*)k + o, (const uint16_t*)v + o,
                              (const uint16_t*)z + o, (const uint16_t*)a + o, (const uint16_t*)dy + o, s + (size_t)b0 * ckpt, sa + o,
                              (uint16_t*)dw + o, (uint16_t*)dq + o, (uint16_t*)dk + o, (uint16_t*)dv + o, (uint16_t*)dz + o, (uint16_t*)da + o};
        const dim3 grid((unsigned)((long)nb * H));
        int rc2;
#ifdef VRWKV_V6_EXPERIMENTS
        if (wkv7exp::is_experiment(var)) rc2 = wkv7exp::launch(var, grid, st, p); else
#endif
        if (var == 9)        // score pieces a step ahead on the P waves
            rc2 = launch_lds(&wkv7v8::bwd_kernel_v8<false, true>, grid, dim3(768), sizeof(wkv7v8::LdsV8), st, p);
        else if (var == 8)   // one copy of dL/dS, T chain on P wave 0, full-row memory role, 12 waves (wkv7_bwd_v8.h)
            rc2 = launch_lds(&wkv7v8::bwd_kernel_v8<false>, grid, dim3(768), sizeof(wkv7v8::LdsV8), st, p);
        else                 // second-generation schedule, 64-bit addressing (wkv7_bwd_v5.h)
            rc2 = launch_lds(&wkv7v5::bwd_kernel_v5<false, BWD_V5_MODE>, grid, dim3(512), sizeof(wkv7v5::LdsV5), st, p);
        if (rc2) return rc2;
    }
    return VRWKV_OK;
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
    return launch_lds(&wkv7v5::bwd_kernel_v5<false, BWD_V5_MODE, true>, dim3((unsigned)((long)B * H * nseg)), dim3(512), sizeof(wkv7v5::LdsV5), (hipStream_t)stream, p);
}

}  // extern "C"
