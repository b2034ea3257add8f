// WKV7 ("wind_backstepping") operator for gfx950: shared constants and the argument blocks of the kernels.
//
// Math (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-130), per (b,h),
// S in R^{64x64}, rows i = value index, columns j = key index, w_t = exp(-exp(w_raw_t)):
//     sa_t = S_{t-1} z_t ;  S_t = S_{t-1} diag(w_t) + sa_t a_t^T + v_t k_t^T ;  y_t = S_t q_t
// (argument names follow the op schema: z = -kk, a = kk*gate).
// The kernels: wkv7_fwd_v3.h (forward), wkv7_bwd_v6.h (backward), wkv7_bwd_v5.h (sequence-parallel backward and the
// shared backward building blocks), wkv7_step.hip (single-token step).  The first generation (sequential in T, one wave per
// head, VALU formulation: 15-33 % / 8-11 % of the HBM roofline) and the first producer / consumer backward were removed
// from the library in round 3; DESIGN.md section 3.1 keeps their measurements.
#pragma once
#include <gfx950_prims.h>

namespace wkv7 {

constexpr int N = 64;        // head size (RUN_CUDA_RWKV7g hard-codes 64, src/model.py:69)
constexpr int CHUNK = 16;    // checkpoint interval of `s` (CHUNK_LEN, src/model.py:41)

template <int X> struct Log2 { static constexpr int v = 1 + Log2<X / 2>::v; };
template <> struct Log2<1> { static constexpr int v = 0; };

struct FwdArgs {
    int T, H;
    const uint16_t *w, *q, *k, *v, *z, *a;   // bf16 (B,T,H,N)
    uint16_t* y;                             // bf16 (B,T,H,N)
    float* s;                                // f32 (B,H,T/16,N,N)  holds S^T at chunk ends
    float* sa;                               // f32 (B,T,H,N)
    unsigned long long* dbg = nullptr;       // optional: per-phase cycle counts of workgroup 0 (profiling builds)
    // inference / stateful extensions honoured by fwd_kernel_v3 only (s and sa may then be null = not written):
    const float* s0 = nullptr;               // f32 (B,H,N,N) initial state S[i][j] (i = value row, j = key column)
    float* s_final = nullptr;                // f32 (B,H,N,N) state after the last token, same layout
};

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
struct BwdArgs {
    int T, H;
    const uint16_t *w, *q, *k, *v, *z, *a, *dy;   // bf16 (B,T,H,N)
    const float* s;                               // f32 (B,H,T/16,N,N)
    const float* sa;                              // f32 (B,T,H,N)
    uint16_t *dw, *dq, *dk, *dv, *dz, *da;        // bf16 (B,T,H,N)
    unsigned long long* dbg = nullptr;
    // sequence-parallel backward (bwd_kernel_v3<.., TPAR = true>): nseg workgroups per head, each walks a contiguous range
    // of chunks from ds_in[b,h,seg] (dL/dS at the END of its range; null = 0) and leaves dL/dS at the START in ds_out
    const float* ds_in = nullptr;                 // f32 (B,H,nseg,N,N), [i][j]
    float* ds_out = nullptr;
    int nseg = 1;
};


}  // namespace wkv7
