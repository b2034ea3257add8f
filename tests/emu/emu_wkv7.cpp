// Runs the WKV7 device kernels (visualrwkv_amd/csrc/wkv7_kernels.h) under the host lockstep
// emulator.  TEST INFRASTRUCTURE ONLY -- built by tests/emu/build.py into tests/emu/libemu_kernels.so.
#include <gfx950_prims.h>   // resolves to tests/emu/gfx950_prims.h (-I order)
#include <wkv7_kernels.h>
#include <wkv7_chunked.h>
#include <wkv7_fwd_v3.h>
#include <wkv7_fwd_v4.h>
#include <wkv7_bwd_v6.h>
#include <wkv7_bwd_v6_kernel.h>   // benchmarks/experiments (the round-3 kernel, A/B partner; kept lane-exact here)
#include <wkv7_bwd_v7.h>   // benchmarks/experiments (A/B partner; kept lane-exact here)
#include <wkv7_bwd_v8.h>
#include <wkv7_bwd_v8x.h>  // benchmarks/experiments: the v8 kernel with its knobs (JTAIL, OPT bits; kept lane-exact here)
#include <wkv7_bwd_v5.h>
#include <wkv6_chunked.h>
#include <wkv6_bwd_v2.h>

extern "C" {

int emu_wkv7_forward(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                     const void* z, const void* a, void* y, float* s, float* sa, int variant) {
    wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa};
    dim3 grid((unsigned)(B * H));
    if (variant == 6) emu::launch(dim3((unsigned)(2 * B * H)), dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true, true>(p); });   // two workgroups per head
    else if (variant == 7) emu::launch(grid, dim3(512), [&] { wkv7f4::fwd_kernel_v4<false>(p); });                                        // full-row memory traffic
    else if (variant == 1) emu::launch(grid, dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1>(p); });                                // round-2 instantiation
    else if (variant == 2) emu::launch(grid, dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1, 1, false, false, true>(p); });   // no Ab / Kb images
    else emu::launch(grid, dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>(p); });                      // default: + tr16 reads
    return 0;
}

int emu_wkv7_forward_state(int B, int T, int H, const void* w, const void* q, const void* k, const void* v, const void* z,
                           const void* a, void* y, const float* s0, float* s_final) {
    wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, nullptr, nullptr, nullptr, s0, s_final};
    emu::launch(dim3((unsigned)(B * H)), dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>(p); });
    return 0;
}

int emu_wkv7_backward_chunked(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                              const void* z, const void* a, const void* dy, const float* s, const float* sa,
                              void* dw, void* dq, void* dk, void* dv, void* dz, void* da, int mode) {
    // mode 6: producer / consumer schedule of 8 waves (wkv7_bwd_v5.h); 7: three-stage wave pipeline (wkv7_bwd_v6.h, the default)
    wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                    (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da};
    static_assert(sizeof(wkv7v5::LdsV5) <= 160 * 1024, "LDS budget");
    const dim3 grid((unsigned)(B * H));
    if (mode == 6) { emu::launch(grid, dim3(512), [&] { wkv7v5::bwd_kernel_v5<false, 2 + 4 + 128>(p); }); return (int)sizeof(wkv7v5::LdsV5); }
    if (mode == 7) { emu::launch(grid, dim3(768), [&] { wkv7v6::bwd_kernel_v6<false>(p); }); return (int)sizeof(wkv7v6::LdsV6); }   // three-stage wave pipeline
    if (mode == 8) { emu::launch(grid, dim3(768), [&] { wkv7v7::bwd_kernel_v7<false>(p); }); return (int)sizeof(wkv7v7::LdsV7); }   // + full-row memory role
    if (mode == 9) { emu::launch(grid, dim3(768), [&] { wkv7v8::bwd_kernel_v8<false>(p); }); return (int)sizeof(wkv7v8::LdsV8); }   // one copy of dS, T on P wave 0
    if (mode == 10) { emu::launch(grid, dim3(768), [&] { wkv7v8::bwd_kernel_v8<false, true>(p); }); return (int)sizeof(wkv7v8::LdsV8); }   // + score pieces a step ahead on the P waves
    if (mode == 11) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, false, true>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }  // tail on the J waves
    if (mode == 12) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, true, true>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // + score pieces a step ahead
    if (mode == 13) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, true, false, 3>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 9 + dealt tile-pair reads + swizzled dS image
    if (mode == 14) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, false, false, 3>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 8 + the same
    if (mode == 15) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, true, false, 4>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 9 + S0 by register prefetch in the J waves
    if (mode == 16) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, false, false, 4>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 8 + the same
    if (mode == 17) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, true, false, 8192>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 9 + full-row tail stores
    if (mode == 18) { emu::launch(grid, dim3(768), [&] { wkv7v8x::bwd_kernel_v8<false, 0, 0, 1, 0, true, 1, true, false, 16384 + 32768>(p); }); return (int)sizeof(wkv7v8x::LdsV8); }   // variant 9 + early S0 hand-back + S0 requested before the prepare
    return -1;
}

int emu_wkv6_forward(int B, int T, int H, const void* r, const void* k, const void* v, const float* ew, const void* u,
                     void* y, float* s) {
    wkv6c::Fwd6Args p{T, H, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, ew, (const uint16_t*)u, (uint16_t*)y, s};
    emu::launch(dim3((unsigned)(B * H)), dim3(256), [&] { wkv6c::fwd6_kernel(p); });
    return (int)sizeof(wkv6c::Lds6F);
}

int emu_wkv6_backward(int B, int T, int H, const void* r, const void* k, const void* v, const float* ew, const void* u,
                      const void* gy, const float* s, void* gr, void* gk, void* gv, void* gw, void* gu) {
    wkv6c::Bwd6Args p{T, H, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, ew, (const uint16_t*)u,
                      (const uint16_t*)gy, s, (uint16_t*)gr, (uint16_t*)gk, (uint16_t*)gv, (uint16_t*)gw, (uint16_t*)gu};
    emu::launch(dim3((unsigned)(B * H)), dim3(256), [&] { wkv6c::bwd6_kernel(p); });
    return (int)sizeof(wkv6c::Lds6B);
}

int emu_wkv6_backward_v2(int B, int T, int H, const void* r, const void* k, const void* v, const float* ew, const void* u,
                         const void* gy, const float* s, void* gr, void* gk, void* gv, void* gw, void* gu) {
    wkv6c::Bwd6Args p{T, H, (const uint16_t*)r, (const uint16_t*)k, (const uint16_t*)v, ew, (const uint16_t*)u,
                      (const uint16_t*)gy, s, (uint16_t*)gr, (uint16_t*)gk, (uint16_t*)gv, (uint16_t*)gw, (uint16_t*)gu};
    emu::launch(dim3((unsigned)(B * H)), dim3(768), [&] { wkv6v2::bwd6_kernel_v2(p); });       // three-role pipeline
    return (int)sizeof(wkv6v2::Lds6V2);
}

}  // extern "C"

#include <lora_wgrad.h>
extern "C" int emu_wgrad_skinny(long M, int Nw, int D, int S, const void* wide, const void* narrow, float* part, void* out, int transposed) {
    const lwg::Args a{M, Nw, D, (const uint16_t*)wide, (const uint16_t*)narrow, part};
    const dim3 grid((unsigned)(Nw / lwg::CT * (D == 64 ? 2 : 1)), (unsigned)S);
    if (D == 32) emu::launch(grid, dim3(256), [&] { lwg::wgrad_kernel<2>(a); });
    else if (D == 64) emu::launch(grid, dim3(256), [&] { lwg::wgrad_kernel<2>(a); });      // two column groups of 32
    else if (D == 96) emu::launch(grid, dim3(256), [&] { lwg::wgrad_kernel<6>(a); });
    else return -1;
    const long n = (long)Nw * D;
    emu::launch(dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), [&] { lwg::reduce_kernel(part, S, Nw, D, transposed, (uint16_t*)out); });
    return 0;
}

extern "C" int emu_wkv7_backward_segments_v5(int B, int T, int H, int nseg, const void* w, const void* q, const void* k, const void* v,
                                             const void* z, const void* a, const void* dy, const float* s, const float* sa,
                                             const float* ds_in, float* ds_out,
                                             void* dw, void* dq, void* dk, void* dv, void* dz, void* da) {
    wkv7::BwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa,
                    (uint16_t*)dw, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da};
    p.ds_in = ds_in; p.ds_out = ds_out; p.nseg = nseg;
    emu::launch(dim3((unsigned)(B * H * nseg)), dim3(512), [&] { wkv7v5::bwd_kernel_v5<false, 2 + 4 + 128, true>(p); });
    return 0;
}

extern "C" int emu_wkv7_forward_state_train(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                                            const void* z, const void* a, void* y, const float* s0, float* s_final,
                                            float* s_ckpt, float* sa) {
    wkv7::FwdArgs p{T, H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                    (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s_ckpt, sa, nullptr, s0, s_final};
    emu::launch(dim3((unsigned)(B * H)), dim3(512), [&] { wkv7c::fwd_kernel_v3<false, false, 1, 1, false, true, true>(p); });
    return 0;
}

#include <wgrad_big.h>
extern "C" int emu_wgrad_big(long M, int N1, int N2, int S, const void* A, const void* B, float* part, void* out) {
    const wgb::Args a{M, N1, N2, S, (const uint16_t*)A, (const uint16_t*)B, part, (uint16_t*)out};
    emu::launch(dim3((unsigned)((N1 / wgb::TM) * (N2 / wgb::TN) * S)), dim3(512), [&] { wgb::wgrad_big_kernel(a); });
    if (S > 1) emu::launch(dim3(64), dim3(256), [&] { wgb::wgrad_big_reduce(part, S, (long)N1 * N2, (uint16_t*)out); });
    return (int)(wgb::STAGES * 2 * wgb::OPB);
}

