// Host model of visualrwkv_amd/csrc/gfx950_prims.h (same interface).  TEST INFRASTRUCTURE ONLY.
// Implements the wave64 cross-lane and MFMA semantics the kernels rely on, on top of hip_emu.h.
// The MFMA lane->element maps here are the ones documented in the device header; the GPU test
// tests/test_mfma_layout.py checks the hardware against the same maps.
#pragma once
#include "hip_emu.h"

#define DEVFN static inline
#define KERNEL_MIN_WAVES(n)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

DEVFN float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
DEVFN float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
DEVFN float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
DEVFN uint32_t f32_to_bf16_bits(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
DEVFN uint32_t pack_bf16x2(float lo, float hi) { return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16); }

DEVFN uint32_t cvt_pk_bf16(float x0, float x1) { return pack_bf16x2(x0, x1); }
DEVFN void split_pk(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    lo = cvt_pk_bf16(x0 - bf16_lo(hi), x1 - bf16_hi(hi));
}

DEVFN float fast_exp(float x) { return expf(x); }
DEVFN float fast_exp2(float x) { return exp2f(x); }
DEVFN float fast_rcp(float x) { return 1.0f / x; }
DEVFN float fast_log(float x) { return logf(x); }
DEVFN float fast_tanh(float x) { return tanhf(x); }
DEVFN float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }

DEVFN int lane_id() { return emu::flat_tid() & 63; }
DEVFN int uniform_i32(int x) { return x; }

// value of `x` held by lane `src` of the calling lane's wave (all lanes must call)
DEVFN uint32_t emu_wave_read(uint32_t x, int src) {
    emu::slot(emu::flat_tid())[0] = x;
    emu::wave_barrier();
    uint32_t r = (uint32_t)emu::slot(emu::wave_base() + (src & 63))[0];
    emu::wave_barrier();
    return r;
}
DEVFN float max3_f32(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVFN f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r = {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; return r; }
DEVFN bool wave_any(bool x) {
    emu::slot(emu::flat_tid())[0] = x ? 1u : 0u;
    emu::wave_barrier();
    bool r = false;
    for (int l = 0; l < 64; ++l) r = r || emu::slot(emu::wave_base() + l)[0] != 0;
    emu::wave_barrier();
    return r;
}
DEVFN float lane_xor(float x, int mask) { return __uint_as_float(emu_wave_read(__float_as_uint(x), lane_id() ^ mask)); }
DEVFN float lane_xor16(float x) { return lane_xor(x, 16); }
DEVFN float lane_xor32(float x) { return lane_xor(x, 32); }
DEVFN float lane_bcast(float x, int src) { return __uint_as_float(emu_wave_read(__float_as_uint(x), src)); }
template <int K> DEVFN float dpp_shr(float x) {
    int l = lane_id();
    uint32_t r = emu_wave_read(__float_as_uint(x), l - K);
    return (l & 15) >= K ? __uint_as_float(r) : 0.f;
}
template <int K> DEVFN float dpp_shl(float x) {
    int l = lane_id();
    uint32_t r = emu_wave_read(__float_as_uint(x), l + K);
    return (l & 15) + K <= 15 ? __uint_as_float(r) : 0.f;
}
DEVFN float dpp_row_last(float x) { return __uint_as_float(emu_wave_read(__float_as_uint(x), lane_id() | 15)); }
DEVFN float dpp_shr1_fill(float x, float fill) {
    int l = lane_id();
    uint32_t r = emu_wave_read(__float_as_uint(x), l - 1);
    return (l & 15) >= 1 ? __uint_as_float(r) : fill;
}
DEVFN float lane_xor1(float x) { return lane_xor(x, 1); }
DEVFN float lane_xor2(float x) { return lane_xor(x, 2); }
DEVFN float lane_half_mirror(float x) { int l = lane_id(); return __uint_as_float(emu_wave_read(__float_as_uint(x), (l & ~7) | (7 - (l & 7)))); }
DEVFN float lane_mirror(float x) { int l = lane_id(); return __uint_as_float(emu_wave_read(__float_as_uint(x), (l & ~15) | (15 - (l & 15)))); }

template <int SEL> DEVFN float quad_perm(float x) {
    int l = lane_id();
    return __uint_as_float(emu_wave_read(__float_as_uint(x), (l & ~3) | ((SEL >> (2 * (l & 3))) & 3)));
}
DEVFN f32x4 quad_transpose(f32x4 x) {
    const int m = lane_id();
    const bool o1 = m & 1, o2 = m & 2;
    float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3];
    {
        const float s01 = o1 ? a0 : a1, s23 = o1 ? a2 : a3;
        const float r01 = lane_xor1(s01), r23 = lane_xor1(s23);
        a0 = o1 ? r01 : a0; a1 = o1 ? a1 : r01;
        a2 = o1 ? r23 : a2; a3 = o1 ? a3 : r23;
    }
    {
        const float s02 = o2 ? a0 : a2, s13 = o2 ? a1 : a3;
        const float r02 = lane_xor2(s02), r13 = lane_xor2(s13);
        a0 = o2 ? r02 : a0; a2 = o2 ? a2 : r02;
        a1 = o2 ? r13 : a1; a3 = o2 ? a3 : r13;
    }
    f32x4 y = {a0, a1, a2, a3};
    return y;
}

template <int LOG2> DEVFN float group_sum(float x) {
    if (LOG2 >= 1) x += lane_xor1(x);
    if (LOG2 >= 2) x += lane_xor2(x);
    if (LOG2 >= 3) x += lane_half_mirror(x);
    if (LOG2 >= 4) x += lane_mirror(x);
    if (LOG2 >= 5) x += lane_xor(x, 16);
    if (LOG2 >= 6) x += lane_xor(x, 32);
    return x;
}

// ---- MFMA models: k-ordered fmaf chains (bitwise what the f32 MFMA does; bf16 products are exact in f32)
DEVFN f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c) {
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    float* s = (float*)emu::slot(me); s[0] = a; s[1] = b;
    emu::wave_barrier();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av = ((float*)emu::slot(wb + k * 16 + row))[0];
            float bv = ((float*)emu::slot(wb + k * 16 + col))[1];
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
DEVFN f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    float* s = (float*)emu::slot(me); s[0] = a; s[1] = b;
    emu::wave_barrier();
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = ((float*)emu::slot(wb + k * 32 + row))[0];
            float bv = ((float*)emu::slot(wb + k * 32 + col))[1];
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
DEVFN f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    short* s = (short*)emu::slot(me);
    for (int e = 0; e < 8; ++e) { s[e] = a[e]; s[8 + e] = b[e]; }
    emu::wave_barrier();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            short av = ((short*)emu::slot(wb + (k >> 3) * 16 + row))[k & 7];
            short bv = ((short*)emu::slot(wb + (k >> 3) * 16 + col))[8 + (k & 7)];
            acc += bf16_to_f32((uint16_t)av) * bf16_to_f32((uint16_t)bv);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
DEVFN f32x4 mfma_16x16x16_bf16(bf16x4 a, bf16x4 b, f32x4 c) {
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    short* s = (short*)emu::slot(me);
    for (int e = 0; e < 4; ++e) { s[e] = a[e]; s[4 + e] = b[e]; }
    emu::wave_barrier();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            short av = ((short*)emu::slot(wb + (k >> 2) * 16 + row))[k & 3];
            short bv = ((short*)emu::slot(wb + (k >> 2) * 16 + col))[4 + (k & 3)];
            acc += bf16_to_f32((uint16_t)av) * bf16_to_f32((uint16_t)bv);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
DEVFN bf16x4 mk4(uint2 u) { bf16x4 r; __builtin_memcpy(&r, &u, 8); return r; }
DEVFN f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    short* s = (short*)emu::slot(me);
    for (int e = 0; e < 8; ++e) { s[e] = a[e]; s[8 + e] = b[e]; }
    emu::wave_barrier();
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            short av = ((short*)emu::slot(wb + (k >> 3) * 32 + row))[k & 7];
            short bv = ((short*)emu::slot(wb + (k >> 3) * 32 + col))[8 + (k & 7)];
            acc += bf16_to_f32((uint16_t)av) * bf16_to_f32((uint16_t)bv);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}

DEVFN char* dyn_lds() { static __attribute__((aligned(256))) char buf[160 * 1024]; return buf; }
#ifdef EMU_LDS_TRACE
#include <vector>
namespace emu {
struct LdsRec { uint32_t off; uint16_t tid; uint16_t kind; void* pc; };
inline std::vector<LdsRec>& lds_recs() { static std::vector<LdsRec> v; return v; }
inline bool& lds_trace_on() { static bool on = false; return on; }
inline void lds_trace(int kind, const void* p, void* pc) {
    if (!lds_trace_on() || !blk()) return;
    const char* b = dyn_lds();
    if ((const char*)p < b || (const char*)p >= b + 160 * 1024) return;       // global memory / stack
    lds_recs().push_back(LdsRec{(uint32_t)((const char*)p - b), (uint16_t)flat_tid(), (uint16_t)kind, pc});
}
}  // namespace emu
#endif
// LDS "addresses" as 32-bit values: offsets from the start of the (single) dynamic LDS array
DEVFN unsigned lds_addr_u32(const void* lds_ptr) { return (unsigned)((const char*)lds_ptr - dyn_lds()); }
DEVFN char* emu_lds_from_u32(unsigned a) { return dyn_lds() + a; }

template <int P> DEVFN void wave_priority() {}
DEVFN unsigned long long clock64_() { return 0; }
DEVFN unsigned long long realtime64_() { return 0; }
DEVFN void block_sync() { emu::block_barrier(); }
DEVFN void block_sync_lds() { emu::block_barrier(); }
DEVFN void wave_lds_fence() { emu::wave_barrier(); }
DEVFN uint2 lds_read_tr16(const uint16_t* p) {
    VRWKV_LDS_TRACE(3, p)
    int me = emu::flat_tid(), wb = emu::wave_base(), l = me & 63;
    uint16_t* s = (uint16_t*)emu::slot(me);
    for (int e = 0; e < 4; ++e) s[e] = p[e];
    emu::wave_barrier();
    uint16_t o[4];
    const int grp = l & 48, i = l & 15;
    for (int e = 0; e < 4; ++e) o[e] = ((uint16_t*)emu::slot(wb + grp + 4 * e + (i >> 2)))[i & 3];
    emu::wave_barrier();
    uint2 r;
    r.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
    r.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
    return r;
}
DEVFN void lds_dma16(const void* gsrc, void* lds_wave_base) { VRWKV_LDS_TRACE(7, (char*)lds_wave_base + 16 * (emu::flat_tid() & 63)) memcpy((char*)lds_wave_base + 16 * (emu::flat_tid() & 63), gsrc, 16); }
DEVFN void lds_dma16_sbase(const void* uniform_base, unsigned lane_byte_off, void* lds_wave_base) {
    VRWKV_LDS_TRACE(7, (char*)lds_wave_base + 16 * (emu::flat_tid() & 63))
    memcpy((char*)lds_wave_base + 16 * (emu::flat_tid() & 63), (const char*)uniform_base + lane_byte_off, 16);
}
DEVFN char* emu_lds_from_u32(unsigned a);
DEVFN const void* uniform_ptr(const void* ptr) { return ptr; }
template <int IMM> DEVFN void lds_dma16_lean(const void* uniform_base, unsigned lane_byte_off, unsigned lds_dst_uniform) {
    VRWKV_LDS_TRACE(7, emu_lds_from_u32(lds_dst_uniform) + IMM + 16 * (emu::flat_tid() & 63))
    memcpy(emu_lds_from_u32(lds_dst_uniform) + IMM + 16 * (emu::flat_tid() & 63), (const char*)uniform_base + lane_byte_off + IMM, 16);   // the immediate moves both ends
}
template <int IMM, int CPOL> DEVFN void lds_dma16_lean_cp(const void* uniform_base, unsigned lane_byte_off, unsigned lds_dst_uniform) {
    lds_dma16_lean<IMM>(uniform_base, lane_byte_off, lds_dst_uniform);
}
template <int N_> DEVFN void vmem_wait() {}
// host versions of the experiment-only primitives of benchmarks/experiments/wkv7_bwd_v8x.h
#define VRWKV_EMULATED_PRIMS 1
DEVFN void order_after(bf16x8&, bf16x8&, bf16x8&, bf16x8&) {}
template <int IMM, int NOPS = 0> DEVFN void global_load16_inplace(f32x4& dst, const void* uniform_base, unsigned lane_byte_off) {
    memcpy(&dst, (const char*)uniform_base + lane_byte_off + IMM, 16);
}
template <int N_> DEVFN void vmem_wait_for(f32x4&, f32x4&, f32x4&, f32x4&) {}
// empty asm that "redefines" four registers: keeps the compiler from hoisting a derived (e.g. unpacked) form of a loop invariant
DEVFN void pin_vgpr4(uint32_t&, uint32_t&, uint32_t&, uint32_t&) {}
DEVFN void vmem_drain() {}
DEVFN void lds_flag_add(unsigned* cnt) {
    emu::wave_barrier();
    if ((emu::flat_tid() & 63) == 0) *(volatile unsigned*)cnt += 1;
}
DEVFN void lds_flag_wait(unsigned* cnt, unsigned target) {
    while ((int)(*(volatile unsigned*)cnt - target) < 0) emu::yield();
}
