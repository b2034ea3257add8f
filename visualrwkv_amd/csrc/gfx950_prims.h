// gfx950 (CDNA4 / MI355X) device primitives used by the kernels in this directory.
//
// Everything here is wave64-specific.  The kernels only ever call these wrappers (never the
// raw builtins), which keeps the lane/fragment conventions in one place; tests/emu/ holds a
// host-side model of the SAME interface (test infrastructure) that the CPU test-suite uses to
// run the kernels' index math in lockstep without a GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVFN __device__ __forceinline__
// kernel attribute: keep the register count low enough for at least n waves per SIMD (the host emulator's gfx950_prims.h defines it away)
#define KERNEL_MIN_WAVES(n) __attribute__((amdgpu_waves_per_eu(n)))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- bf16 <-> fp32
DEVFN float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }          // element 0 of a packed pair
DEVFN float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }  // element 1
DEVFN float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN preserved (same result as __float2bfloat16_rn)
DEVFN uint32_t f32_to_bf16_bits(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
DEVFN uint32_t pack_bf16x2(float lo, float hi) { return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16); }

// packed RNE conversion (one v_cvt_pk_bf16_f32): element 0 = x0 (low half), element 1 = x1
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
DEVFN uint32_t cvt_pk_bf16(float x0, float x1) {
    f32x2_t x = {x0, x1};
    bf16x2_t b = __builtin_convertvector(x, bf16x2_t);
    return *reinterpret_cast<uint32_t*>(&b);
}
// split a pair of f32 into hi = bf16(x) and lo = bf16(x - hi)  (x ~= hi + lo to ~2^-17 relative)
DEVFN void split_pk(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    lo = cvt_pk_bf16(x0 - bf16_lo(hi), x1 - bf16_hi(hi));
}

// ---------------------------------------------------------------- math
DEVFN float fast_exp(float x) { return __expf(x); }          // v_exp_f32 path
DEVFN float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32 (no range handling: |x| small here)
DEVFN float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEVFN float fast_log(float x) { return __logf(x); }
DEVFN float fast_tanh(float x) { return tanhf(x); }
DEVFN float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

// ---------------------------------------------------------------- cross-lane (wave64)
// DPP within rows of 16 lanes.  Controls: quad_perm 0x00-0xff, row_shr 0x110+n, row_ror 0x120+n,
// row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL> DEVFN float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// row_shr:K inside each 16-lane row; lanes whose source falls outside the row read 0
// (bound_ctrl: out-of-row sources read 0 without a separate zero-initialised destination register)
template <int K> DEVFN float dpp_shr(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 + K, 0xf, 0xf, true));
}
// row_shl:K (lane t reads lane t+K of its 16-lane row; 0 beyond the row end)
template <int K> DEVFN float dpp_shl(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 + K, 0xf, 0xf, true));
}
// lane 15 of every 16-lane row broadcast to the whole row (row_newbcast:15, gfx90a+): one VALU op instead of ds_bpermute
DEVFN float dpp_row_last(float x) { return dpp_mov<0x15F>(x); }
// row_shr:1 with `fill` for the first lane of every row
DEVFN float dpp_shr1_fill(float x, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x111, 0xf, 0xf, false));
}
DEVFN float lane_xor1(float x) { return dpp_mov<0xB1>(x); }        // quad_perm [1,0,3,2]
DEVFN float lane_xor2(float x) { return dpp_mov<0x4E>(x); }        // quad_perm [2,3,0,1]
DEVFN float lane_half_mirror(float x) { return dpp_mov<0x141>(x); }  // i <-> 7-i   inside each 8 lanes
DEVFN float lane_mirror(float x) { return dpp_mov<0x140>(x); }       // i <-> 15-i  inside each 16 lanes
// arbitrary xor partner (ds_bpermute / permlane paths, chosen by the compiler)
DEVFN float lane_xor(float x, int mask) { return __shfl_xor(x, mask, 64); }
// exchange with the lane 16 / 32 away as VALU ops (v_permlane16_swap / v_permlane32_swap, gfx950) instead of
// an LDS round trip (ds_bpermute): swap(vdst, src) exchanges vdst's odd rows (upper half) with src's even rows
// (lower half); fed with two copies of x, the partner's value lands in result[0] for the odd/upper lanes and
// in result[1] for the even/lower ones.
DEVFN float lane_xor16(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
DEVFN float lane_xor32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
DEVFN float lane_bcast(float x, int src_lane) { return __shfl(x, src_lane, 64); }
DEVFN int lane_id() { return (int)(threadIdx.x & 63); }
// max of three: one v_max3_f32 when the compiler may assume no NaNs (attention.hip is built with -fno-honor-nans;
// otherwise every fmaxf of an MFMA result gets a canonicalising v_max x,x,x first).  NOT inline asm on purpose: the
// hazard recogniser does not look inside asm, and a VALU read of an in-flight MFMA result needs software wait states
// -- an asm v_max3 straight on MFMA accumulators returned garbage on gfx950.
DEVFN float max3_f32(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVFN f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
DEVFN bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; }     // wave-uniform result
// value is wave-uniform by construction (e.g. threadIdx.x >> 6): make that provable -> scalar branches
DEVFN int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }

// arbitrary permutation inside every quad of lanes: lane m reads lane (SEL >> 2m) & 3 of its quad
template <int SEL> DEVFN float quad_perm(float x) { return dpp_mov<SEL>(x); }

// 4x4 transpose inside every quad of lanes: in: lane m (= lane&3) holds x[r], out: lane m holds x_r[m] of lane r
// (two butterfly stages of quad_perm DPP + selects).  Turns "4 consecutive rows, one column per lane" MFMA
// fragments into "one row, 4 consecutive columns per lane", i.e. 16-byte stores.
DEVFN f32x4 quad_transpose(f32x4 x) {
    const int m = lane_id();
    const bool o1 = m & 1, o2 = m & 2;
    float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3];
    {   // stage 1: partner lane^1, element index bit 0
        const float s01 = o1 ? a0 : a1, s23 = o1 ? a2 : a3;
        const float r01 = lane_xor1(s01), r23 = lane_xor1(s23);
        a0 = o1 ? r01 : a0; a1 = o1 ? a1 : r01;
        a2 = o1 ? r23 : a2; a3 = o1 ? a3 : r23;
    }
    {   // stage 2: partner lane^2, element index bit 1
        const float s02 = o2 ? a0 : a2, s13 = o2 ? a1 : a3;
        const float r02 = lane_xor2(s02), r13 = lane_xor2(s13);
        a0 = o2 ? r02 : a0; a2 = o2 ? a2 : r02;
        a1 = o2 ? r13 : a1; a3 = o2 ? a3 : r13;
    }
    f32x4 y = {a0, a1, a2, a3};
    return y;
}

// Sum over the 2^LOG2 lanes that share the high lane bits (all-reduce: every lane gets the sum).
template <int LOG2> DEVFN float group_sum(float x) {
    if (LOG2 >= 1) x += lane_xor1(x);
    if (LOG2 >= 2) x += lane_xor2(x);
    if (LOG2 >= 3) x += lane_half_mirror(x);
    if (LOG2 >= 4) x += lane_mirror(x);
    if (LOG2 >= 5) x += lane_xor16(x);
    if (LOG2 >= 6) x += lane_xor32(x);
    return x;
}

// ---------------------------------------------------------------- MFMA (fragment maps: see DESIGN.md)
// 16x16x4 f32:  A: lane l holds A[i=l&15][k=l>>4];  B: lane l holds B[k=l>>4][j=l&15];
//               C/D reg r: row = (l>>4)*4 + r, col = l&15.
DEVFN f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// 32x32x2 f32:  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; C/D reg r: row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31
DEVFN f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// 16x16x32 bf16: A[i=l&15][k=(l>>4)*8+e], B[k=(l>>4)*8+e][j=l&15], e=0..7; C/D as 16x16x4.
DEVFN f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// 16x16x16 bf16 (the CDNA3 form, still present): A[i=l&15][k=(l>>4)*4+e], B[k=(l>>4)*4+e][j=l&15], e=0..3 -- the k index
// of a lane is exactly the row index of its C/D registers, so an accumulator tile is a B operand as it stands.
DEVFN f32x4 mfma_16x16x16_bf16(bf16x4 a, bf16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
DEVFN bf16x4 mk4(uint2 u) { return __builtin_bit_cast(bf16x4, u); }
// 32x32x16 bf16: A[i=l&31][k=(l>>5)*8+e], B[k=(l>>5)*8+e][j=l&31]; C/D as 32x32x2.
DEVFN f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------- dynamic LDS
// All LDS of a kernel that needs more than the 64 KB static limit lives in this one array.
extern __shared__ __attribute__((aligned(16))) char vrwkv_dyn_lds[];
DEVFN char* dyn_lds() { return vrwkv_dyn_lds; }
// Hook of the host emulator's LDS bank-conflict tracer (benchmarks/lds_conflicts.py); nothing on the device.
#define VRWKV_LDS_TRACE(kind, ptr)

// static wave priority (0..3); scalar, ignores EXEC: call only under wave-uniform control flow
template <int P> DEVFN void wave_priority() { __builtin_amdgcn_s_setprio(P); }

// shader clock (s_memtime), for in-kernel phase timing
DEVFN unsigned long long clock64_() { return __builtin_readcyclecounter(); }
// constant-rate counter (s_memrealtime, 100 MHz): shader cycles / its ticks = the clock the kernel actually ran at
DEVFN unsigned long long realtime64_() { return (unsigned long long)wall_clock64(); }

// ---------------------------------------------------------------- LDS transpose read (gfx950)
// ds_read_b64_tr_b16: every lane supplies the LDS address of 4 consecutive 16-bit elements; inside each group of 16
// lanes the 16 x 4 elements are transposed: lane i receives element (i & 3) of lanes 4e + (i >> 2), e = 0..3.
// With lane i pointing at row (i >> 2), columns 4 (i & 3) .. +3 of a row-major [4][16] block (any row stride), lane i
// gets column i of the block, rows 0..3 -- an MFMA operand whose k index runs along the rows of a row-major image,
// with no transposed copy in LDS.  Checked on hardware by tests/test_probe_gpu.py.
typedef short s16x4 __attribute__((ext_vector_type(4)));
DEVFN uint2 lds_read_tr16(const uint16_t* p) {
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)p);   // low 32 bits of a generic LDS pointer = LDS offset
    return __builtin_bit_cast(uint2, v);
}

// ---------------------------------------------------------------- LDS-DMA (global -> LDS without passing through VGPRs)
// global_load_lds_dwordx4: every lane supplies its own 16-byte global source; the destination is wave-uniform:
// lane l lands at lds_base + 16 l.  Completion is counted by vmcnt of the issuing wave; readers in other waves need
// that wave's s_waitcnt vmcnt followed by a workgroup barrier (vmem_drain() + block_sync_lds()).
// Issued as inline asm on purpose: hipcc cannot tell which LDS bytes a DMA it knows about will write (all LDS here is one
// dynamic array), so it puts s_waitcnt vmcnt(0) in front of every later LDS access of the wave -- which also drains the
// wave's register prefetch loads (measured: +2k cycles per chunk in the WKV7 backward's producers).  The asm form is
// invisible to its counters; the kernel waits for it explicitly.  M0 carries the LDS address (saved and restored).
DEVFN void lds_dma16(const void* gsrc, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// the scalar-base form: wave-uniform 64-bit base in SGPRs + one 32-bit byte offset per lane (no 64-bit address arithmetic on
// the VALU; the offset register can be kept across steps while the base advances on the scalar unit)
DEVFN void lds_dma16_sbase(const void* uniform_base, unsigned lane_byte_off, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
    const unsigned long long ub = (unsigned long long)(uintptr_t)uniform_base;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)ub), bhi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
    const unsigned long long base = ((unsigned long long)bhi << 32) | blo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base), "s"(dst) : "memory");
}
// the lean form for inner loops: scalar base, 32-bit lane offset, compile-time byte offset (0 .. 4095) that the hardware adds to BOTH
// the global address and the LDS address (lane l lands at M0 + IMM + 16 l; checked on gfx950: tests/test_probe_gpu.py), and M0
// is NOT saved / restored (it is declared clobbered instead: the compiler keeps nothing in M0 here -- gfx9 LDS instructions do not read
// it -- so the clobber costs nothing today and stays correct if a later compiler or code change does use M0), which saves two scalar moves per request: on gfx950 every instruction
// of any class takes an issue slot of its SIMD (profiles/r4_wkv7_pmc_v6_v8.txt), so they cost what VALU instructions cost
template <int IMM> DEVFN void lds_dma16_lean(const void* uniform_base, unsigned lane_byte_off, unsigned lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst_uniform), "n"(IMM) : "memory", "m0");
}
// a wave-uniform pointer the compiler keeps in VGPRs (derived next to per-lane arithmetic) -> SGPR pair, for the "s" operands above
DEVFN const void* uniform_ptr(const void* ptr) {
    const unsigned long long u = (unsigned long long)(uintptr_t)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
DEVFN unsigned lds_addr_u32(const void* lds_ptr) { return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_ptr); }
// wait until at most N of this wave's vector-memory operations (loads, LDS-DMA and stores, in issue order) are outstanding
template <int N_> DEVFN void vmem_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
// empty asm that "redefines" four registers: keeps the compiler from hoisting a derived (e.g. unpacked) form of a loop invariant
DEVFN void pin_vgpr4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
DEVFN void vmem_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------- sync
DEVFN void block_sync() { __syncthreads(); }
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for
// global prefetch loads issued for a later chunk (measured: ~2.9k cycles per barrier in the WKV7 forward).
// Global loads stay tracked by the compiler (it waits before their first use); global stores need no ordering
// against other waves of the workgroup here.
DEVFN void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// One-directional hand-off between groups of waves of a workgroup through a monotonically increasing counter in
// LDS -- s_barrier makes EVERY wave of the workgroup wait, a flag only the waves that consume the data (gfx950 has no
// named barriers).  lds_flag_add: whole wave calls it after its LDS writes; lane 0 increments (the LDS unit executes a
// wave's instructions in issue order, so the increment lands after the wave's writes).  lds_flag_wait: spin until
// the counter reaches `target` (wrap-safe compare); the reads issued afterwards see the producers' writes.
#ifndef VRWKV_FLAG_SLEEP
#define VRWKV_FLAG_SLEEP 1          // units of 64 cycles between two polls of a waiting wave
#endif
DEVFN void lds_flag_add(unsigned* cnt) {
    const unsigned a = (unsigned)(uintptr_t)cnt;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(1u) : "memory");
}
DEVFN void lds_flag_wait(unsigned* cnt, unsigned target) {
    const unsigned a = (unsigned)(uintptr_t)cnt;
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if ((int)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(VRWKV_FLAG_SLEEP);
    }
}
// LDS ops of one wave execute in order; this only stops the compiler from moving LDS accesses
// across the point where lanes exchange data through LDS.
DEVFN void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
