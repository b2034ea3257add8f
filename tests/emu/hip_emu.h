// Host-side lockstep emulator for the gfx950 kernels.  TEST INFRASTRUCTURE ONLY.
//
// Runs a HIP kernel body on the CPU, one ucontext fiber per GPU thread, one workgroup at a
// time.  Fibers switch only at synchronisation points (block barrier, wave-level data exchange),
// which is exactly where a real wave64 needs its lanes to be convergent, so the kernels' lane
// and LDS index math runs unchanged.  `__shared__` becomes `static` (one workgroup is live at a
// time).  Nothing in the product links against this.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
#ifdef EMU_LDS_TRACE
// LDS bank-conflict tracer (benchmarks/lds_conflicts.py): kinds 0 read_b32 1 read_b64 2 read_b128 3 read_b64_tr_b16 4 write_b32 5 write_b64
// 6 write_b128 7 LDS-DMA landing (16 B per lane).  Sites are code addresses (symbolised afterwards from the -g build).
namespace emu { inline void lds_trace(int kind, const void* p, void* pc); }
#define EMU_PC() ({ void* pc_; asm volatile("lea 0(%%rip), %0" : "=r"(pc_)); pc_; })
#define VRWKV_LDS_TRACE(kind, ptr) emu::lds_trace(kind, ptr, EMU_PC());
struct float4 {
    float x, y, z, w;
    float4() = default;
    float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    __attribute__((always_inline)) float4(const float4& o) : x(o.x), y(o.y), z(o.z), w(o.w) { emu::lds_trace(2, &o, EMU_PC()); }
    __attribute__((always_inline)) float4& operator=(const float4& o) {
        emu::lds_trace(2, &o, EMU_PC());
        emu::lds_trace(6, this, EMU_PC());
        x = o.x; y = o.y; z = o.z; w = o.w;
        return *this;
    }
};
#else
#define VRWKV_LDS_TRACE(kind, ptr)
struct float4 { float x, y, z, w; };
#endif
struct float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {

struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
};

struct BlockState {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int nthreads = 0;
    int cur = 0;
    int block_arrived = 0;
    unsigned block_gen = 0;
    int wave_arrived[32] = {0};
    unsigned wave_gen[32] = {0};
    std::vector<uint64_t> xchg;   // 16 x 8-byte slots per thread
    std::function<void()> body;
};

inline BlockState*& blk() { static BlockState* b = nullptr; return b; }
inline emu_uint3& tidx() { static emu_uint3 t; return t; }
inline emu_uint3& bidx() { static emu_uint3 t; return t; }
inline dim3& bdim() { static dim3 t; return t; }
inline dim3& gdim() { static dim3 t; return t; }

inline void yield() {
    BlockState* b = blk();
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}
inline int flat_tid() { return blk()->cur; }

inline void block_barrier() {
    BlockState* b = blk();
    unsigned g = b->block_gen;
    if (++b->block_arrived == b->nthreads) { b->block_arrived = 0; b->block_gen++; }
    else while (b->block_gen == g) yield();
}
inline void wave_barrier() {
    BlockState* b = blk();
    int w = b->cur >> 6;
    int nw = b->nthreads - (w << 6); if (nw > 64) nw = 64;
    unsigned g = b->wave_gen[w];
    if (++b->wave_arrived[w] == nw) { b->wave_arrived[w] = 0; b->wave_gen[w]++; }
    else while (b->wave_gen[w] == g) yield();
}
// per-thread exchange slots (16 x u64) visible to the whole wave
inline uint64_t* slot(int tid) { return &blk()->xchg[(size_t)tid * 16]; }
inline int wave_base() { return blk()->cur & ~63; }

inline void trampoline() {
    BlockState* b = blk();
    b->body();
    b->fibers[b->cur].done = true;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

template <class F>
void launch(dim3 grid, dim3 block, F&& body, size_t stack_bytes = 256 * 1024) {
    BlockState st;
    blk() = &st;
    st.nthreads = (int)(block.x * block.y * block.z);
    st.body = body;
    st.xchg.assign((size_t)st.nthreads * 16, 0);
    bdim() = block; gdim() = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        bidx() = emu_uint3{bx, by, bz};
        st.fibers.clear();
        st.fibers.resize(st.nthreads);
        st.block_arrived = 0;
        for (int w = 0; w < 32; ++w) st.wave_arrived[w] = 0;
        for (int t = 0; t < st.nthreads; ++t) {
            Fiber& f = st.fibers[t];
            f.stack.resize(stack_bytes);
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = f.stack.size();
            f.ctx.uc_link = &st.sched;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        int remaining = st.nthreads;
        while (remaining > 0) {
            for (int t = 0; t < st.nthreads; ++t) {
                Fiber& f = st.fibers[t];
                if (f.done) continue;
                st.cur = t;
                tidx() = emu_uint3{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y),
                                   (unsigned)(t / (block.x * block.y))};
                swapcontext(&st.sched, &f.ctx);
                if (f.done) --remaining;
            }
        }
    }
    blk() = nullptr;
}

}  // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

static inline void __syncthreads() { emu::block_barrier(); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }     // device: v_rsq_f32 (1 ulp); parity to the tests' tolerance
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
