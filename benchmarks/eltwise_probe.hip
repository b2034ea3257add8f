// Launch-shape probe for the step's streaming element-wise kernels on gfx950 (relu^2 forward 1 read : 1 write, relu^2 backward
// 2 : 1 at the benchmark shape 41 984 x 8192 bf16): grid-stride loop over a capped grid (as shipped until round 4) against U
// 16-byte vectors per thread with no loop, plain and non-temporal.  The box's plain copy does 6.2 TB/s with one vector per thread
// and 4.6-5.7 with a grid-stride loop (profiles/r4_mem_role_probe_final_box.jsonl).
//   hipcc --offload-arch=gfx950 -O3 benchmarks/eltwise_probe.hip -o benchmarks/_alt/eltwise_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline float lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ inline float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ inline uint32_t pk(float a, float b) {
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ inline u32x4 f_fwd(u32x4 h) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = fmaxf(lo(h[e]), 0.f), b = fmaxf(hi(h[e]), 0.f); o[e] = pk(a * a, b * b); }
    return o;
}
__device__ inline u32x4 f_bwd(u32x4 h, u32x4 d) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pk(2.f * fmaxf(lo(h[e]), 0.f) * lo(d[e]), 2.f * fmaxf(hi(h[e]), 0.f) * hi(d[e]));
    return o;
}
template <bool BWD>
__global__ __launch_bounds__(256) void k_stride(long n, const u32x4* __restrict__ h, const u32x4* __restrict__ d, u32x4* __restrict__ o) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) o[i] = BWD ? f_bwd(h[i], d[i]) : f_fwd(h[i]);
}
template <bool BWD, int U, bool NT>
__global__ __launch_bounds__(256) void k_flat(long n, const u32x4* __restrict__ h, const u32x4* __restrict__ d, u32x4* __restrict__ o) {
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    u32x4 a[U], b[U];
#pragma unroll
    for (int q = 0; q < U; ++q) if (base + q * 256 < n) {
        a[q] = NT ? __builtin_nontemporal_load(h + base + q * 256) : h[base + q * 256];
        if (BWD) b[q] = NT ? __builtin_nontemporal_load(d + base + q * 256) : d[base + q * 256];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) if (base + q * 256 < n) {
        const u32x4 r = BWD ? f_bwd(a[q], b[q]) : f_fwd(a[q]);
        if (NT) __builtin_nontemporal_store(r, o + base + q * 256); else o[base + q * 256] = r;
    }
}
// ---- a token-row kernel with post_bwd's access mix: 6 rows read, 5 written per token (C = 2048: 256 threads x 16 B), trivial arithmetic.
// MODE 0: contiguous token range per workgroup (G workgroups), 1: tokens blockIdx.x + k G, 2: the same with the next token's rows
// requested before the current token's stores, 3: one token per workgroup (G = ntok)
struct Rows { const u32x4* in[6]; u32x4* out[5]; };
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k_rows(long ntok, Rows p) {
    const long G = gridDim.x;
    long lo, hi, step;
    if (MODE == 0) { lo = ntok * blockIdx.x / G; hi = ntok * (blockIdx.x + 1) / G; step = 1; }
    else { lo = blockIdx.x; hi = ntok; step = G; }
    auto ld = [&](int a, long n) { const u32x4* q = p.in[a] + n * 256 + threadIdx.x; return NT ? __builtin_nontemporal_load(q) : *q; };
    u32x4 nx[6];
    if (MODE == 2) for (int a = 0; a < 6; ++a) nx[a] = ld(a, lo);
    for (long n = lo; n < hi; n += step) {
        u32x4 v[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = MODE == 2 ? nx[a] : ld(a, n);
        if (MODE == 2 && n + step < hi) {
#pragma unroll
            for (int a = 0; a < 6; ++a) nx[a] = ld(a, n + step);
        }
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const u32x4 r = f_bwd(v[a], v[a + 1]);
            u32x4* q = p.out[a] + n * 256 + threadIdx.x;
            if (NT) __builtin_nontemporal_store(r, q); else *q = r;
        }
    }
}
template <class F>
static void run(const char* name, long bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    printf("{\"probe\": \"%s\", \"ms\": %.4f, \"GBps\": %.1f}\n", name, ms, bytes / ms * 1e-6);
}
int main() {
    const long n = 41984L * 8192 / 8;              // 16-byte vectors
    u32x4 *h, *d, *o;
    hipMalloc(&h, n * 16); hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
    hipMemset(h, 0x3f, n * 16); hipMemset(d, 0x3e, n * 16);
#define FLAT(BWD, U, NT, label) run(label, (BWD ? 3 : 2) * n * 16, [&] { hipLaunchKernelGGL((k_flat<BWD, U, NT>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, n, h, d, o); });
    for (int blocks : {1024, 4096, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "relusq fwd grid-stride %d blocks", blocks);
        run(nm, 2 * n * 16, [&] { hipLaunchKernelGGL(k_stride<false>, dim3(blocks), dim3(256), 0, 0, n, h, d, o); });
        snprintf(nm, sizeof nm, "relusq bwd grid-stride %d blocks", blocks);
        run(nm, 3 * n * 16, [&] { hipLaunchKernelGGL(k_stride<true>, dim3(blocks), dim3(256), 0, 0, n, h, d, o); });
    }
    FLAT(false, 1, false, "relusq fwd 1 vector / thread") FLAT(false, 2, false, "relusq fwd 2 vectors / thread") FLAT(false, 4, false, "relusq fwd 4 vectors / thread")
    FLAT(false, 1, true, "relusq fwd 1 vector / thread nontemporal") FLAT(false, 4, true, "relusq fwd 4 vectors / thread nontemporal")
    FLAT(true, 1, false, "relusq bwd 1 vector / thread") FLAT(true, 2, false, "relusq bwd 2 vectors / thread") FLAT(true, 4, false, "relusq bwd 4 vectors / thread")
    FLAT(true, 1, true, "relusq bwd 1 vector / thread nontemporal") FLAT(true, 4, true, "relusq bwd 4 vectors / thread nontemporal")
    {
        const long ntok = 41984;
        Rows r;
        u32x4* buf;
        hipMalloc(&buf, 11 * ntok * 256 * 16);
        hipMemset(buf, 0x3e, 11 * ntok * 256 * 16);
        for (int a = 0; a < 6; ++a) r.in[a] = buf + (long)a * ntok * 256;
        for (int a = 0; a < 5; ++a) r.out[a] = buf + (long)(6 + a) * ntok * 256;
        const long bytes = 11 * ntok * 256 * 16;
#define ROWS(MODE, NT, G, label) run(label, bytes, [&] { hipLaunchKernelGGL((k_rows<MODE, NT>), dim3(G), dim3(256), 0, 0, ntok, r); });
        ROWS(0, false, 1024, "rows 6r5w: contiguous ranges, 1024 wg") ROWS(0, true, 1024, "rows 6r5w: contiguous ranges, 1024 wg, nontemporal")
        ROWS(0, true, 2048, "rows 6r5w: contiguous ranges, 2048 wg, nontemporal")
        ROWS(1, true, 1024, "rows 6r5w: strided tokens, 1024 wg, nontemporal") ROWS(2, true, 1024, "rows 6r5w: strided tokens + prefetch, 1024 wg, nontemporal")
        ROWS(2, true, 2048, "rows 6r5w: strided tokens + prefetch, 2048 wg, nontemporal")
        ROWS(3, false, 41984, "rows 6r5w: one token per wg") ROWS(3, true, 41984, "rows 6r5w: one token per wg, nontemporal")
        ROWS(1, true, 10496, "rows 6r5w: strided, 4 tokens per wg, nontemporal") ROWS(1, true, 5248, "rows 6r5w: strided, 8 tokens per wg, nontemporal")
        ROWS(0, true, 10496, "rows 6r5w: contiguous, 4 tokens per wg, nontemporal") ROWS(0, true, 5248, "rows 6r5w: contiguous, 8 tokens per wg, nontemporal")
    }
    return 0;
}
