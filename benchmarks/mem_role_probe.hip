// Memory-role probe for the WKV7 kernels on gfx950: what the HBM side of the backward's P role (and of the forward's
// load / store waves) can reach on its own, by ACCESS SHAPE -- the kernels' "lane = token, 8 bytes = 4 channels" loads and
// stores (a wave touches 16 token rows x 32 B per instruction) against full 128-byte rows at 16 B per lane (registers or
// LDS-DMA), at the kernels' occupancy (one or two workgroups per CU, forced through the dynamic LDS size) and with the
// kernels' step structure (one workgroup barrier per 16-token chunk, prefetch one step ahead).  Also the plain copy
// ceilings of the box: the guide's float4 grid-stride copy next to the tiled non-temporal copy of csrc/probe.hip.
// Standalone:
//   hipcc --offload-arch=gfx950 -O3 -I visualrwkv_amd/csrc benchmarks/mem_role_probe.hip -o benchmarks/_alt/mem_role_probe
//   benchmarks/_alt/mem_role_probe [B=16]  > profiles/r4_mem_role_probe.jsonl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <gfx950_prims.h>

constexpr int N = 64, L = 16;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

struct Args {
    const uint16_t* in[7];     // w q k z a v dy   (B,T,H,N) bf16
    const float* sa;           // (B,T,H,N) f32
    const float* s;            // (B,H,T/16,N,N) f32
    uint16_t* out[6];          // dw dq dk dz da dv
    float* sa_out;             // forward: sa
    float* s_out;              // forward: checkpoints
    int T, H;
};

// ------------------------------------------------------------------------------------------------ backward-like P role
// LD: 0 none | 1 narrow (lane = token c16, 4 channels = 8 B; wave w = channels 16w..) | 2 wide registers (16 B per lane,
//     8 lanes = one 128-byte token row) | 3 wide LDS-DMA into a staging image
// ST: 0 none | 1 narrow 8 B x 6 arrays | 2 wide 16 B per lane (data through LDS)
// S0: 16 KB of checkpoint per chunk by LDS-DMA (as the kernel does)
// PF: prefetch distance in steps (register loads)
template <int LD, int ST, bool S0, int PF>
__global__ __launch_bounds__(256) void bwd_role(Args p) {
    char* lds = dyn_lds();
    float* s0img = reinterpret_cast<float*>(lds);                       // 2 x 16 KB
    uint32_t* stage = reinterpret_cast<uint32_t*>(lds + 32768);         // 2 x 18 KB (LD == 3) / 12 KB out image (ST == 2)
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63, w = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const unsigned bh = blockIdx.x;
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    const float* sbase = p.s + (size_t)bh * nchunk * N * N;
    const unsigned narrow_off = (unsigned)c16 * ts + 16u * w + 4u * g;          // elements
    const unsigned wide_row = (unsigned)(lane >> 3), wide_oct = (unsigned)(lane & 7);

    u32x2_t rq[PF][7]; u32x4_t rsa[PF];                 // narrow prefetch queue
    u32x4_t wq[PF][5];                                  // wide prefetch queue (<= 5 x 1 KB instructions per wave)
    u32x4_t acc = {0u, 0u, 0u, 0u};

    auto narrow_fetch = [&](int slot, int c) {
        const size_t u = head_base + (size_t)c * L * ts + narrow_off;
#pragma unroll
        for (int a = 0; a < 7; ++a) rq[slot][a] = *reinterpret_cast<const u32x2_t*>(p.in[a] + u);
        rsa[slot] = *reinterpret_cast<const u32x4_t*>(p.sa + u);
    };
    // wide: 18 instructions of 1 KB per chunk: i = 2a + h (array a, token half h) for the 7 bf16 arrays, 14..17 = sa quarters
    auto wide_addr = [&](int i, int c) -> const void* {
        const size_t cb = head_base + (size_t)c * L * ts;
        if (i < 14) return p.in[i >> 1] + cb + (size_t)(8 * (i & 1) + wide_row) * ts + 8 * wide_oct;
        const int qd = i - 14;                                              // 4 tokens x 256 B per instruction
        return p.sa + cb + (size_t)(4 * qd + (lane >> 4)) * ts + 4 * (lane & 15);
    };
    auto wide_fetch = [&](int slot, int c) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = w + 4 * k;
            if (i < 18) wq[slot][k] = *reinterpret_cast<const u32x4_t*>(wide_addr(i, c));
        }
    };
    auto dma_fetch = [&](int c) {
        uint32_t* st = stage + (c & 1) * (18 * 256);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int i = w + 4 * k;
            if (i < 18) lds_dma16(wide_addr(i, c), st + i * 256);
        }
    };

    if (LD == 1) for (int d = 0; d < PF; ++d) narrow_fetch(d, nchunk - 1 - d > 0 ? nchunk - 1 - d : 0);
    if (LD == 2) for (int d = 0; d < PF; ++d) wide_fetch(d, nchunk - 1 - d > 0 ? nchunk - 1 - d : 0);
    if (LD == 3) dma_fetch(nchunk - 1);

    for (int n = 0; n < nchunk; ++n) {
        const int c = nchunk - 1 - n;
        const int cn = c - PF > 0 ? c - PF : 0;
        // S0 of the next chunk by LDS-DMA (4 KB per wave)
        if (S0 && c > 0) {
            const float* sc = sbase + (size_t)(c - 1) * N * N;
#pragma unroll
            for (int k = 4 * w; k < 4 * w + 4; ++k) {
                const int row = 4 * k + (lane >> 4);
                lds_dma16(sc + row * N + (((lane & 15) ^ (row & 15)) << 2), s0img + (c & 1) * N * N + 4 * k * N);
            }
        }
        // consume the oldest slot, then refill it for chunk c - PF
        if (LD == 1) {
#pragma unroll
            for (int a = 0; a < 7; ++a) { acc[0] ^= rq[0][a][0]; acc[1] ^= rq[0][a][1]; }
            acc ^= rsa[0];
#pragma unroll
            for (int d = 0; d + 1 < PF; ++d) {
#pragma unroll
                for (int a = 0; a < 7; ++a) rq[d][a] = rq[d + 1][a];
                rsa[d] = rsa[d + 1];
            }
            narrow_fetch(PF - 1, cn);
        } else if (LD == 2) {
#pragma unroll
            for (int k = 0; k < 5; ++k) if (w + 4 * k < 18) acc ^= wq[0][k];
#pragma unroll
            for (int d = 0; d + 1 < PF; ++d)
#pragma unroll
                for (int k = 0; k < 5; ++k) wq[d][k] = wq[d + 1][k];
            wide_fetch(PF - 1, cn);
        } else if (LD == 3) {
            // this chunk's staging image landed before the previous barrier; every wave reads its own 8-byte pieces
            const uint32_t* st = stage + (c & 1) * (18 * 256);
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                const u32x2_t v = *reinterpret_cast<const u32x2_t*>(st + a * 512 + c16 * 32 + ((2 * w + (g >> 1)) ^ (c16 & 7)) * 4 + (g & 1) * 2);
                acc[0] ^= v[0]; acc[1] ^= v[1];
            }
            acc ^= *reinterpret_cast<const u32x4_t*>(st + 14 * 256 + c16 * 64 + (4 * w + g) * 4);
            if (c > 0) dma_fetch(c - 1);
        }
        // stores
        if (ST == 1) {
            const size_t u = head_base + (size_t)c * L * ts + narrow_off;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                u32x2_t v = {acc[0] + (unsigned)a, acc[1] ^ acc[2]};
                *reinterpret_cast<u32x2_t*>(p.out[a] + u) = v;
            }
        } else if (ST == 2) {
            // results -> LDS image (8 B per lane and array, as the tail would) -> 12 wide stores of 1 KB per chunk
            uint32_t* oimg = stage + 2 * 18 * 256;                   // 6 x 2 KB
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                u32x2_t v = {acc[0] + (unsigned)a, acc[1] ^ acc[2]};
                *reinterpret_cast<u32x2_t*>(oimg + a * 512 + c16 * 32 + ((2 * w + (g >> 1)) ^ (c16 & 7)) * 4 + (g & 1) * 2) = v;
            }
            block_sync_lds();
            const size_t cb = head_base + (size_t)c * L * ts;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = w + 4 * k;                              // array i >> 1, token half i & 1
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(oimg + (i >> 1) * 512 + (8 * (i & 1) + wide_row) * 32 + wide_oct * 4);
                *reinterpret_cast<u32x4_t*>(p.out[i >> 1] + cb + (size_t)(8 * (i & 1) + wide_row) * ts + 8 * wide_oct) = v;
            }
        }
        if (S0) acc[3] ^= __float_as_uint(s0img[((c + 1) & 1) * N * N + tid]);
        // the S0 DMA (oldest of the step) and the staging DMA have landed; the stores -- and with a two-step queue the newest
        // register loads -- stay in flight
        constexpr int WAITN = (ST == 1 ? 6 : ST == 2 ? 3 : 0) + ((PF > 1 && LD == 1) ? 8 : (PF > 1 && LD == 2) ? 4 : 0);
        if (LD == 3 || S0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        block_sync_lds();
    }
    if (acc[0] == 0x12345678u && acc[3] == 0x9abcdef1u) p.out[0][head_base + tid] = (uint16_t)acc[1];
}

// ------------------------------------------------------------------------------------------------ forward-like roles
// 512 threads: waves 4-7 load 6 arrays (narrow, one chunk ahead), waves 0-3 store y (bf16), sa (f32), s (16 KB per chunk).
// ST: 1 = as shipped (2-byte y stores x4, 4-byte sa x4, 4-byte s x16)   2 = 8-byte y, 16-byte sa, 16-byte s x4 (quad-transposed)
//     3 = full rows through LDS: s as 16 x 1 KB (4 per wave), sa 4 x 1 KB, y 2 x 1 KB
template <int ST, int LD = 1>          // LD: 1 narrow register loads (as shipped) | 3 full rows by LDS-DMA (12 x 1 KB per chunk, 3 per producer wave)
__global__ __launch_bounds__(512) void fwd_role(Args p) {
    char* lds = dyn_lds();
    uint32_t* img = reinterpret_cast<uint32_t*>(lds);
    const int T = p.T, H = p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunk = T / L;
    const unsigned ts = (unsigned)(H * N);
    const unsigned bh = blockIdx.x;
    const size_t head_base = ((size_t)(bh / H) * T * H + (bh % H)) * N;
    if (wave >= 4) {
        const int w = wave - 4;
        const unsigned off = (unsigned)c16 * ts + 16u * w + 4u * g;
        u32x2_t r[6], acc = {0u, 0u};
        auto fetch = [&](int c) {
            const size_t u = head_base + (size_t)c * L * ts + off;
#pragma unroll
            for (int a = 0; a < 6; ++a) r[a] = *reinterpret_cast<const u32x2_t*>(p.in[a] + u);
        };
        uint32_t* stage = img + 8192;                                   // 2 x 12 KB from byte 32768
        const unsigned wide_row = (unsigned)(lane >> 3), wide_oct = (unsigned)(lane & 7);
        auto dma_fetch = [&](int c) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = w + 4 * k;                                // array i >> 1, token half i & 1
                lds_dma16(p.in[i >> 1] + head_base + (size_t)c * L * ts + (size_t)(8 * (i & 1) + wide_row) * ts + 8 * wide_oct, stage + (c & 1) * 3072 + i * 256);
            }
        };
        if (LD == 1) fetch(0); else { dma_fetch(0); vmem_drain(); block_sync_lds(); }
        for (int c = 0; c < nchunk; ++c) {
            if (LD == 1) {
#pragma unroll
                for (int a = 0; a < 6; ++a) { acc[0] ^= r[a][0]; acc[1] ^= r[a][1]; }
                if (c + 1 < nchunk) fetch(c + 1);
            } else {
                if (c + 1 < nchunk) dma_fetch(c + 1);
                const uint32_t* st = stage + (c & 1) * 3072;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(st + a * 512 + c16 * 32 + ((2 * w + (g >> 1)) ^ (c16 & 7)) * 4 + (g & 1) * 2);
                    acc[0] ^= v[0]; acc[1] ^= v[1];
                }
                vmem_drain();
            }
            img[4096 + tid] = acc[0] ^ acc[1];
            block_sync_lds();
        }
        return;
    }
    const int w = wave;
    if (LD != 1) block_sync_lds();
    float* s_c0 = p.s_out + (size_t)bh * nchunk * N * N;
    for (int c = 0; c < nchunk; ++c) {
        const uint32_t seed = img[4096 + 256 + tid];
        float* sa_c = p.sa_out + head_base + (size_t)c * L * ts;
        uint16_t* y_c = p.out[0] + head_base + (size_t)c * L * ts;
        float* s_c = s_c0 + (size_t)c * N * N;
        if (ST == 1) {
            const unsigned o = (unsigned)(4 * g) * ts + 16u * w + c16;
            const unsigned so = (unsigned)(4 * g) * N + 16u * w + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sa_c[o + r * ts] = __uint_as_float(seed + r); y_c[o + r * ts] = (uint16_t)(seed >> r); }
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_c[so + (unsigned)(16 * jb + r) * N] = __uint_as_float(seed ^ (jb * 4 + r));
        } else if (ST == 2) {
            const unsigned o = (unsigned)(4 * g + (c16 & 3)) * ts + 16u * w + (c16 & ~3);
            const unsigned so = (unsigned)(4 * g + (c16 & 3)) * N + 16u * w + (c16 & ~3);
            u32x4_t v = {seed, seed + 1, seed + 2, seed + 3};
            *reinterpret_cast<u32x4_t*>(sa_c + o) = v;
            u32x2_t yv = {seed, seed ^ 5u};
            *reinterpret_cast<u32x2_t*>(y_c + o) = yv;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) { v[0] ^= jb; *reinterpret_cast<u32x4_t*>(s_c + so + (unsigned)(16 * jb) * N) = v; }
        } else {
            // stage through LDS (one 16-byte write per tile, as a quad-transposed fragment would be), then full rows
            u32x4_t v = {seed, seed + 1, seed + 2, seed + 3};
            uint32_t* simg = img + (c & 1) * 2048;                                  // not to scale: timing of the global side only
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) *reinterpret_cast<u32x4_t*>(simg + ((jb * 256 + tid) & 2047) / 4 * 4) = v;
            wave_lds_fence();
            // s: rows j = 16w .. 16w+15 of the [j][i] checkpoint: 256 B per row, 4 rows per instruction, 4 instructions per wave
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 16 * w + 4 * k + (lane >> 4);
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(simg + ((k * 64 + lane) * 4 & 2047));
                *reinterpret_cast<u32x4_t*>(s_c + (size_t)row * N + 4 * (lane & 15)) = x;
            }
            {   // sa: 16 tokens x 256 B: wave w takes tokens 4w..4w+3
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(simg + (lane * 4 & 2047));
                *reinterpret_cast<u32x4_t*>(sa_c + (size_t)(4 * w + (lane >> 4)) * ts + 4 * (lane & 15)) = x;
            }
            if (w < 2) {   // y: 16 tokens x 128 B: waves 0, 1 take 8 tokens each
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(simg + ((lane + 64) * 4 & 2047));
                *reinterpret_cast<u32x4_t*>(y_c + (size_t)(8 * w + (lane >> 3)) * ts + 8 * (lane & 7)) = x;
            }
        }
        block_sync_lds();
    }
}

// ------------------------------------------------------------------------------------------------ copies
__global__ __launch_bounds__(256) void copy_plain(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_tiled(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long nvec) {
    constexpr int U = 8;
    const long ntiles = nvec / (256 * U);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long base = t * 256 * U + threadIdx.x;
        u32x4_t v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = NT ? __builtin_nontemporal_load(src + base + q * 256) : src[base + q * 256];
#pragma unroll
        for (int q = 0; q < U; ++q) { if (NT) __builtin_nontemporal_store(v[q], dst + base + q * 256); else dst[base + q * 256] = v[q]; }
    }
}
__global__ void fill_random(uint32_t* p, long n, uint32_t seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & 0x7fff7fffu) | 0x30003000u;        // finite bf16 pairs / floats
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F> static float time_ms(F&& launch, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best) best = ms / iters;
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, T = 2624, H = 32;
    const size_t elems = (size_t)B * T * H * N;
    Args a{};
    std::vector<void*> all;
    auto alloc = [&](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); all.push_back(p); hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, (uint32_t*)p, (long)(bytes / 4), (uint32_t)all.size()); return p; };
    for (int i = 0; i < 7; ++i) a.in[i] = (const uint16_t*)alloc(elems * 2);
    a.sa = (const float*)alloc(elems * 4);
    a.s = (const float*)alloc(elems * 4 * N / L);
    for (int i = 0; i < 6; ++i) a.out[i] = (uint16_t*)alloc(elems * 2);
    a.sa_out = (float*)alloc(elems * 4);
    a.s_out = (float*)alloc(elems * 4 * N / L);
    a.T = T; a.H = H;
    CK(hipDeviceSynchronize());
    const dim3 grid(B * H);

    // ---- copies
    {
        const long nbytes = 1L << 30, nvec = nbytes / 16;
        const u32x4_t* src = (const u32x4_t*)a.s; u32x4_t* dst = (u32x4_t*)a.s_out;
        for (int gmul : {4, 8, 16, 32, 64}) {
            const float ms = time_ms([&] { hipLaunchKernelGGL(copy_plain, dim3(256 * gmul), dim3(256), 0, 0, src, dst, nvec); });
            printf("{\"probe\": \"copy float4 grid-stride plain (guide)\", \"blocks\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", 256 * gmul, ms, 2.0 * nbytes / ms * 1e-6);
        }
        {
            const float ms = time_ms([&] { hipLaunchKernelGGL(copy_plain, dim3((unsigned)(nvec / 256)), dim3(256), 0, 0, src, dst, nvec); });
            printf("{\"probe\": \"copy float4 one vector per thread plain\", \"blocks\": %ld, \"ms\": %.4f, \"GBps\": %.1f}\n", nvec / 256, ms, 2.0 * nbytes / ms * 1e-6);
        }
        for (int nt = 0; nt < 2; ++nt) {
            const float ms = time_ms([&] {
                if (nt) hipLaunchKernelGGL(copy_tiled<true>, dim3(2048), dim3(256), 0, 0, src, dst, nvec);
                else hipLaunchKernelGGL(copy_tiled<false>, dim3(2048), dim3(256), 0, 0, src, dst, nvec);
            });
            printf("{\"probe\": \"copy tiled 32 KB per workgroup (csrc/probe.hip form)%s\", \"blocks\": 2048, \"ms\": %.4f, \"GBps\": %.1f}\n", nt ? " nontemporal" : " plain", ms, 2.0 * nbytes / ms * 1e-6);
        }
        fflush(stdout);
    }

    const bool only_fwd = argc > 2;
    // ---- backward-like P role
    auto run_bwd = [&](const char* name, auto kern, size_t lds, double bytes_per_elem) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float ms = time_ms([&] { hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a); });
        printf("{\"probe\": \"bwd role: %s\", \"B\": %d, \"lds_KB\": %zu, \"ms\": %.4f, \"bytes_per_elem\": %.0f, \"GBps\": %.1f}\n", name, B, lds / 1024, ms, bytes_per_elem, bytes_per_elem * elems / ms * 1e-6);
        fflush(stdout);
    };
    const size_t L1 = 155 * 1024, L2 = 80 * 1024;
    if (!only_fwd) {
    //                                                        LD ST S0 PF
    run_bwd("narrow loads + S0 dma + narrow stores (as shipped), 1 wg/cu", &bwd_role<1, 1, true, 1>, L1, 46);
    run_bwd("narrow loads + S0 dma + narrow stores, prefetch 2, 1 wg/cu", &bwd_role<1, 1, true, 2>, L1, 46);
    run_bwd("narrow loads + S0 dma + narrow stores, 2 wg/cu", &bwd_role<1, 1, true, 1>, L2, 46);
    run_bwd("narrow loads + S0 dma, no stores, 1 wg/cu", &bwd_role<1, 0, true, 1>, L1, 34);
    run_bwd("narrow loads only, 1 wg/cu", &bwd_role<1, 0, false, 1>, L1, 18);
    run_bwd("S0 dma only, 1 wg/cu", &bwd_role<0, 0, true, 1>, L1, 16);
    run_bwd("narrow stores only, 1 wg/cu", &bwd_role<0, 1, false, 1>, L1, 12);
    run_bwd("wide stores only, 1 wg/cu", &bwd_role<0, 2, false, 1>, L1, 12);
    run_bwd("wide register loads + S0 dma + narrow stores, 1 wg/cu", &bwd_role<2, 1, true, 1>, L1, 46);
    run_bwd("wide register loads + S0 dma + wide stores, 1 wg/cu", &bwd_role<2, 2, true, 1>, L1, 46);
    run_bwd("wide register loads + S0 dma + wide stores, prefetch 2, 1 wg/cu", &bwd_role<2, 2, true, 2>, L1, 46);
    run_bwd("wide LDS-DMA loads + S0 dma + narrow stores, 1 wg/cu", &bwd_role<3, 1, true, 1>, L1, 46);
    run_bwd("wide LDS-DMA loads + S0 dma + wide stores, 1 wg/cu", &bwd_role<3, 2, true, 1>, L1, 46);
    run_bwd("wide LDS-DMA loads + S0 dma + wide stores, 2 wg/cu", &bwd_role<3, 2, true, 1>, L2, 46);
    run_bwd("narrow loads + S0 dma + wide stores, 1 wg/cu", &bwd_role<1, 2, true, 1>, L1, 46);
    run_bwd("wide LDS-DMA loads + S0 dma, no stores, 1 wg/cu", &bwd_role<3, 0, true, 1>, L1, 34);

    }
    // ---- forward-like roles (two workgroups per CU, as the kernel at B = 16)
    auto run_fwd = [&](const char* name, auto kern, size_t lds) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const float ms = time_ms([&] { hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, a); });
        printf("{\"probe\": \"fwd roles: %s\", \"B\": %d, \"lds_KB\": %zu, \"ms\": %.4f, \"bytes_per_elem\": 34, \"GBps\": %.1f}\n", name, B, lds / 1024, ms, 34.0 * elems / ms * 1e-6);
        fflush(stdout);
    };
    run_fwd("narrow loads + scalar stores (as shipped), 2 wg/cu", &fwd_role<1>, L2);
    run_fwd("narrow loads + 16-byte fragment stores, 2 wg/cu", &fwd_role<2>, L2);
    run_fwd("narrow loads + full-row stores through LDS, 2 wg/cu", &fwd_role<3>, L2);
    run_fwd("full-row LDS-DMA loads + scalar stores, 2 wg/cu", &fwd_role<1, 3>, L2);
    run_fwd("full-row LDS-DMA loads + 16-byte fragment stores, 2 wg/cu", &fwd_role<2, 3>, L2);
    run_fwd("full-row LDS-DMA loads + full-row stores through LDS, 2 wg/cu", &fwd_role<3, 3>, L2);
    run_fwd("full-row LDS-DMA loads + full-row stores through LDS, 1 wg/cu", &fwd_role<3, 3>, L1);
    run_fwd("narrow loads + scalar stores (as shipped), 1 wg/cu", &fwd_role<1>, L1);
    run_fwd("narrow loads + full-row stores through LDS, 1 wg/cu", &fwd_role<3>, L1);
    for (void* q : all) hipFree(q);
    return 0;
}
