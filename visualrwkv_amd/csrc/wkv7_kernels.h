// WKV7 ("wind_backstepping") forward / backward for gfx950 -- sequential-in-T, VALU formulation.
//
// Math (reference: VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-130), per (b,h),
// S in R^{64x64}, rows i = value index, columns j = key index, w_t = exp(-exp(w_raw_t)):
//     sa_t = S_{t-1} z_t ;  S_t = S_{t-1} diag(w_t) + sa_t a_t^T + v_t k_t^T ;  y_t = S_t q_t
// (argument names follow the op schema: z = -kk, a = kk*gate).
//
// This is NOT the reference's mapping (one thread per state row, 64-float register rows, smem
// broadcast, two block barriers per token).  Here a wave64 owns a tile of state rows and each
// lane a RL x JL register block of S:
//   * lane = rg * NCG + cg;  cg owns JL contiguous columns, rg owns RL contiguous rows;
//   * reductions over j are DPP butterflies over the NCG low lane bits (no LDS, no barrier);
//   * a head can be split over WPH = 64/(RL*JL) waves (rows are independent in the forward) so
//     that B*H*WPH covers the 1024 SIMDs of an MI355X even at small batch;
//   * inputs are fetched per sub-chunk of SC tokens with 16-byte loads into registers while the
//     previous sub-chunk computes, converted once (bf16->f32, decay = exp(-exp(w))) and parked in
//     LDS as f32; the inner loop reads them with ds_read_b128 (RL=4 rows amortise each read);
//   * one block barrier per sub-chunk (not two per token).
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
// forward
// ------------------------------------------------------------------------------------------
template <int JL, int SC>
__global__ __launch_bounds__(64 * (64 / (4 * JL))) void fwd_kernel(FwdArgs p) {
    constexpr int RL = 4;
    constexpr int NCG = N / JL;              // lanes that share a row set
    constexpr int NRG = 64 / NCG;            // row groups per wave
    constexpr int RW = RL * NRG;             // rows per wave
    constexpr int WPH = N / RW;              // waves per head
    constexpr int NT = 64 * WPH;
    constexpr int NVEC = 6;                  // w q k z a v
    constexpr int SEGS = SC * NVEC * 8;      // 16-byte segments per sub-chunk
    constexpr int LOADS = SEGS / NT;
    static_assert(SEGS % NT == 0, "segment split");
    static_assert((SC * 8) % 64 == 0, "a wave must stay inside one vector");

    __shared__ __attribute__((aligned(16))) float lds[2][SC][NVEC][N];

    const int T = p.T, H = p.H;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cg = lane & (NCG - 1), rg = lane / NCG;
    const int row0 = wave * RW + rg * RL;
    const int col0 = cg * JL;
    const size_t head_base = ((size_t)b * T * H + h) * N;   // + t*H*N + n
    const size_t tstride = (size_t)H * N;

    float S[RL][JL];
#pragma unroll
    for (int r = 0; r < RL; ++r)
#pragma unroll
        for (int j = 0; j < JL; ++j) S[r][j] = 0.f;

    uint4 pre[LOADS];
    auto issue = [&](int t0) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int g = i * NT + tid;
            const int vec = g / (SC * 8), rem = g % (SC * 8), step = rem >> 3, seg = rem & 7;
            const uint16_t* src = vec == 0 ? p.w : vec == 1 ? p.q : vec == 2 ? p.k : vec == 3 ? p.z : vec == 4 ? p.a : p.v;
            pre[i] = *reinterpret_cast<const uint4*>(src + head_base + (size_t)(t0 + step) * tstride + seg * 8);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int g = i * NT + tid;
            const int vec = g / (SC * 8), rem = g % (SC * 8), step = rem >> 3, seg = rem & 7;
            float f[8] = {bf16_lo(pre[i].x), bf16_hi(pre[i].x), bf16_lo(pre[i].y), bf16_hi(pre[i].y),
                          bf16_lo(pre[i].z), bf16_hi(pre[i].z), bf16_lo(pre[i].w), bf16_hi(pre[i].w)};
            if (vec == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fast_exp(-fast_exp(f[e]));
            }
            float4* dst = reinterpret_cast<float4*>(&lds[buf][step][vec][seg * 8]);
            dst[0] = make_float4(f[0], f[1], f[2], f[3]);
            dst[1] = make_float4(f[4], f[5], f[6], f[7]);
        }
    };

    const int nsc = T / SC;
    issue(0);
    park(0);
    block_sync();

    for (int sc = 0; sc < nsc; ++sc) {
        const int buf = sc & 1;
        if (sc + 1 < nsc) issue((sc + 1) * SC);

#pragma unroll 1
        for (int s = 0; s < SC; ++s) {
            const int t = sc * SC + s;
            const float* L = &lds[buf][s][0][0];
            float vv[RL], sav[RL], yv[RL];
            {
                const float4 x = *reinterpret_cast<const float4*>(L + 5 * N + row0);
                vv[0] = x.x; vv[1] = x.y; vv[2] = x.z; vv[3] = x.w;
            }
            // sa_i = sum_j S[i][j] z_j
#pragma unroll
            for (int r = 0; r < RL; ++r) sav[r] = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < JL; j4 += 4) {
                const float4 zz = *reinterpret_cast<const float4*>(L + 3 * N + col0 + j4);
                const float zf[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RL; ++r) sav[r] = fmaf(S[r][j4 + e], zf[e], sav[r]);
            }
#pragma unroll
            for (int r = 0; r < RL; ++r) sav[r] = group_sum<Log2<NCG>::v>(sav[r]);
            // S = S*w + sa*a + v*k ;  y_i = sum_j S[i][j] q_j
#pragma unroll
            for (int r = 0; r < RL; ++r) yv[r] = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < JL; j4 += 4) {
                const float4 ww = *reinterpret_cast<const float4*>(L + 0 * N + col0 + j4);
                const float4 qq = *reinterpret_cast<const float4*>(L + 1 * N + col0 + j4);
                const float4 kk = *reinterpret_cast<const float4*>(L + 2 * N + col0 + j4);
                const float4 aa = *reinterpret_cast<const float4*>(L + 4 * N + col0 + j4);
                const float wf[4] = {ww.x, ww.y, ww.z, ww.w}, qf[4] = {qq.x, qq.y, qq.z, qq.w};
                const float kf[4] = {kk.x, kk.y, kk.z, kk.w}, af[4] = {aa.x, aa.y, aa.z, aa.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < RL; ++r) {
                        float x = fmaf(S[r][j4 + e], wf[e], fmaf(sav[r], af[e], kf[e] * vv[r]));
                        S[r][j4 + e] = x;
                        yv[r] = fmaf(x, qf[e], yv[r]);
                    }
            }
#pragma unroll
            for (int r = 0; r < RL; ++r) yv[r] = group_sum<Log2<NCG>::v>(yv[r]);

            if (cg == 0) {
                const size_t o = head_base + (size_t)t * tstride + row0;
                *reinterpret_cast<float4*>(p.sa + o) = make_float4(sav[0], sav[1], sav[2], sav[3]);
                *reinterpret_cast<uint2*>(p.y + o) = make_uint2(pack_bf16x2(yv[0], yv[1]), pack_bf16x2(yv[2], yv[3]));
            }
            if ((t + 1) % CHUNK == 0) {
                // s[b,h,c,j,i] = S[i][j]
                float* dst = p.s + (((size_t)blockIdx.x * (T / CHUNK) + t / CHUNK) * N) * N + row0;
#pragma unroll
                for (int j = 0; j < JL; ++j)
                    *reinterpret_cast<float4*>(dst + (size_t)(col0 + j) * N) = make_float4(S[0][j], S[1][j], S[2][j], S[3][j]);
            }
        }
        if (sc + 1 < nsc) park(buf ^ 1);
        block_sync();
    }
}

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
