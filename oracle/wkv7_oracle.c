/* CPU oracle for the WKV7 operator -- plain C restatement.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows the reference CUDA kernels token by token and thread by thread
 * (VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu): forward_kernel :10-52, backward_kernel :54-130.
 * "thread i" of the reference becomes the loop variable i; the per-thread register arrays
 * state[C] / stateT[C] / dstate[C] / dstateT[C] become rows of C x C matrices; the j-loops keep the
 * reference's sequential summation order.  bf16 in/out with round-to-nearest-even (to_bf, :6),
 * fp32 arithmetic, expf instead of __expf.  (b,h) pairs are independent and are spread over
 * OpenMP threads.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product never does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define C 64
#define CHUNK 16

static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int wkv7_oracle_forward(int B, int T, int H, const uint16_t* w_, const uint16_t* q_, const uint16_t* k_,
                        const uint16_t* v_, const uint16_t* a_, const uint16_t* b_,
                        uint16_t* y_, float* s_, float* sa_) {
    if (T % CHUNK) return -2;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int bb = 0; bb < B; ++bb)
    for (int hh = 0; hh < H; ++hh) {
        float (*state)[C] = calloc(C, sizeof(*state));            /* state[i][j], :14 */
        float q[C], k[C], w[C], a[C], b[C];
        for (int t = 0; t < T; ++t) {
            const size_t base = (size_t)bb * T * H * C + (size_t)t * H * C + (size_t)hh * C;
            for (int i = 0; i < C; ++i) {                          /* :19-24 */
                q[i] = bf2f(q_[base + i]);
                w[i] = expf(-expf(bf2f(w_[base + i])));
                k[i] = bf2f(k_[base + i]);
                a[i] = bf2f(a_[base + i]);
                b[i] = bf2f(b_[base + i]);
            }
            for (int i = 0; i < C; ++i) {
                float sa = 0;                                      /* :27-31 */
                for (int j = 0; j < C; ++j) sa += a[j] * state[i][j];
                sa_[base + i] = sa;
                const float v = bf2f(v_[base + i]);
                float y = 0;                                       /* :34-41 */
                for (int j = 0; j < C; ++j) {
                    float s = state[i][j];
                    s = s * w[j] + sa * b[j] + k[j] * v;
                    state[i][j] = s;
                    y += s * q[j];
                }
                y_[base + i] = f2bf(y);
            }
            if ((t + 1) % CHUNK == 0) {                            /* :44-50, stored transposed */
                const size_t sb = ((size_t)(bb * H + hh) * (T / CHUNK) + t / CHUNK) * C * C;
                for (int i = 0; i < C; ++i)
                    for (int j = 0; j < C; ++j) s_[sb + (size_t)j * C + i] = state[i][j];
            }
        }
        free(state);
    }
    return 0;
}

int wkv7_oracle_backward(int B, int T, int H, const uint16_t* w_, const uint16_t* q_, const uint16_t* k_,
                         const uint16_t* v_, const uint16_t* a_, const uint16_t* b_, const uint16_t* dy_,
                         const float* s_, const float* sa_,
                         uint16_t* dw_, uint16_t* dq_, uint16_t* dk_, uint16_t* dv_, uint16_t* da_, uint16_t* db_) {
    if (T % CHUNK) return -2;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int bb = 0; bb < B; ++bb)
    for (int hh = 0; hh < H; ++hh) {
        float (*stateT)[C] = calloc(C, sizeof(*stateT));           /* per thread i: stateT[i][j] = S[j][i] */
        float (*dstate)[C] = calloc(C, sizeof(*dstate));           /* dstate[i][j]  = dS[i][j]             */
        float (*dstateT)[C] = calloc(C, sizeof(*dstateT));         /* dstateT[i][j] = dS[j][i]             */
        float w[C], q[C], k[C], v[C], a[C], b[C], dy[C], sa[C], dSb_shared[C], wfac[C];
        for (int t = T - 1; t >= 0; --t) {
            const size_t base = (size_t)bb * T * H * C + (size_t)t * H * C + (size_t)hh * C;
            for (int i = 0; i < C; ++i) {                          /* :63-74 */
                q[i] = bf2f(q_[base + i]);
                wfac[i] = -expf(bf2f(w_[base + i]));
                w[i] = expf(wfac[i]);
                k[i] = bf2f(k_[base + i]);
                a[i] = bf2f(a_[base + i]);
                b[i] = bf2f(b_[base + i]);
                v[i] = bf2f(v_[base + i]);
                dy[i] = bf2f(dy_[base + i]);
                sa[i] = sa_[base + i];
            }
            if ((t + 1) % CHUNK == 0) {                            /* :76-82 */
                const size_t sb = ((size_t)(bb * H + hh) * (T / CHUNK) + t / CHUNK) * C * C;
                for (int i = 0; i < C; ++i)
                    for (int j = 0; j < C; ++j) stateT[i][j] = s_[sb + (size_t)i * C + j];
            }
            for (int i = 0; i < C; ++i) {
                float dq = 0;                                      /* :84-89 */
                for (int j = 0; j < C; ++j) dq += stateT[i][j] * dy[j];
                dq_[base + i] = f2bf(dq);
                const float iwi = 1.0f / w[i];                     /* :91-97 */
                for (int j = 0; j < C; ++j) {
                    stateT[i][j] = (stateT[i][j] - k[i] * v[j] - b[i] * sa[j]) * iwi;
                    dstate[i][j] += dy[i] * q[j];
                    dstateT[i][j] += q[i] * dy[j];
                }
                float dw = 0, dk = 0, dv = 0, db = 0, dSb = 0;     /* :99-111 */
                for (int j = 0; j < C; ++j) {
                    dw += dstateT[i][j] * stateT[i][j];
                    dk += dstateT[i][j] * v[j];
                    dv += dstate[i][j] * k[j];
                    dSb += dstate[i][j] * b[j];
                    db += dstateT[i][j] * sa[j];
                }
                dw_[base + i] = f2bf(dw * w[i] * wfac[i]);
                dk_[base + i] = f2bf(dk);
                dv_[base + i] = f2bf(dv);
                db_[base + i] = f2bf(db);
                dSb_shared[i] = dSb;                               /* :113-115 */
            }
            for (int i = 0; i < C; ++i) {
                float da = 0;                                      /* :117-122 */
                for (int j = 0; j < C; ++j) da += stateT[i][j] * dSb_shared[j];
                da_[base + i] = f2bf(da);
                for (int j = 0; j < C; ++j) {                      /* :124-128 */
                    dstate[i][j] = dstate[i][j] * w[j] + dSb_shared[i] * a[j];
                    dstateT[i][j] = dstateT[i][j] * w[i] + a[i] * dSb_shared[j];
                }
            }
        }
        free(stateT); free(dstate); free(dstateT);
    }
    return 0;
}
