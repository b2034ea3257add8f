// Host-core implementation of the WKV7 operator: what the `CPU` dispatch key of torch.ops.wind_backstepping runs
// (BASELINE config 1, "RWKV_FLOAT_MODE=fp32 CPU WKV path"; the reference registers the CUDA key only,
// VisualRWKV-v7/v7.00/cuda/wkv7_op.cpp:26, so its model cannot step on a host without a GPU).
//
// Same operator contract as the device kernels (include/visualrwkv_hip.h): activations (B,T,H,64), `s` (B,H,T/16,64,64)
// holding S^T at the end of every 16-token chunk, `sa` (B,T,H,64) -- the tensors WindBackstepping saves
// (src/model.py:52-56).  Not the device algorithm and not the reference's either: one task per (b, head) on a pool of
// host threads, the 64x64 state as 64 rows of 64 contiguous floats (a row is 4 AVX-512 / 8 AVX2 vectors, every inner
// loop runs over the key index j and vectorises).  The backward never divides by the decay (the reference un-steps
// the state with 1/w, cuda/wkv7_cuda.cu:91-95): it re-walks each chunk forward from the checkpoint of the chunk
// before, keeps the 16 intermediate states (256 KB, L2-resident) and then walks the chunk backwards through them.
#include <stdint.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <pthread.h>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/visualrwkv_hip.h"

namespace {

constexpr int N = VRWKV_HEAD_SIZE;
constexpr int L = VRWKV_CHUNK_LEN;

inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline uint16_t f2bf(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float ld(const uint16_t* p) { return bf2f(*p); }
inline float ld(const float* p) { return *p; }
inline void st(uint16_t* p, float x) { *p = f2bf(x); }
inline void st(float* p, float x) { *p = x; }

// A persistent pool of host threads (created on first use, reused by every call: the op runs once per layer and direction) that
// hands out task indices through one atomic counter.  Nothing throws across the C boundary: a pool that cannot be created runs the
// tasks on the calling thread.
// fork(): only the forking thread exists in the child, so a pool inherited from the parent lists workers that are not there and a
// call would wait for them forever.  All state lives in a heap object tagged with the pid that created it; a call from another
// pid abandons that object (no joins, no destructors: its threads and possibly-held locks belong to the parent) and starts a new one.
class Pool {
    struct State {
        std::mutex call_mu, mu;
        std::condition_variable cv, done;
        std::vector<std::thread> workers;
        std::function<void(int)>* fn = nullptr;
        std::atomic<int> next{0};
        int n_tasks = 0, active = 0, wanted = 0;
        unsigned long epoch = 0;
        bool stop = false;
        long pid = 0;                                              // the process that created this state (published with it: one atomic pointer)
    };
public:
    static Pool& get() { static Pool p; return p; }
    // run body(0 .. n_tasks-1) on up to n_threads threads (0: hardware_concurrency), the caller included
    template <class F>
    void run(int n_tasks, int n_threads, F&& body) {
        if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads > n_tasks) n_threads = n_tasks;
        if (n_threads <= 1) { for (int i = 0; i < n_tasks; ++i) body(i); return; }
        State* s = state();
        if (!s) { for (int i = 0; i < n_tasks; ++i) body(i); return; }
        std::lock_guard<std::mutex> call(s->call_mu);               // one parallel region at a time
        grow(s, n_threads - 1);
        const int helpers = (int)std::min<size_t>(s->workers.size(), (size_t)(n_threads - 1));
        std::function<void(int)> fn = body;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->fn = &fn; s->n_tasks = n_tasks; s->next.store(0); s->active = helpers; s->wanted = helpers; ++s->epoch;
        }
        s->cv.notify_all();
        for (int i = s->next.fetch_add(1); i < n_tasks; i = s->next.fetch_add(1)) body(i);
        std::unique_lock<std::mutex> lk(s->mu);
        s->done.wait(lk, [&] { return s->active == 0; });
        s->fn = nullptr;
    }
private:
    Pool() { pthread_atfork(nullptr, nullptr, &Pool::after_fork_in_child); }
    // the child of a fork() has one thread: whoever held the swap flag in the parent does not exist here, so release it (the state itself is
    // recognised as the parent's by its pid and replaced on first use)
    static void after_fork_in_child() { Pool::get().swap_.store(false, std::memory_order_release); }
    ~Pool() {
        State* s = st_.load();
        if (!s || s->pid != (long)getpid()) return;                // a forked child never joins the parent's threads
        { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; ++s->epoch; }
        s->cv.notify_all();
        for (auto& t : s->workers) if (t.joinable()) t.join();
        delete s;
    }
    // the state of THIS process (nullptr: allocation failed, run serially)
    State* state() {
        const long me = (long)getpid();
        State* s = st_.load(std::memory_order_acquire);
        if (s && s->pid == me) return s;                            // pid and state travel together: no window in which a reader pairs one with the other's
        while (swap_.exchange(true, std::memory_order_acquire)) std::this_thread::yield();      // not a std::mutex: one held across a fork stays held in the child (the flag is reset by the atfork handler)
        s = st_.load(std::memory_order_acquire);
        if (!s || s->pid != me) {
            State* fresh = new (std::nothrow) State;               // the old one (if any) is the parent's: abandoned, not destroyed
            if (fresh) fresh->pid = me;
            st_.store(fresh, std::memory_order_release);
            s = fresh;
        }
        swap_.store(false, std::memory_order_release);
        return s;
    }
    static void grow(State* s, int n) {
        while ((int)s->workers.size() < n) {
            try {
                const int id = (int)s->workers.size();
                s->workers.emplace_back([s, id] { loop(s, id); });
            } catch (...) { break; }                              // out of threads: run with what exists
        }
    }
    static void loop(State* s, int id) {
        unsigned long seen = 0;
        for (;;) {
            std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(s->mu);
                s->cv.wait(lk, [&] { return s->epoch != seen; });
                seen = s->epoch;
                if (s->stop) return;
                if (id >= s->wanted) continue;                    // more workers exist than this call asked for
                fn = s->fn; n = s->n_tasks;
            }
            for (int i = s->next.fetch_add(1); i < n; i = s->next.fetch_add(1)) (*fn)(i);
            std::lock_guard<std::mutex> lk(s->mu);
            if (--s->active == 0) s->done.notify_one();
        }
    }
    std::atomic<State*> st_{nullptr};
    std::atomic<bool> swap_{false};
};

template <class F>
void for_each_head(int n_tasks, int n_threads, F&& body) { Pool::get().run(n_tasks, n_threads, body); }

// One token of the recurrence on S[i][j] (i = value row, j = key column), src/model.py's op contract:
//   sa_i = sum_j z_j S_ij ;  S_ij <- S_ij w_j + sa_i a_j + v_i k_j ;  y_i = sum_j S_ij q_j
struct Tok { float w[N], q[N], k[N], v[N], z[N], a[N]; };

template <class T>
inline void load_tok(Tok& x, const T* w, const T* q, const T* k, const T* v, const T* z, const T* a, size_t base) {
    for (int j = 0; j < N; ++j) {
        x.w[j] = expf(-expf(ld(w + base + j)));
        x.q[j] = ld(q + base + j); x.k[j] = ld(k + base + j); x.v[j] = ld(v + base + j);
        x.z[j] = ld(z + base + j); x.a[j] = ld(a + base + j);
    }
}

inline void step(float (*S)[N], const Tok& x, float* sa_out, float* y_out) {
    for (int i = 0; i < N; ++i) {
        float* row = S[i];
        float sa = 0.f;
        for (int j = 0; j < N; ++j) sa += x.z[j] * row[j];
        const float vi = x.v[i];
        float y = 0.f;
        for (int j = 0; j < N; ++j) {
            const float s = row[j] * x.w[j] + sa * x.a[j] + vi * x.k[j];
            row[j] = s;
            y += s * x.q[j];
        }
        sa_out[i] = sa;
        if (y_out) y_out[i] = y;
    }
}

template <class T>
void forward_head(int T_, int H, int b, int h, const T* w, const T* q, const T* k, const T* v, const T* z, const T* a,
                  T* y, float* s, float* sa) {
    alignas(64) float S[N][N];
    memset(S, 0, sizeof(S));
    Tok x;
    float yrow[N];
    for (int t = 0; t < T_; ++t) {
        const size_t base = (((size_t)b * T_ + t) * H + h) * N;
        load_tok(x, w, q, k, v, z, a, base);
        step(S, x, sa + base, yrow);
        for (int i = 0; i < N; ++i) st(y + base + i, yrow[i]);
        if ((t + 1) % L == 0) {
            float* ck = s + (((size_t)b * H + h) * (T_ / L) + t / L) * N * N;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) ck[(size_t)j * N + i] = S[i][j];       // checkpoint holds S^T
        }
    }
}

template <class T>
void backward_head(int T_, int H, int b, int h, const T* w, const T* q, const T* k, const T* v, const T* z, const T* a,
                   const T* dy, const float* s, const float* sa, T* dw, T* dq, T* dk, T* dv, T* dz, T* da) {
    // hist[t] = state BEFORE token t of the current chunk, hist[L] = state after its last token
    std::vector<float> hist_buf((size_t)(L + 1) * N * N), ds_buf((size_t)N * N, 0.f);
    auto hist = reinterpret_cast<float (*)[N][N]>(hist_buf.data());
    auto dS = reinterpret_cast<float (*)[N]>(ds_buf.data());
    std::vector<Tok> toks(L);
    float scratch[N], dsa[N], gw[N], gk[N], gz[N], ga[N], gq[N];
    for (int c = T_ / L - 1; c >= 0; --c) {
        if (c == 0) memset(hist[0], 0, sizeof(float) * N * N);
        else {
            const float* ck = s + (((size_t)b * H + h) * (T_ / L) + (c - 1)) * N * N;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) hist[0][i][j] = ck[(size_t)j * N + i];
        }
        for (int u = 0; u < L; ++u) {
            const size_t base = (((size_t)b * T_ + c * L + u) * H + h) * N;
            load_tok(toks[u], w, q, k, v, z, a, base);
            memcpy(hist[u + 1], hist[u], sizeof(float) * N * N);
            step(hist[u + 1], toks[u], scratch, nullptr);
        }
        for (int u = L - 1; u >= 0; --u) {
            const size_t base = (((size_t)b * T_ + c * L + u) * H + h) * N;
            const Tok& x = toks[u];
            const float (*Sn)[N] = hist[u + 1];       // after token u
            const float (*Sp)[N] = hist[u];           // before token u
            for (int j = 0; j < N; ++j) gw[j] = gk[j] = gz[j] = ga[j] = gq[j] = 0.f;
            for (int i = 0; i < N; ++i) {
                const float dyi = ld(dy + base + i), vi = x.v[i], sai = sa[base + i];
                float* d = dS[i];
                float gv = 0.f, gsa = 0.f;
                for (int j = 0; j < N; ++j) {
                    gq[j] += Sn[i][j] * dyi;
                    const float g = d[j] + dyi * x.q[j];       // dL/dS after token u
                    d[j] = g;
                    gw[j] += g * Sp[i][j];
                    gk[j] += g * vi;
                    ga[j] += g * sai;
                    gv += g * x.k[j];
                    gsa += g * x.a[j];
                }
                st(dv + base + i, gv);
                dsa[i] = gsa;
            }
            for (int i = 0; i < N; ++i) {
                float* d = dS[i];
                const float g = dsa[i];
                for (int j = 0; j < N; ++j) {
                    gz[j] += g * Sp[i][j];
                    d[j] = d[j] * x.w[j] + g * x.z[j];         // dL/dS before token u
                }
            }
            for (int j = 0; j < N; ++j) {
                const float e = -expf(ld(w + base + j));       // d w / d w_raw = w * (-exp(w_raw))
                st(dw + base + j, gw[j] * x.w[j] * e);
                st(dq + base + j, gq[j]); st(dk + base + j, gk[j]); st(dz + base + j, gz[j]); st(da + base + j, ga[j]);
            }
        }
    }
}

int check(int B, int T, int H, int dtype) {
    if (B <= 0 || T <= 0 || H <= 0 || (dtype != 0 && dtype != 1)) return VRWKV_EINVAL;
    if (T % L != 0) return VRWKV_ESHAPE;
    return VRWKV_OK;
}

}  // namespace

extern "C" {

int vrwkv_wkv7_forward_host(int B, int T, int H, int dtype, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, void* y, float* s, float* sa, int n_threads) {
    int rc = check(B, T, H, dtype);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !y || !s || !sa) return VRWKV_EINVAL;
    for_each_head(B * H, n_threads, [&](int i) {
        if (dtype == 0)
            forward_head<uint16_t>(T, H, i / H, i % H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                                   (const uint16_t*)z, (const uint16_t*)a, (uint16_t*)y, s, sa);
        else
            forward_head<float>(T, H, i / H, i % H, (const float*)w, (const float*)q, (const float*)k, (const float*)v,
                                (const float*)z, (const float*)a, (float*)y, s, sa);
    });
    return VRWKV_OK;
}

int vrwkv_wkv7_backward_host(int B, int T, int H, int dtype, const void* w, const void* q, const void* k, const void* v,
                             const void* z, const void* a, const void* dy, const float* s, const float* sa,
                             void* dw, void* dq, void* dk, void* dv, void* dz, void* da, int n_threads) {
    int rc = check(B, T, H, dtype);
    if (rc) return rc;
    if (!w || !q || !k || !v || !z || !a || !dy || !s || !sa || !dw || !dq || !dk || !dv || !dz || !da) return VRWKV_EINVAL;
    for_each_head(B * H, n_threads, [&](int i) {
        if (dtype == 0)
            backward_head<uint16_t>(T, H, i / H, i % H, (const uint16_t*)w, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,
                                    (const uint16_t*)z, (const uint16_t*)a, (const uint16_t*)dy, s, sa, (uint16_t*)dw, (uint16_t*)dq,
                                    (uint16_t*)dk, (uint16_t*)dv, (uint16_t*)dz, (uint16_t*)da);
        else
            backward_head<float>(T, H, i / H, i % H, (const float*)w, (const float*)q, (const float*)k, (const float*)v,
                                 (const float*)z, (const float*)a, (const float*)dy, s, sa, (float*)dw, (float*)dq, (float*)dk,
                                 (float*)dv, (float*)dz, (float*)da);
    });
    return VRWKV_OK;
}

}  // extern "C"
