// VALU issue-rate probe for gfx950: cycles per wave-instruction and SIMD for the instructions the WKV7 kernels' hi/lo operand
// split is made of, at 1..3 waves per SIMD (the WKV7 backward runs 3).  Standalone:
//   hipcc --offload-arch=gfx950 -O3 benchmarks/valu_rate.hip -o benchmarks/_alt/valu_rate && benchmarks/_alt/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(768) rate_kernel(int iters, const float* in, float* out, long long* cycles) {
    float a[16], b[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[(threadIdx.x + i) & 1023]; b[i] = in[(threadIdx.x + 2 * i + 5) & 1023]; u[i] = __float_as_uint(a[i]); }
    const unsigned km1 = 0x0000BF80u;       // packed bf16 (-1, 0)
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define SUB(i)   asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define DOT2(i)  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "s"(km1));
#define DOT2C(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define CVT(i)   asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define SHL(i)   asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define AND(i)   asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define EXP(i)   asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define FMA(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
#define MOV(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
#define PERM(i)  asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]), "s"(0x07060302u));
#define MULDPP(i) asm volatile("v_mul_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]));
#define SPLIT6(i) asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_lshlrev_b32 %1, 16, %0\n\tv_sub_f32 %2, %2, %1\n\tv_and_b32 %1, 0xffff0000, %0\n\tv_sub_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %1, %2, %3" : "=&v"(u[i]), "=&v"(u[(i + 8) & 15]), "+v"(a[i]), "+v"(b[i]));
#define SPLIT4(i) asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_dot2_f32_bf16 %2, %0, %4, %2\n\tv_dot2_f32_bf16 %3, %0, %5, %3\n\tv_cvt_pk_bf16_f32 %1, %2, %3" : "=&v"(u[i]), "=&v"(u[(i + 8) & 15]), "+v"(a[i]), "+v"(b[i]) : "s"(km1), "s"(0xBF800000u));
        if (OP == 0) { REP16(SUB) REP16(SUB) }
        if (OP == 1) { REP16(DOT2) REP16(DOT2) }
        if (OP == 2) { REP16(DOT2C) REP16(DOT2C) }
        if (OP == 3) { REP16(CVT) REP16(CVT) }
        if (OP == 4) { REP16(SHL) REP16(SHL) }
        if (OP == 5) { REP16(AND) REP16(AND) }
        if (OP == 6) { REP16(EXP) REP16(EXP) }
        if (OP == 7) { REP16(FMA) REP16(FMA) }
        if (OP == 8) { REP16(MOV) REP16(MOV) }
        if (OP == 9) { REP16(PERM) REP16(PERM) }
        if (OP == 10) { REP16(MULDPP) REP16(MULDPP) }
        if (OP == 11) { SPLIT6(0) SPLIT6(1) SPLIT6(2) SPLIT6(3) SPLIT6(4) SPLIT6(5) SPLIT6(6) SPLIT6(7) }
        if (OP == 12) { SPLIT4(0) SPLIT4(1) SPLIT4(2) SPLIT4(3) SPLIT4(4) SPLIT4(5) SPLIT4(6) SPLIT4(7) }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i] + __uint_as_float(u[i] & 0x3fffffffu);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    const int iters = 2000, grid = 256;
    float *in, *out; long long* cyc;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, grid * 768 * 4); hipMalloc(&cyc, grid * 12 * 8);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 0.37f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[] = {"v_sub_f32", "v_dot2_f32_bf16", "v_dot2c_f32_bf16", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_and_b32 (literal)", "v_exp_f32",
                           "v_fma_f32", "v_mov_b32", "v_perm_b32", "v_mul_f32_dpp", "split: cvt shl sub and sub cvt (6)", "split: cvt dot2 dot2 cvt (4)"};
    const int per_iter[] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 48, 32};
    void (*k[])(int, const float*, float*, long long*) = {rate_kernel<0>, rate_kernel<1>, rate_kernel<2>, rate_kernel<3>, rate_kernel<4>, rate_kernel<5>, rate_kernel<6>,
                                                        rate_kernel<7>, rate_kernel<8>, rate_kernel<9>, rate_kernel<10>, rate_kernel<11>, rate_kernel<12>};
    printf("{\"unit\": \"shader cycles per wave-instruction and SIMD\", \"rows\": [\n");
    for (int op = 0; op < 13; ++op) {
        printf(" {\"op\": \"%s\"", names[op]);
        for (int wps = 1; wps <= 3; ++wps) {
            const int threads = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k[op], dim3(grid), dim3(threads), 0, 0, iters, in, out, cyc);
            hipDeviceSynchronize();
            std::vector<long long> c(grid * 4 * wps);
            hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto x : c) s += (double)x;
            const double per = s / c.size() / ((double)iters * per_iter[op] * wps);
            printf(", \"waves_per_simd_%d\": %.2f", wps, per);
        }
        printf("}%s\n", op == 12 ? "" : ",");
    }
    printf("]}\n");
    return 0;
}
