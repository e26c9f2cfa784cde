// Microbenchmark: what a vector instruction COSTS on gfx950 in the regime of the message kernel (1, 2 or 4 waves per SIMD, all CUs
// busy, optionally one MFMA per 8 vector instructions in the same wave).  For each instruction a loop of 8 independent chains is
// timed; reported: SIMD cycles per wave-instruction (= wall time x clock / instructions per SIMD), clock from s_memrealtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// OP: 0 v_mul_f32, 1 v_pk_mul_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 v_fma_f32, 5 v_pk_fma_f32, 6 v_pk_add_f32, 7 v_cvt_pk_f16_f32,
//     8 v_fma_mixlo_f16, 9 v_cndmask_b32, 10 the swish pipeline in scalar form (per 2 values), 11 the same with packed ops
template <int OP, bool WITH_MFMA>
__global__ void k(const float* in, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float x[8], y[8];
    f32x2 p[8], q[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = in[lane + 64 * i];
        y[i] = in[512 + lane + 64 * i] * 0.999f + 1.0f;
        p[i] = f32x2{x[i], y[i]};
        q[i] = f32x2{y[i], x[i]};
    }
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)in[lane * 8 + j]; b[j] = (_Float16)in[512 + lane * 8 + j]; }
    f32x16 c0;
    for (int r = 0; r < 16; ++r) c0[r] = 0.f;
    const float cst = in[3] * 1e-3f + 0.9999f;
    const f32x2 cst2 = {cst, cst};
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (WITH_MFMA) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        if (OP == 0) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(cst));
            REP8(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cst2));
            REP8(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            REP8(X)
#undef X
        } else if (OP == 3) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(y[i]));
            REP8(X)
#undef X
        } else if (OP == 4) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(cst));
            REP8(X)
#undef X
        } else if (OP == 5) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cst2));
            REP8(X)
#undef X
        } else if (OP == 6) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cst2));
            REP8(X)
#undef X
        } else if (OP == 7) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(cst));
            REP8(X)
#undef X
        } else if (OP == 8) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(y[i]), "v"(cst));
            REP8(X)
#undef X
        } else if (OP == 9) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(cst) : "vcc");
            REP8(X)
#undef X
        } else if (OP == 10) {      // swish of 2 values, scalar ops: add, mul, exp, add, rcp, mul  (x2) = 12 instructions
#define X(i) { float s0, s1, e0, e1;                                                                  \
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x[i]), "v"(cst));                    \
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(y[i]), "v"(cst));                    \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(s0), "v"(cst));                      \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(s1), "v"(cst));                      \
            asm volatile("v_exp_f32 %0, %0" : "+v"(e0));                                              \
            asm volatile("v_exp_f32 %0, %0" : "+v"(e1));                                              \
            asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e0));                                         \
            asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e1));                                         \
            asm volatile("s_nop 0\n\tv_rcp_f32 %0, %0" : "+v"(e0));                                   \
            asm volatile("v_rcp_f32 %0, %0" : "+v"(e1));                                              \
            asm volatile("s_nop 0\n\tv_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(s0), "v"(e0));          \
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y[i]) : "v"(s1), "v"(e1)); }
            REP8(X)
#undef X
        } else if (OP == 11) {      // the same with packed ops: pk_add, pk_mul, exp x2, pk_add, rcp x2, pk_mul = 8 instructions
#define X(i) { f32x2 s, e;                                                                             \
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(p[i]), "v"(cst2));                  \
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(e) : "v"(s), "v"(cst2));                     \
            asm volatile("s_nop 0\n\tv_exp_f32 %0, %0" : "+v"(e[0]));                                  \
            asm volatile("v_exp_f32 %0, %0" : "+v"(e[1]));                                             \
            asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(e) : "v"(f32x2{1.0f, 1.0f}));     \
            asm volatile("s_nop 0\n\tv_rcp_f32 %0, %0" : "+v"(e[0]));                                  \
            asm volatile("v_rcp_f32 %0, %0" : "+v"(e[1]));                                             \
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(s), "v"(e)); }
            REP8(X)
#undef X
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i] + p[i][0] + p[i][1] + q[i][0];
    for (int r = 0; r < 16; ++r) s += c0[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP, bool M>
void run(const char* name, int per_iter, int threads, const float* in, float* out, long long* cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((k<OP, M>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, M>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * grid);
    hipMemcpy(h.data(), cyc, 2 * grid * sizeof(long long), hipMemcpyDeviceToHost);
    double ticks = 0, real = 0; for (int i = 0; i < grid; ++i) { ticks += h[2 * i]; real += h[2 * i + 1]; }
    const double clock_ghz = ticks / (real * 10.0);                 // s_memrealtime ticks at 100 MHz
    const double waves_per_simd = threads / 256.0;
    const double n_inst = (double)iters * per_iter * waves_per_simd;    // per SIMD
    printf("%-40s %s waves/SIMD %.0f: %7.3f ms  clock %.2f GHz  SIMD cycles per wave-instruction %5.2f\n", name, M ? "+1 MFMA/iter" : "            ",
           waves_per_simd, ms, clock_ghz, (ticks / grid) / ((double)iters * per_iter * waves_per_simd));
}

int main() {
    float *in, *out; long long* cyc;
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        run<0, false>("v_mul_f32", 8, threads, in, out, cyc, iters);
        run<1, false>("v_pk_mul_f32", 8, threads, in, out, cyc, iters);
        run<4, false>("v_fma_f32", 8, threads, in, out, cyc, iters);
        run<5, false>("v_pk_fma_f32", 8, threads, in, out, cyc, iters);
        run<6, false>("v_pk_add_f32", 8, threads, in, out, cyc, iters);
        run<2, false>("v_exp_f32", 8, threads, in, out, cyc, iters);
        run<3, false>("v_rcp_f32", 8, threads, in, out, cyc, iters);
        run<7, false>("v_cvt_pk_f16_f32", 8, threads, in, out, cyc, iters);
        run<8, false>("v_fma_mixlo_f16", 8, threads, in, out, cyc, iters);
        run<9, false>("v_cndmask_b32", 8, threads, in, out, cyc, iters);
        run<10, false>("swish x16 values, scalar (96 instr)", 96, threads, in, out, cyc, iters / 4);
        run<11, false>("swish x16 values, packed (64 instr)", 64, threads, in, out, cyc, iters / 4);
        run<0, true>("v_mul_f32", 8, threads, in, out, cyc, iters);
        run<1, true>("v_pk_mul_f32", 8, threads, in, out, cyc, iters);
        run<2, true>("v_exp_f32", 8, threads, in, out, cyc, iters);
        run<10, true>("swish x16 values, scalar (96 instr)", 96, threads, in, out, cyc, iters / 4);
        run<11, true>("swish x16 values, packed (64 instr)", 64, threads, in, out, cyc, iters / 4);
    }
    return 0;
}
