// Microbenchmark: how fast can ONE wave issue vector instructions on gfx950?  Straight-line blocks of 256 instructions (no loop
// overhead to speak of), N independent chains (1, 2, 4, 8, 16), 1 / 2 / 4 waves per SIMD.  Reports cycles per instruction PER WAVE
// (s_memtime deltas of the wave itself) and per SIMD.   Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue.bin valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R256(X) R16(R16(X))

template <int OP, int CH>   // OP 0: v_mul_f32, 1: v_fma_f32, 2: v_exp_f32, 3: v_pk_mul_f32;  CH: independent chains
__global__ void k(const float* in, float* out, long long* cyc, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[threadIdx.x + 64 * i] * 0.5f + 1.0f;
    const float c = in[3] * 1e-3f + 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{x[2 * i], x[2 * i + 1]};
    const f2 c2 = {c, c};
    int n = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define STEP { const int i = (n++) % CH; \
        if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c)); \
        else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c)); \
        else if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); \
        else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i % 8]) : "v"(c2)); }
        R256(STEP)
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CH>
void run(const char* name, int threads, const float* in, float* out, long long* cyc) {
    const int iters = 20, grid = 256;
    hipLaunchKernelGGL((k<OP, CH>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CH>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double n = 256.0 * iters;
    printf("%-14s chains %2d  waves/SIMD %d: %7.2f counter ticks per instruction per wave   (%.3f ms wall, %.2f ns per instruction per SIMD)\n", name, CH, threads / 256,
           avg / n, ms, ms * 1e6 / (n * threads / 256));
}

int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    std::vector<float> h(4096, 0.7f); hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512, 1024}) {
        run<0, 1>("v_mul_f32", threads, in, out, cyc);
        run<0, 2>("v_mul_f32", threads, in, out, cyc);
        run<0, 4>("v_mul_f32", threads, in, out, cyc);
        run<0, 16>("v_mul_f32", threads, in, out, cyc);
        run<1, 16>("v_fma_f32", threads, in, out, cyc);
        run<2, 1>("v_exp_f32", threads, in, out, cyc);
        run<2, 16>("v_exp_f32", threads, in, out, cyc);
        run<3, 1>("v_pk_mul_f32", threads, in, out, cyc);
        run<3, 8>("v_pk_mul_f32", threads, in, out, cyc);
    }
    return 0;
}
