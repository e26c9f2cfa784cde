// Under the power cap, does the 16x16x32 f16 MFMA deliver more FLOP/s than the 32x32x16 one (same FLOPs per clock on paper)?
// (MI355X_MICROARCH.md, "Shape": 1.15 x in bare bf16 loops on random data.)  Bare loops, random operands in registers, WAVES waves per SIMD,
// every wave 4 independent accumulator tiles of 32 x 32 (= 16 tiles of 16 x 16), the same FLOPs per iteration in both kernels;
// ~3 s of back-to-back launches each, wall time per launch.   hipcc --offload-arch=gfx950 -O3 -o mfma_shape scripts/micro/mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(const half8* in, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    half8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = in[(t * 4 + i) & 65535]; b[i] = in[(t * 4 + 2 + i) & 65535]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)                // K = 32 per tile and rep: two K = 16 MFMAs
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[(s + i) & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}
__global__ __launch_bounds__(256) void k16(const half8* in, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    half8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = in[(t * 4 + i) & 65535]; b[i] = in[(t * 4 + 2 + i) & 65535]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i)               // K = 32 per 16 x 16 tile and rep: one MFMA
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[t] = s;
}
template <typename K> double run(K kern, const half8* in, float* out, int blocks, int iters, double seconds) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        (void)hipDeviceSynchronize();
        n += 50;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return el / n;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 1, iters = 400;
    half8* in; float* out;
    (void)hipMalloc(&in, 65536 * sizeof(half8)); (void)hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    _Float16* h = (_Float16*)malloc(65536 * 16);
    srand(1);
    for (int i = 0; i < 65536 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    (void)hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    const int blocks = 256 * waves;
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*rep*/ * 4 /*tiles*/ * 2 /*K steps*/ * 2.0 * 32 * 32 * 16;
    for (int round = 0; round < 2; ++round) {
        const double t32 = run(k32, in, out, blocks, iters, 3.0), t16 = run(k16, in, out, blocks, iters, 3.0);
        printf("%d wave(s) per SIMD, random operands: 32x32x16 %.1f us per launch = %.0f TFLOP/s   16x16x32 %.1f us = %.0f TFLOP/s   ratio %.3f\n", waves,
               t32 * 1e6, flop / t32 / 1e12, t16 * 1e6, flop / t16 / 1e12, t32 / t16);
    }
    return 0;
}
