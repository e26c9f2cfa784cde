// Microbenchmark: issue rate of v_mfma_f32_32x32x16_{f16,bf16} and v_mfma_f32_32x32x2_f32 on gfx950, one or two waves per SIMD,
// one accumulation chain or two alternating ones.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>   // 5: rotating A (VGPR), 6: rotating A (AGPR), 7: B from LDS, 8: rotating AGPR A + LDS B, 3+3;  0: f16 one chain, 1: f16 two alternating chains, 2: bf16 one chain, 3: f16 chains of 3 alternating, 4: f32 32x32x2
__global__ void k(const float* in, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a, b;
    bf8 ab, bb;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)in[lane * 8 + j];
        b[j] = (_Float16)in[512 + lane * 8 + j];
        ab[j] = (__bf16)in[lane * 8 + j];
        bb[j] = (__bf16)in[512 + lane * 8 + j];
    }
    __shared__ half8 lb[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) lb[i] = b;
    half8 wa[16];
    for (int i = 0; i < 16; ++i) {
        wa[i] = a;
        wa[i][i & 7] = (_Float16)(0.01f * i);
        if (MODE == 6 || MODE == 8) asm volatile("" : "+a"(wa[i]));
    }
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    const float fa = in[lane], fb = in[64 + lane];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            } else if (MODE == 1) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            } else if (MODE == 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            } else if (MODE == 3) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            } else if (MODE == 5 || MODE == 6) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 0], b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 1], b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 2], b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 3], b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 0], b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 1], b, c0, 0, 0, 0);
            } else if (MODE == 7) {
                const half8 b0 = lb[(u * 2 + 0) * 64 + lane], b1 = lb[(u * 2 + 1) * 64 + lane];
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, c0, 0, 0, 0);
            } else if (MODE == 8) {
                const half8 b0 = lb[(u * 2 + 0) * 64 + lane], b1 = lb[(u * 2 + 1) * 64 + lane];
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 1], b0, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 0], b1, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 0], b0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 3], b0, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 2], b1, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u * 4 + 2], b0, c1, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, const float* in, float* out, long long* cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    const double n_mfma = (double)iters * 24;                  // per wave
    const double waves_per_simd = threads / 256.0;
    printf("%-34s threads %3d: %.3f ms, %.1f ns per MFMA per SIMD, counter ticks per MFMA per wave %.1f\n", name, threads, ms,
           ms * 1e6 / (n_mfma * waves_per_simd), avg / n_mfma);
}

int main() {
    float *in, *out; long long* cyc;
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        run<0>("f16 32x32x16, one chain", threads, in, out, cyc, iters);
        run<1>("f16 32x32x16, two alternating", threads, in, out, cyc, iters);
        run<3>("f16 32x32x16, 3+3", threads, in, out, cyc, iters);
        run<2>("bf16 32x32x16, one chain", threads, in, out, cyc, iters);
        run<4>("f32 32x32x2, one chain", threads, in, out, cyc, iters);
        run<5>("f16, rotating A in VGPRs", threads, in, out, cyc, iters);
        run<6>("f16, rotating A in AGPRs", threads, in, out, cyc, iters);
        run<7>("f16, B from LDS", threads, in, out, cyc, iters);
        run<8>("f16, AGPR A + LDS B, 3+3", threads, in, out, cyc, iters);
    }
    return 0;
}
