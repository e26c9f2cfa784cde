// Microbenchmark (round 4): when does a wave's OWN vector work hide behind its MFMAs on gfx950?  One wave per SIMD (256 threads per
// workgroup, one workgroup per CU), 9 600 v_mfma_f32_32x32x16_f16 per wave with K vector instructions behind each:
//   acc file   : the accumulators (srcC / vDst) in AGPRs ("a") or in architectural VGPRs ("v": what -amdgpu-mfma-vgpr-form selects),
//                optionally with the A operand read from AGPRs (the one-wave-per-SIMD LEM kernel keeps its 256 weight registers there);
//   chain      : 3 rotating accumulators (a dependent MFMA is three MFMAs behind its producer), 2 alternating, or 1 (every MFMA
//                depends on the previous one, the vector instructions sit between producer and consumer);
//   vector mix : v_mul_f32 on 16 independent registers, or the LEM state update's shape (mul, exp, fma, rcp, fma: 2 of 5 transcendental).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_acc_file.bin mfma_acc_file.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_A(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[J]) : "v"(a), "v"(b));
#define MF_V(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[J]) : "v"(a), "v"(b));
#define MF_W(J) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[J]) : "a"(a), "v"(b));       // the A operand (stationary weights) from an AGPR
#define VM { const int i = (n++) % 16; asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c)); }
#define VL { const int i = (n++) % 16; asm volatile("v_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1\n\tv_rcp_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c)); }

template <int FILE_, int CHAIN, int MIX, int K>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[(threadIdx.x & 63) + 64 * i] * 0.5f + 1.0f;
    const float c = in[3] * 1e-3f + 0.9999f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(x[i] * 0.01f); b[i] = (_Float16)(x[8 + i] * 0.01f); }
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    int n = 0;
    for (int it = 0; it < iters; ++it) {
#define VEC { if (MIX == 0) { if (K >= 1) VM if (K >= 2) VM if (K >= 3) VM if (K >= 4) VM if (K >= 5) VM if (K >= 6) VM } else { if (K >= 1) VL } }
#define STEP(J) { if (FILE_ == 0) MF_A(J) else if (FILE_ == 1) MF_V(J) else MF_W(J) VEC }
#define STEP3 STEP(0) STEP(CHAIN >= 2 ? 1 : 0) STEP(CHAIN >= 3 ? 2 : 0)
        R16(STEP3)
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + acc[0][i] + acc[1][i] + acc[2][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILE_, int CHAIN, int MIX, int K>
static float run(const float* in, float* out) {
    const int iters = 200;
    hipLaunchKernelGGL((k<FILE_, CHAIN, MIX, K>), dim3(256), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<FILE_, CHAIN, MIX, K>), dim3(256), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 256 * 4);
    std::vector<float> h(4096, 0.7f); hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("one wave per SIMD, 9600 MFMAs per wave; us per launch (MFMAs alone = the first column)\n");
    printf("%-44s %9s %9s %9s %9s\n", "accumulators / chain", "0 vector", "4 v_mul", "6 v_mul", "LEM x5");
#define ROW(NAME, F, C) printf("%-44s %9.1f %9.1f %9.1f %9.1f\n", NAME, run<F, C, 0, 0>(in, out), run<F, C, 0, 4>(in, out), run<F, C, 0, 6>(in, out), run<F, C, 1, 1>(in, out));
    ROW("AGPR accumulators, 3 rotating", 0, 3)
    ROW("AGPR accumulators, 2 alternating", 0, 2)
    ROW("AGPR accumulators, 1 (dependent chain)", 0, 1)
    ROW("VGPR accumulators, 3 rotating", 1, 3)
    ROW("VGPR accumulators, 2 alternating", 1, 2)
    ROW("VGPR accumulators, 1 (dependent chain)", 1, 1)
    ROW("VGPR acc., A operand in AGPRs, 2 alternating", 2, 2)
    ROW("VGPR acc., A operand in AGPRs, 1 (chain)", 2, 1)
    return 0;
}
