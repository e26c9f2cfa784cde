// Microbenchmark: do matrix (MFMA) and vector instructions of DIFFERENT waves of one SIMD overlap on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 share the four SIMDs pairwise.  Role of a wave: 0 idle, 1 MFMA chain
// (v_mfma_f32_32x32x16_f16, three accumulators), 2 v_mul_f32 (16 chains), 3 v_exp_f32 (16 chains), 4 one MFMA + 6 v_mul alternating.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap.bin mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOPS>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int role_lo, int role_hi, int it_m, int it_v) {
    const int wave = threadIdx.x >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave < 4 ? role_lo : role_hi);
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[(threadIdx.x & 63) + 64 * i] * 0.5f + 1.0f;
    const float c = in[3] * 1e-3f + 0.9999f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(x[i] * 0.01f); b[i] = (_Float16)(x[8 + i] * 0.01f); }
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    int n = 0;
    if (role == 1) {
        for (int it = 0; it < it_m; ++it) {
#define MSTEP { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0); \
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2], 0, 0, 0); asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])); }
            R16(MSTEP)
        }
    } else if (role == 2) {
        for (int it = 0; it < it_v; ++it) {
#define VSTEP { const int i = (n++) % 16; asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c)); }
            R64(VSTEP) R64(VSTEP) R64(VSTEP) R64(VSTEP)
        }
    } else if (role == 3) {
        for (int it = 0; it < it_v; ++it) {
#define ESTEP { const int i = (n++) % 16; asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); }
            R64(ESTEP) R64(ESTEP) R64(ESTEP) R64(ESTEP)
        }
    } else if (role >= 5) {          // MFMA chain with the issue port released between MFMAs: s_nop (role - 5) x 4 cycles after each
        for (int it = 0; it < it_m; ++it) {
#define NSTEP(J) { acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[J], 0, 0, 0); asm volatile("" : "+v"(acc[J])); \
                   if (NOPS >= 1) asm volatile("s_nop 3"); if (NOPS >= 2) asm volatile("s_nop 3"); if (NOPS >= 3) asm volatile("s_nop 3"); if (NOPS >= 4) asm volatile("s_nop 3"); \
                   if (NOPS >= 5) asm volatile("s_nop 3"); if (NOPS >= 6) asm volatile("s_nop 3"); if (NOPS >= 7) asm volatile("s_nop 3"); }
#define NSTEP3 NSTEP(0) NSTEP(1) NSTEP(2)
            R16(NSTEP3)
        }
    } else if (role == 20) {          // one MFMA + three v_exp_f32 alternating (transcendentals behind the wave's own MFMAs)
        for (int it = 0; it < it_m; ++it) {
#define TSTEP(J) { acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[J], 0, 0, 0); asm volatile("" : "+v"(acc[J])); ESTEP ESTEP ESTEP }
#define TSTEP3 TSTEP(0) TSTEP(1) TSTEP(2)
            R16(TSTEP3)
        }
    } else if (role == 21) {          // one MFMA + a dependent chain mul -> exp -> fma -> rcp -> mul (a Swish pair's shape)
        for (int it = 0; it < it_m; ++it) {
#define CSTEP(J) { acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[J], 0, 0, 0); asm volatile("" : "+v"(acc[J])); \
                   asm volatile("v_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %1\n\tv_rcp_f32 %0, %0\n\tv_mul_f32 %0, %0, %1" : "+v"(x[J]) : "v"(c)); }
#define CSTEP3 CSTEP(0) CSTEP(1) CSTEP(2)
            R16(CSTEP3)
        }
    } else if (role == 4) {
        for (int it = 0; it < it_m; ++it) {
#define XSTEP(J) { acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[J], 0, 0, 0); asm volatile("" : "+v"(acc[J])); VSTEP VSTEP VSTEP VSTEP VSTEP VSTEP }
#define XSTEP3 XSTEP(0) XSTEP(1) XSTEP(2)
            R16(XSTEP3)
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + acc[0][i] + acc[1][i] + acc[2][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NOPS = 0>
static float run(const float* in, float* out, int lo, int hi, int it_m, int it_v) {
    hipLaunchKernelGGL(k<NOPS>, dim3(256), dim3(512), 0, 0, in, out, lo, hi, it_m, it_v);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NOPS>, dim3(256), dim3(512), 0, 0, in, out, lo, hi, it_m, it_v);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(4096, 0.7f); hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int it_m = 200;                        // 200 x 48 MFMAs = 9600 per wave
    for (int vrole : {2, 3}) {
        const int it_v = vrole == 2 ? 300 : 100;   // x 256 vector instructions per wave
        printf("vector role %s: %d instructions per wave; matrix role: %d MFMAs per wave\n", vrole == 2 ? "v_mul_f32" : "v_exp_f32", it_v * 256, it_m * 48);
        printf("  MFMA wave alone (one per SIMD)            %8.1f us\n", run(in, out, 1, 0, it_m, it_v));
        printf("  vector wave alone (one per SIMD)          %8.1f us\n", run(in, out, 0, vrole, it_m, it_v));
        printf("  MFMA wave + vector wave on each SIMD      %8.1f us   (max of the two = full overlap, sum = none)\n", run(in, out, 1, vrole, it_m, it_v));
        printf("  two MFMA waves per SIMD                   %8.1f us\n", run(in, out, 1, 1, it_m, it_v));
        printf("  two vector waves per SIMD                 %8.1f us\n", run(in, out, vrole, vrole, it_m, it_v));
    }
#define NROW(N) printf("MFMA wave with s_nop of %2d cycles after each MFMA: alone %8.1f us, with a v_mul_f32 wave (76800) on the SIMD %8.1f us, with a v_exp_f32 wave (25600) %8.1f us\n", \
               4 * N, run<N>(in, out, 5, 0, it_m, 300), run<N>(in, out, 5, 2, it_m, 300), run<N>(in, out, 5, 3, it_m, 100));
    NROW(0) NROW(1) NROW(2) NROW(3) NROW(4) NROW(5) NROW(6) NROW(7)
    printf("one wave per SIMD, 1 MFMA + 3 v_exp_f32 alternating (%d MFMAs, %d v_exp):           %8.1f us   (two waves per SIMD: %8.1f us)\n", it_m * 48, it_m * 48 * 3,
           run(in, out, 20, 0, it_m, 0), run(in, out, 20, 20, it_m, 0));
    printf("one wave per SIMD, 1 MFMA + a dependent mul-exp-fma-rcp-mul chain alternating:       %8.1f us   (two waves per SIMD: %8.1f us)\n",
           run(in, out, 21, 0, it_m, 0), run(in, out, 21, 21, it_m, 0));
    printf("one wave per SIMD, 1 MFMA + 6 v_mul_f32 alternating (%d MFMAs, %d v_mul): %8.1f us\n", it_m * 48, it_m * 48 * 6, run(in, out, 4, 0, it_m, 0));
    printf("two waves per SIMD, the same each:                                        %8.1f us\n", run(in, out, 4, 4, it_m, 0));
    return 0;
}
