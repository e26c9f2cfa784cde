// Does the hardware interlock a VALU write to the SrcA / SrcB registers of an MFMA that has been ISSUED but is still waiting behind
// a dependent chain (same accumulator)?  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_war scripts/micro/mfma_war.hip && /tmp/mfma_war
// Each wave: acc = 0; three MFMAs acc += A*B on the same accumulator, the third with its own B registers (b2); right behind it, GAP
// s_nops and then v_mov writes zeros into b2.  Expected acc = 3 * (A*B) with b2 = b; a race shows as 2 * (A*B) (or a mix).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int GAP, int DEPTH>
__global__ void k(float* out) {
    half8 a, b, b2;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; b2[i] = (_Float16)1.0f; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    (void)b2;
    asm volatile(
        "v_mov_b32 v100, 0x3c003c00\n\tv_mov_b32 v101, 0x3c003c00\n\tv_mov_b32 v102, 0x3c003c00\n\tv_mov_b32 v103, 0x3c003c00\n\t"      // b2 = 1.0 x 8
        "s_nop 7\n\ts_nop 7\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
        ".rept %c3\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
        ".endr\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %1, v[100:103], %0\n\t"
        ".rept %c4\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        : "+v"(acc) : "v"(a), "v"(b), "n"(DEPTH), "n"(GAP) : "v100", "v101", "v102", "v103");
    out[threadIdx.x] = acc[0];
}
template <int GAP, int DEPTH> void run(float* d) {
    k<GAP, DEPTH><<<1, 64>>>(d);
    float h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    float mn = 1e9, mx = -1e9; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("dependent MFMAs ahead %d, s_nops between the MFMA and the overwrite %2d: acc[0] min %.0f max %.0f (expected %d)\n", DEPTH + 1, GAP, mn, mx, 16 * (DEPTH + 2));
}
int main() {
    float* d; hipMalloc(&d, 256);
    run<0, 1>(d); run<1, 1>(d); run<4, 1>(d); run<8, 1>(d); run<16, 1>(d);
    run<0, 3>(d); run<4, 3>(d); run<16, 3>(d); run<32, 3>(d); run<0, 0>(d);
    return 0;
}
