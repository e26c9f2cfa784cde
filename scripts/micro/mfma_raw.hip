// Does a VALU write followed closely by an MFMA that reads the register as SrcA / SrcB need software wait states on gfx950?
// (the compiler's hazard recogniser inserts them for its own instructions, not for inline asm)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_raw scripts/micro/mfma_raw.hip && /tmp/mfma_raw
// v100..v103 = 0; [busy: one MFMA in flight or none]; OP writes 1.0 halves into v100..v103; GAP x s_nop 0; MFMA acc = A * v[100:103].
// Expected acc[0] = 16; a missed dependency shows as 0 .. 14.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int GAP, int OP, int BUSY>
__global__ void k(float* out) {
    half8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    float one = 1.0f, zero = 0.0f;
    asm volatile(
        "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        ".if %c6\n\t"
        "v_mfma_f32_32x32x16_f16 %1, %2, %2, %1\n\t"          // an independent MFMA in flight
        ".endif\n\t"
        ".if %c5 == 0\n\t"
        "v_mov_b32 v100, 0x3c003c00\n\tv_mov_b32 v101, 0x3c003c00\n\tv_mov_b32 v102, 0x3c003c00\n\tv_mov_b32 v103, 0x3c003c00\n\t"
        ".else\n\t"
        "v_fma_mixlo_f16 v100, %3, 1.0, %4\n\tv_fma_mixhi_f16 v100, %3, 1.0, %4\n\t"
        "v_fma_mixlo_f16 v101, %3, 1.0, %4\n\tv_fma_mixhi_f16 v101, %3, 1.0, %4\n\t"
        "v_fma_mixlo_f16 v102, %3, 1.0, %4\n\tv_fma_mixhi_f16 v102, %3, 1.0, %4\n\t"
        "v_fma_mixlo_f16 v103, %3, 1.0, %4\n\tv_fma_mixhi_f16 v103, %3, 1.0, %4\n\t"
        ".endif\n\t"
        ".rept %c7\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %2, v[100:103], %0\n\t"
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
        : "+v"(acc), "+v"(acc2) : "v"(a), "v"(one), "v"(zero), "n"(OP), "n"(BUSY), "n"(GAP) : "v100", "v101", "v102", "v103");
    out[threadIdx.x] = acc[0] + 0.f * acc2[0];
}
template <int GAP, int OP, int BUSY> void run(float* d) {
    k<GAP, OP, BUSY><<<1, 64>>>(d);
    float h[64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    float mn = 1e9, mx = -1e9; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("%s, %s, %2d s_nops between the write and the MFMA: acc[0] min %.0f max %.0f (expected 16)\n", OP ? "v_fma_mixlo/hi_f16" : "v_mov_b32", BUSY ? "another MFMA in flight" : "matrix pipe idle", GAP, mn, mx);
}
int main() {
    float* d; (void)hipMalloc(&d, 256);
    run<0, 0, 0>(d); run<1, 0, 0>(d); run<2, 0, 0>(d); run<4, 0, 0>(d);
    run<0, 1, 0>(d); run<1, 1, 0>(d); run<2, 1, 0>(d); run<4, 1, 0>(d);
    run<0, 0, 1>(d); run<1, 0, 1>(d); run<2, 0, 1>(d); run<4, 0, 1>(d);
    run<0, 1, 1>(d); run<1, 1, 1>(d); run<2, 1, 1>(d); run<4, 1, 1>(d); run<8, 1, 1>(d);
    return 0;
}
