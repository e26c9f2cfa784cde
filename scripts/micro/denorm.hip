// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  a = 2^-20 (subnormal), b = 1024: product 2^-10 if kept, 0 if flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(float* out, float av, float bv) {
    half8 a, b; f32x16 c;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    a[0] = (_Float16)av; b[0] = (_Float16)bv;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 8); float h[2];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ldexpf(1.f, -20), 1024.f);
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 1024: mfma gives %g (kept: %g), cvt to fp16 kept the value: %g\n", h[0], ldexp(1.0, -10), h[1]);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024.f, ldexpf(1.f, -20));
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("1024 x subnormal B (2^-20): mfma gives %g\n", h[0]);
    return 0;
}
