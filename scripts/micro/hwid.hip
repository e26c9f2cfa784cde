// Which hardware slot does a workgroup land in?  512 workgroups of 256 threads with 70 KB of LDS each (two fit a CU) record
// HW_ID (wave slot, SIMD, CU, SE, threadgroup slot) and XCC_ID, then spin ~20 us so that the whole first round is resident at once.
// Build: hipcc --offload-arch=gfx950 -O3 -o hwid hwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    __shared__ float lds[17664];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));          // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));        // HW_REG_XCC_ID
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(10);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (unsigned)lds[threadIdx.x] * 0;
    }
}
int main() {
    const int n = 1024;
    unsigned* d; hipMalloc(&d, n * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 2000);   // 2000 ticks of 100 MHz = 20 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 8);
    hipMemcpy(h.data(), d, n * 8 * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < n; b += (b < 80 ? 1 : 37)) {
        printf("wg %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  [wave %u simd %u pipe %u cu %u sh %u se %u tg %u | xcc %u]", hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15, x & 15);
        }
        printf("\n");
    }
    // histogram of (tg, wave slot of wave 0) over the first 512 and the second 512 workgroups
    for (int half = 0; half < 2; ++half) {
        std::map<unsigned, int> m;
        for (int b = 512 * half; b < 512 * half + 512; ++b) { const unsigned hw = h[(b * 4) * 2]; m[((hw >> 16) & 15) * 16 + (hw & 15)]++; }
        printf("workgroups %d..%d: (tg slot, wave slot) counts:", 512 * half, 512 * half + 511);
        for (auto& kv : m) printf("  (%u,%u): %d", kv.first >> 4, kv.first & 15, kv.second);
        printf("\n");
    }
    return 0;
}
