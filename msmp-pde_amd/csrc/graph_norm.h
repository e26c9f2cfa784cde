// Per-graph column statistics shared by the InstanceNorm kernels (aux_kernels.hip) and their backward (train_kernels.hip).
// One 256-thread workgroup per graph: thread = (16-B channel group cg = tid & 31, row slice rs = tid >> 5).
#pragma once
#include "msmp_common.h"

namespace msmp {

__device__ __forceinline__ f32x4 block_colsum(f32x4 v, f32x4* red, int cg, int rs) {
    red[rs * 32 + cg] = v;
    __syncthreads();
    f32x4 s = red[cg];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += red[i * 32 + cg];
    __syncthreads();
    return s;
}

__device__ __forceinline__ void graph_stats(const float* __restrict__ x, int n0, int n1, int cg, int rs, f32x4* red,
                                            float eps, f32x4& mean, f32x4& rstd) {
    const f32x4* xp = reinterpret_cast<const f32x4*>(x);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = n0 + rs; r < n1; r += 8) s += xp[(size_t)r * (H / 4) + cg];
    const float inv = 1.0f / (float)max(n1 - n0, 1);
    mean = block_colsum(s, red, cg, rs) * inv;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (int r = n0 + rs; r < n1; r += 8) {
        const f32x4 d = xp[(size_t)r * (H / 4) + cg] - mean;
        q += d * d;
    }
    const f32x4 var = block_colsum(q, red, cg, rs) * inv;
#pragma unroll
    for (int m = 0; m < 4; ++m) rstd[m] = 1.0f / sqrtf(var[m] + eps);
}

}  // namespace msmp
