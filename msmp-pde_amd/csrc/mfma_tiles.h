// Channel-major fp32-MFMA building blocks shared by the MLP kernels and the LEM encoder kernel.
// See the header comment of mlp_kernels.hip for the operand orientation.
#pragma once
#include "msmp_common.h"

namespace msmp {

struct WStage {
    f32x4 r[4];
};

__device__ __forceinline__ void wstage_load(WStage& s, const float* __restrict__ chunk, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.r[i] = *reinterpret_cast<const f32x4*>(chunk + 4 * (tid + 256 * i));
}

__device__ __forceinline__ void wstage_store(const WStage& s, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<f32x4*>(buf + (idx >> 3) * LDW + (idx & 7) * 4) = s.r[i];
    }
}

// acc[T][nb] += W_chunk[32T.., k] * B[k][item]  for the 32 k of one staged chunk.
template <int NB>
__device__ __forceinline__ void mma_chunk(const float* wl, int c, int hh, const f32x4 (&b)[NB][4],
                                          f32x16 (&acc)[4][NB]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 a[4];
#pragma unroll
        for (int T = 0; T < 4; ++T)
            a[T] = *reinterpret_cast<const f32x4*>(wl + (32 * T + c) * LDW + 8 * q + 4 * hh);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[T][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[T][m], b[nb][q][m], acc[T][nb], 0, 0, 0);
    }
}

// Same, with the B operand taken from the accumulators of the previous GEMM (tile t = chunk index):
// register r = 4q + m of that tile is channel 32t + 8q + 4hh + m, the k this chunk's fragment (q, m) covers.
template <int NB>
__device__ __forceinline__ void mma_chunk_from_acc(const float* wl, int c, int hh, const f32x16 (&x)[NB],
                                                   f32x16 (&acc)[4][NB]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 a[4];
#pragma unroll
        for (int T = 0; T < 4; ++T)
            a[T] = *reinterpret_cast<const f32x4*>(wl + (32 * T + c) * LDW + 8 * q + 4 * hh);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[T][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[T][m], x[nb][4 * q + m], acc[T][nb], 0, 0, 0);
    }
}

template <int NB>
__device__ __forceinline__ void acc_init_bias(const float* __restrict__ bias, int hh, f32x16 (&acc)[4][NB]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 32 * T + 8 * q + 4 * hh);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[T][nb][4 * q + m] = bv[m];
        }
}

}  // namespace msmp
