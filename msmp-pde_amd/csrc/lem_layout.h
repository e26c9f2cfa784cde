// Layout of the packed LEM weight blob (msmp_pack_lem_f32), shared by the inference kernels (lem_kernel.hip) and the
// training kernels (lem_train_kernel.hip).
#pragma once
#include "mfma_tiles.h"


namespace msmp {

constexpr int LEM_MAX_INP = 8;
// one [128 out][32 k] chunk as bf16 hi / mid / lo MFMA A fragments, acc order: [s 2][T 4][plane 3][lane 64][8 bf16] = 24 KB
constexpr int LEM_B3_CHUNK_U4 = 2 * 4 * 3 * 64;
constexpr int LEM_B3_CHUNK_FLOATS = LEM_B3_CHUNK_U4 * 4;
// input columns as fp16 "slot" fragments (weight-stationary kernel): [4 groups g2,g3,g1,lin][4 T][2 m][64 lanes][8 halfs]
constexpr int LEM_WXH_FLOATS = 4 * 4 * 2 * 64 * 8 / 2;

// Slot s (0..31) of the K axis of the input MFMAs pairs  A: w_hi[f] | w_hi[f] | w_lo[f]   with   B: x_hi[f] | x_lo[f] | x_hi[f]
// for s in [0,P) | [P,2P) | [2P,3P), then  A: bias_hi | bias_lo  with  B: 1 | 1  at s = 3P, 3P + 1 (zero beyond): one K=16 MFMA
// (P <= 4) or two (P <= 8) produce  bias + W[:, H:] x  in fp32-class accuracy from a ZERO accumulator input, so the gate
// accumulators need no initialisation at all.
__host__ __device__ inline int lem_slot_feature(int slot, int P) { return slot < 3 * P ? slot % P : -1; }
__host__ __device__ inline int lem_slot_part(int slot, int P) { return slot / P; }   // 0: (hi,hi) 1: (hi,lo) 2: (lo,hi)

// packed LEM blob (floats):  rec (16 chunks: g2 x4, g3 x4, g1 x4, lin x4) | mlp (8 chunks: Wa x4, Wb x4) |
//   bias [512] (g1, g2, g3, bz in the reference's row order) |
//   wxf [4 groups: g1,g2,g3,lin][4 T][4 s][64 lanes]: the input columns as MFMA A fragments, value
//        = W[128*group + 32T + (lane & 31)][H + 2s + (lane >> 5)] (0 past ninp) |
//   mlp bias [256] (ba, bb)
//   fp16-split copies for the split kernel (mfma_tiles.h): scales [8] (2^s of W, Wz, Wa, Wb, then 2^-s) |
//   rec_s (16 split chunks, acc order, same consumption order) | mlp_s (8 split chunks) |
//   bias_s [512], wxf_s [4096], mlpb_s [256]: the fp32 bias / input-column fragments pre-multiplied by 2^s of their matrix |
//   wx_h: the scaled input columns as fp16 slot fragments (see lem_slot_feature) |
//   rec_b3: the 16 recurrent chunks as bf16x3 fragments, acc order (training forward: fp32-exact products on the bf16 pipe)
struct LemLayout {
    int64_t rec, mlp, bias, wx, mlpb, scales, rec_s, mlp_s, bias_s, wx_s, mlpb_s, wx_h, rec_b3, total;
};

__host__ __device__ inline LemLayout lem_layout() {
    LemLayout L;
    int64_t o = 0;
    L.rec = o; o += 16 * CHUNK_FLOATS;
    L.mlp = o; o += 8 * CHUNK_FLOATS;
    L.bias = o; o += 4 * H;
    L.wx = o; o += 4 * H * LEM_MAX_INP;
    L.mlpb = o; o += 2 * H;
    L.scales = o; o += 8;
    L.rec_s = o; o += 16 * CHUNK_FLOATS;
    L.mlp_s = o; o += 8 * CHUNK_FLOATS;
    L.bias_s = o; o += 4 * H;
    L.wx_s = o; o += 4 * H * LEM_MAX_INP;
    L.mlpb_s = o; o += 2 * H;
    L.wx_h = o; o += LEM_WXH_FLOATS;
    L.rec_b3 = o; o += 16 * LEM_B3_CHUNK_FLOATS;
    L.total = o;
    return L;
}

__device__ __forceinline__ float tanhf_(float x) {
#if MSMP_PRECISE_ACT
    return tanhf(x);
#endif
    // (1 - e^{-2|x|}) / (1 + e^{-2|x|}) with the sign restored; absolute error ~1e-7
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.88539008177792681472f);
    const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return copysignf(r, x);
}

}  // namespace msmp
