// Channel-major fp32-MFMA building blocks shared by the MLP kernels and the LEM encoder kernel.
// See the header comment of mlp_kernels.hip for the operand orientation.
#pragma once
#include "msmp_common.h"

namespace msmp {

// a wave-uniform, read-only fp32 word (the 2^s scales of the packed layer): read through the constant address space, i.e. as a scalar load
// that the compiler may issue early and keep -- as a plain global load it was re-issued after every barrier and waited for at once
__device__ __forceinline__ float uniform_ro(const float* p, int i) { return ((const __attribute__((address_space(4))) float*)p)[i]; }

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


// ------------------------------------------------------------------------------------------------
// fp32 GEMM on the fp16 matrix pipe ("2-way fp16 split"), 5.3x fewer matrix-pipe cycles than
// v_mfma_f32_32x32x2_f32 at fp32-class accuracy:
//   x = hi + lo,  hi = fp16_rne(x),  lo = fp16_rne(x - hi)          (|x - hi - lo| <= 2^-24 |x| while lo is normal)
//   a.b = hi_a hi_b + hi_a lo_b + lo_a hi_b + O(2^-24 |a b|)
// Three v_mfma_f32_32x32x16_f16 per K = 16 step into ONE fp32 accumulator.  fp16 products are exact in the
// fp32 accumulation and a K = 128 dot product is 24 accumulation steps instead of 64, so the result is at
// least as accurate as the fp32 MFMA chain (measured in tests/test_gpu_kernels.py: 2.3-3.1e-7 vs 4.0-6.6e-7).
// Range: the low part of a small number falls into the fp16 subnormals (quantum 6e-8).  For activations
// (O(1) values) that is an absolute error <= 3e-8, below fp32's own at that magnitude.  Weights are small
// (~0.05), so every weight matrix is pre-multiplied by a power of two 2^s chosen by the pack kernel such that
// max|w| 2^s is in [16, 32) (exact; low parts of all significant weights stay normal); accumulators are
// initialised with bias * 2^s and the result is multiplied by 2^-s (both exact).
// Weights are pre-split by the pack kernels into "split chunks" of 16 KB:
//   [s 2][T 4][plane 2][lane 64][8 halfs]   (one ds_read_b128 = one A fragment, lane-linear: conflict-free)
// covering [128 out][32 k]; element j of lane (r, h) is W[32T + r][k] with
//   natural order  k = 16s + 8h + j                        (B gathered from memory, 8 consecutive k per lane)
//   acc order      k = 16s + 8(j >> 2) + 4h + (j & 3)      (B taken from a 32x32 accumulator: registers 8s..8s+7)
// ------------------------------------------------------------------------------------------------
using half8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int SPLIT_CHUNK_FLOATS = 4096;
// Diagnostic builds (`MSMP_LOLO=1|2 python msmp-pde_amd/build.py`, never shipped): the fourth product lo_a lo_b of the split,
// level 1 in update_net_1 / update_net_2 of the node kernels (what feeds each InstanceNorm), level 2 in every split GEMM.
// Used by scripts/diag_lolo.py to attribute the full-depth error to the dropped term or to the 22-bit operands themselves.
#ifndef MSMP_LOLO
#define MSMP_LOLO 0
#endif
#define MSMP_MFMA_LOLO(level, acc, alo, blo) do { if (MSMP_LOLO >= (level)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, blo, acc, 0, 0, 0); } while (0)      // 16 KB, same footprint as an fp32 chunk

// packed fp32 arithmetic (two values per instruction at the single-value issue cost); the compiler scalarises most
// <2 x float> expressions, so the activation pipeline names the instructions
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {           // a - b
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 b, f32x2 c) {  // c - a b
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
using half2 = __attribute__((ext_vector_type(2))) _Float16;
// lo = fp16(x - float(hi)) for a pair: one mixed-precision FMA per value (x - hi is exact in fp32, so this equals the
// convert-back / subtract / convert sequence bit for bit, in 2 instructions instead of 5).  Inline asm is invisible to the
// compiler's hazard recogniser: x and hi must not be the direct result of an MFMA or a transcendental instruction
// (every caller passes loaded values or the output of ordinary VALU arithmetic), and the RESULT must not be read by an MFMA
// as SrcA / SrcB in the next instruction slot: gfx950 does not interlock VALU write -> MFMA operand read (scripts/micro/mfma_raw.hip:
// acc 12-14 of 16 with no instruction in between, right with one s_nop; the compiler pads its own instructions).  split8 ends
// with mfma_operand_guard for that reason.
__device__ __forceinline__ half2 split_lo_pair(half2 hi, f32x2 x) {
    half2 lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(x[0]), "v"(x[1]));
    return lo;
}

// orders every later reader of `v` behind two wait states after the inline-asm writes above it (see split_lo_pair)
// (NOT volatile: the data dependence through `v` is all the ordering needed; as a volatile asm every guard was also ordered against every
// other one, the gather message kernel kept all its split fragments alive at once and went from 128 to 194 registers)
__device__ __forceinline__ void mfma_operand_guard(half8& v) { asm("s_nop 1" : "+v"(v)); }

__device__ __forceinline__ void split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 v = {x[j], x[j + 1]};
        const half2 h = __builtin_convertvector(v, half2);
        const half2 l = split_lo_pair(h, v);
        hi[j] = h[0];
        hi[j + 1] = h[1];
        lo[j] = l[0];
        lo[j + 1] = l[1];
    }
    mfma_operand_guard(lo);
}

using half2v = __attribute__((ext_vector_type(2))) _Float16;
// hi = fp16(256 x), lo = fp16(256 x - hi) for a pair of node-row values: four mixed-precision FMAs (the scaling is exact, the
// difference is formed exactly inside the FMA: the same bits as multiply / convert / convert back / subtract / convert, which took
// eleven instructions with the saturating clamp round 3 had here).  No clamp: |x| >= 256 becomes an fp16 infinity, every product with
// it is Inf / NaN, the aggregate of every target that reads the row is not finite and the epilogue raises the status word (and
// Solver.forward evaluates the forward again on the exact-fp32 kernels) -- nothing saturates silently.
__device__ __forceinline__ void split_node_pair(float x0, float x1, half2v& hi, half2v& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo) : "v"(x0), "v"(x1), "s"(256.0f));
}

// eight node-row values -> B fragments (hi, lo) of one K = 16 step, scaled by 2^8: split_node_pair four times, both results guarded
__device__ __forceinline__ void split8_node(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        half2v h, l;
        split_node_pair(x[j], x[j + 1], h, l);
        hi[j] = h[0];
        hi[j + 1] = h[1];
        lo[j] = l[0];
        lo[j + 1] = l[1];
    }
    asm("s_nop 1" : "+v"(hi), "+v"(lo));      // mfma_operand_guard for both
}

// Per-node scalar columns (the <= 8 equation variables) enter a split GEMM as two K=16 fp16 MFMAs.  K slot s pairs
//   s in [0,8): w_hi[f] x_hi[f]     [8,16): w_hi[f] x_lo[f]     [16,24): w_lo[f] x_hi[f]     [24,32): zero      (f = s & 7)
// so the node side needs no indexing: k-step 0 is (x_hi | x_lo) on the (hh = 0 | 1) lanes, k-step 1 is (x_hi | 0).
__host__ __device__ inline int var_slot_part(int slot) { return slot >> 3; }      // 0: hi*hi  1: hi*lo  2: lo*hi  3: unused
__device__ __forceinline__ void var_slot_frags(const float (&x)[8], int hh, half8 (&bx)[2]) {
    half8 xh, xl, zero;
    split8(x, xh, xl);
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (_Float16)0.f;
    bx[0] = hh ? xl : xh;
    bx[1] = hh ? zero : xh;
}

__device__ __forceinline__ void wstage_store_linear(const WStage& s, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(buf + 4 * (tid + 256 * i)) = s.r[i];
}

// acc += W_chunk * B  for the 32 k of one split chunk; bhi/blo[nb][s] are the B fragments of the two K=16 steps.
template <int NB>
__device__ __forceinline__ void mma_chunk_split(const float* wl, int lane, const half8 (&bhi)[NB][2], const half8 (&blo)[NB][2],
                                                f32x16 (&acc)[4][NB]) {
    const half8* w = reinterpret_cast<const half8*>(wl) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const half8 ahi = w[((s * 4 + T) * 2 + 0) * 64];
            const half8 alo = w[((s * 4 + T) * 2 + 1) * 64];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                MSMP_MFMA_LOLO(1, acc[T][nb], alo, blo[nb][s]);
                acc[T][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[nb][s], acc[T][nb], 0, 0, 0);
                acc[T][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[nb][s], acc[T][nb], 0, 0, 0);
                acc[T][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[nb][s], acc[T][nb], 0, 0, 0);
            }
        }
}

// B fragments of one accumulator tile (acc order): step s takes registers 8s..8s+7.
template <int NB>
__device__ __forceinline__ void split_acc_tile(const f32x16 (&x)[NB], half8 (&bhi)[NB][2], half8 (&blo)[NB][2]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = x[nb][8 * s + j];
            split8(v, bhi[nb][s], blo[nb][s]);
        }
}

// acc[T][nb][r] = bias[row] * scale   (accumulator initialisation of a split GEMM whose weights carry 2^s)
template <int NB>
__device__ __forceinline__ void acc_init_bias_scaled(const float* __restrict__ bias, float scale, int hh, f32x16 (&acc)[4][NB]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 32 * T + 8 * q + 4 * hh);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[T][nb][4 * q + m] = bv[m] * scale;
        }
}

template <int NB>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[4][NB]) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[T][nb][r] = 0.f;
}

// Host/device: value of element (s, T, plane, lane, j) of a split chunk built from W[row][k0 + k] (row stride ld).
// fp32 products on the bf16 matrix pipe: x = hi + mid + lo EXACTLY with three truncated 8-bit pieces (bf16 has fp32's exponent
// range, so gradients of any magnitude survive -- the fp16 split of the inference kernels would need a per-tensor scale), and
//   a b = hi hi + (hi mid + mid hi) + (mid mid + hi lo + lo hi) + O(2^-24 a b):  six MFMAs per K = 16 step, fp32 accumulation.
// 6 x 32 cycles per 16 rows against 8 x 64 for v_mfma_f32_32x32x2_f32: the matrix time drops 2.7x and the kernel becomes what it
// should be, bound by reading A and B once (round 2; the exact-fp32 edition ran at 30 % of the fp32 MFMA peak, 9.4 ms per
// batch-512 iteration).
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
struct Bf3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ Bf3 split_bf16x3(const float (&x)[8]) {
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hb[2], mb[2], lb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float v = x[2 * i + k];
            hb[k] = __float_as_uint(v) & 0xffff0000u;
            const float r1 = v - __uint_as_float(hb[k]);              // exact
            mb[k] = __float_as_uint(r1) & 0xffff0000u;
            lb[k] = __float_as_uint(r1 - __uint_as_float(mb[k]));     // exact; its top 16 bits are taken by the permute below
        }
        h[i] = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);      // {x1.b3, x1.b2, x0.b3, x0.b2}: two bf16 per register
        m[i] = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
        l[i] = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
    }
    return Bf3{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, m), __builtin_bit_cast(bf16x8, l)};
}

__host__ __device__ inline int split_k_natural(int s, int h, int j) { return 16 * s + 8 * h + j; }
__host__ __device__ inline int split_k_acc(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

}  // namespace msmp
