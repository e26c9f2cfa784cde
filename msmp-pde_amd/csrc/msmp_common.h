// Shared device/host helpers for libmsmp_pde.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "msmp_pde.h"

int msmp_tune_get(const char* key);
int msmp_pair_project_aggregate(const float* h, const float* u, const float* pos, const float* vars, const int32_t* rowptr,
                                const int32_t* col, const int32_t* tgt, int64_t n_nodes, int64_t n_edges, int max_in_degree, int tw,
                                int nv, const float* packed_a, const float* packed_b, float* p_a, float* q_a, float* p_b, float* q_b,
                                float* agg_a, float* agg_b, msmp_stream_t stream);   // mlp_kernels.hip, library-internal
    // current value of a msmp_tune switch ("split", "tail"); library-internal
int msmp_edge_aggregate_tiled_pair(const float* h, const float* u, const float* pos, const float* vars, const float* feat, const int32_t* rowptr,
                                   const msmp_tiles_t* tiles, int64_t n_nodes, int64_t n_edges, int tw, int nv, const float* packed_a,
                                   const float* packed_b, float* agg_a, float* agg_b, msmp_stream_t stream);   // tile_kernels.hip

bool msmp_tiles_ok(const msmp_tiles_t* t, int64_t n_nodes);
int msmp_node_tail_impl(const float* h, const float* agg_main, const float* agg_gate, const float* vars, const int32_t* graph_ptr,
                        int64_t n_nodes, int64_t n_graphs, int max_graph_nodes, int nv, const float* packed_main, const float* packed_gate,
                        int mode, float eps, float* out, const msmp_decoder_t* dec, msmp_stream_t stream);       // mlp_kernels.hip    // tile_kernels.hip: geometry of a caller-supplied tile descriptor

namespace msmp {

constexpr int H = MSMP_HIDDEN;          // hidden width
constexpr int KC = 32;                  // k-chunk of a weight matrix staged through LDS
constexpr int LDW = 36;                 // LDS row stride (dwords) of a staged chunk: 4*odd, so the
                                        // 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots
constexpr int CHUNK_FLOATS = H * KC;    // one packed chunk: [128 out][32 k]
constexpr int VAR_SLOT_FLOATS = 2 * 4 * 64 * 8 / 2;   // [2 k-steps][4 row tiles][64 lanes][8 halfs]

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Packed layer blob (floats).  Chunks are [128 out][32 k] row-major, k ascending, zero padded; "split chunks"
// are the fp16-split copies of mfma_tiles.h (16 KB each, same footprint).  Sections whose size does not
// depend on tw come first, so the node kernels need no tw:
//   w3 (8 chunks: h | agg columns of update_net_1) | w4 (4 chunks) | b1 b2 b3 b4 ([128] each) |
//   w3v [128][MSMP_MAX_VARS] (variables columns of update_net_1) |
//   w3s (8 split chunks, natural k order) | w4s (4, acc order) | scales [8]: 2^s of w1..w4, then 2^-s of w1..w4 |
//   w3vh (the variables columns as two K=16 fp16 "slot" fragments per row tile, see var_slot_* in mfma_tiles.h) |
//   w4t (4 split chunks, acc order, fragment row of (tile T, lane c) = W4 row 4 c + T: the transposed node tail's B operand) |
//   w2t (4 split chunks, natural k order, rows dealt the same way: the weight-stationary edge kernel's B operand) |
//   w1 (nc1 chunks: h_i | h_j | u_i-u_j, p_i-p_j, vars_i, 0-pad) | w2 (4 chunks) |
//   w1s (nc1 split chunks, natural) | w2s (4, acc order) |
//   w1t (nc1 split chunks, natural, rows dealt round-robin like w4t: node_proj computes P / Q transposed)
// w4 directly follows w3 and w2 directly follows w1 (also in the split copies): the staging pipeline
// prefetches across the seam.
struct PackedLayout {
    int nc1;        // chunks of W1 (4 h_i + 4 h_j + tail chunks)
    int64_t w3, w4, b1, b2, b3, b4, w3v, w3s, w4s, scales, w3vh, w4t, w2t, w1, w2, w1s, w2s, w1t, total;
};

__host__ __device__ inline int tail_chunks(int tw, int nv) { return (tw + 1 + nv + KC - 1) / KC; }

__host__ __device__ inline PackedLayout packed_layout(int tw, int nv) {
    PackedLayout L;
    L.nc1 = 8 + tail_chunks(tw, nv);
    int64_t o = 0;
    L.w3 = o; o += 8 * CHUNK_FLOATS;
    L.w4 = o; o += 4 * CHUNK_FLOATS;
    L.b1 = o; o += H;
    L.b2 = o; o += H;
    L.b3 = o; o += H;
    L.b4 = o; o += H;
    L.w3v = o; o += H * MSMP_MAX_VARS;
    L.w3s = o; o += 8 * CHUNK_FLOATS;
    L.w4s = o; o += 4 * CHUNK_FLOATS;
    L.scales = o; o += 8;
    L.w3vh = o; o += VAR_SLOT_FLOATS;
    L.w4t = o; o += 4 * CHUNK_FLOATS;
    L.w2t = o; o += 4 * CHUNK_FLOATS;
    L.w1 = o; o += (int64_t)L.nc1 * CHUNK_FLOATS;
    L.w2 = o; o += 4 * CHUNK_FLOATS;
    L.w1s = o; o += (int64_t)L.nc1 * CHUNK_FLOATS;
    L.w2s = o; o += 4 * CHUNK_FLOATS;
    L.w1t = o; o += (int64_t)L.nc1 * CHUNK_FLOATS;
    L.total = o;
    return L;
}

// Swish(x) = x * sigmoid(x) (experiments/models_gnn.py:20-21) = x / (1 + 2^(-x*log2 e)).
// v_exp_f32 and v_rcp_f32 are 1-ulp; measured against the float64 oracle in tests/test_gpu_kernels.py.
// MSMP_PRECISE_ACT (diagnostic build, `MSMP_PRECISE=1 python msmp-pde_amd/build.py`): correctly-rounded division and libm exp /
// tanh / sqrt everywhere, to attribute the full-depth error of the network to the activation approximations or to the GEMMs.
#if MSMP_PRECISE_ACT
__device__ __forceinline__ float msmp_rcp(float x) { return __fdiv_rn(1.0f, x); }
__device__ __forceinline__ float msmp_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float msmp_rsq(float x) { return __fdiv_rn(1.0f, __fsqrt_rn(x)); }
__device__ __forceinline__ float swishf(float x) { return __fdiv_rn(x, 1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }
#else
__device__ __forceinline__ float msmp_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float msmp_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float msmp_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float swishf(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float sigmoidf_(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
#endif

// Row of a 32x32 MFMA accumulator register: D[row][col = lane & 31].
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// Sticky range status (msmp_last_status): `status` is the device-visible address of a host-mapped word, or nullptr.
constexpr float NODE_RANGE = 65504.0f / 256.0f;      // |x| representable as a node row of the split path (2^8 scaling)
__device__ __forceinline__ void status_raise(int* status, int bits) {
    if (status) __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool out_of_range(float x) { return !(fabsf(x) <= NODE_RANGE); }      // true for NaN as well
__device__ __forceinline__ bool out_of_range(f32x4 v) {
    return (int)out_of_range(v[0]) | (int)out_of_range(v[1]) | (int)out_of_range(v[2]) | (int)out_of_range(v[3]);
}
int* status_ptr();          // aux_kernels.hip: the status word as the device sees it (lazily created; nullptr if that failed)

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Optional event bracket around a kernel launch (msmp_timing_*): no-ops unless the family is enabled.
void timing_begin(int kernel, hipStream_t st);
void timing_end(int kernel, hipStream_t st);

}  // namespace msmp

#define MSMP_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            msmp::set_error(__VA_ARGS__);      \
            return (code);                     \
        }                                      \
    } while (0)
