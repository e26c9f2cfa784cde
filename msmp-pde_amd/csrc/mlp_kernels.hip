// Edge-message MLP (row L1) and node-update MLP (row L3) on the fp32 MFMA of gfx950.
//
// Orientation ("channel-major"): every GEMM is computed transposed, D[out][item] = W[out][k] * F[k][item],
// with the weight matrix as the MFMA A operand and the per-edge / per-node feature vectors as the
// B operand, one item (edge or node) per lane:
//   * lane (c = lane & 31, hh = lane >> 5) of a wave owns item c of the wave's 32-item block and feeds
//     B[k][c] for the two k values {4hh + m} of each v_mfma_f32_32x32x2_f32;
//   * features are gathered with 16-byte loads straight from the node arrays (h rows are L2 resident:
//     every node row is re-read by ~deg edges), so no activation ever goes through LDS;
//   * the 32x32 accumulator then has the item on the lane and 16 output channels in registers, which
//     is exactly the B-operand shape of the next GEMM (k = channel): after bias + Swish the first
//     GEMM's accumulators feed the second GEMM from registers, no LDS round trip, no shuffles;
//   * weights (A operand) are shared by the 4 waves of a workgroup and streamed through LDS in
//     [128 out][32 k] chunks (row stride 36 dwords -> conflict-free ds_read_b128), double buffered,
//     one barrier per chunk; the packed blob stores chunks contiguously so staging is a flat copy.
// k order inside a chunk is k = 8q + 4hh + m (q, m = 0..3): one 16-byte fragment load covers four
// MFMA steps for both operands.  fp32 MFMA is an exact k-ordered fmaf chain (no reduced precision).
#include "mfma_tiles.h"

namespace msmp {

// ----------------------------------------------------------------------------------------------
// L1: edge messages
// ----------------------------------------------------------------------------------------------
struct EdgeArgs {
    const float* h;
    const float* u;
    const float* pos;
    const float* vars;
    const int* tgt;
    const int* col;
    long n_edges;
    int tw, nv, nc1;
    const float* w1;   // nc1 chunks
    const float* w2;   // 4 chunks
    const float* b1;
    const float* b2;
    float* msg;
};

// B fragments of chunk `c` of the concatenated edge feature [h_i | h_j | u_i-u_j | p_i-p_j | v_i | 0...]
__device__ __forceinline__ void edge_gather(const EdgeArgs& a, int c, int i, int j, int hh, f32x4 (&b)[4]) {
    if (c < 8) {
        const float* p = a.h + (size_t)(c < 4 ? i : j) * H + 32 * (c & 3) + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
    } else {
        const int k0 = 32 * (c - 8) + 4 * hh;
        const float* ui = a.u + (size_t)i * a.tw;
        const float* uj = a.u + (size_t)j * a.tw;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = k0 + 8 * q + m;
                float v = 0.f;
                if (k < a.tw) v = ui[k] - uj[k];
                else if (k == a.tw) v = a.pos[i] - a.pos[j];
                else if (k <= a.tw + a.nv) v = a.vars[(size_t)i * a.nv + (k - a.tw - 1)];
                b[q][m] = v;
            }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void edge_mlp_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long e0 = (long)blockIdx.x * (128 * NB) + (long)wave * (32 * NB);

    long e[NB];
    int ni[NB], nj[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        e[nb] = e0 + 32 * nb + c;
        const long ec = e[nb] < a.n_edges ? e[nb] : a.n_edges - 1;
        ni[nb] = a.tgt[ec];
        nj[nb] = a.col[ec];
    }

    f32x16 z[4][NB];
    acc_init_bias<NB>(a.b1, hh, z);

    WStage ws;
    f32x4 bcur[NB][4], bnext[NB][4];
    wstage_load(ws, a.w1, tid);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) edge_gather(a, 0, ni[nb], nj[nb], hh, bcur[nb]);
    wstage_store(ws, lds, tid);
    __syncthreads();

    // GEMM1 over nc1 chunks; the last iteration prefetches W2 chunk 0 (w2 follows w1 in the blob).
    for (int ch = 0; ch < a.nc1; ++ch) {
        wstage_load(ws, a.w1 + (size_t)(ch + 1) * CHUNK_FLOATS, tid);
        if (ch + 1 < a.nc1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) edge_gather(a, ch + 1, ni[nb], nj[nb], hh, bnext[nb]);
        }
        mma_chunk<NB>(lds + (ch & 1) * H * LDW, c, hh, bcur, z);
        wstage_store(ws, lds + ((ch + 1) & 1) * H * LDW, tid);
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) bcur[nb][q] = bnext[nb][q];
    }

    // Swish in place: z becomes the B operand of GEMM2.
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[T][nb][r] = swishf(z[T][nb][r]);

    f32x16 y[4][NB];
    acc_init_bias<NB>(a.b2, hh, y);
    const int par = a.nc1 & 1;   // LDS buffer holding W2 chunk 0
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) wstage_load(ws, a.w2 + (size_t)(t + 1) * CHUNK_FLOATS, tid);
        mma_chunk_from_acc<NB>(lds + ((par + t) & 1) * H * LDW, c, hh, z[t], y);
        if (t < 3) {
            wstage_store(ws, lds + ((par + t + 1) & 1) * H * LDW, tid);
            __syncthreads();
        }
    }

    // msg[e][32T + 8q + 4hh + m] = Swish(y)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (e[nb] < a.n_edges) {
            float* o = a.msg + (size_t)e[nb] * H + 4 * hh;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = swishf(y[T][nb][4 * q + m]);
                    *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = v;
                }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// L3: node update
// ----------------------------------------------------------------------------------------------
struct NodeArgs {
    const float* h;
    const float* agg;
    const float* vars;
    long n_nodes;
    int nv, mode;
    const float* w3;   // 8 chunks
    const float* w4;   // 4 chunks
    const float* b3;
    const float* b4;
    const float* w3v;  // [128][MSMP_MAX_VARS]
    float* out;
};

template <int NB>
__global__ __launch_bounds__(256) void node_update_kernel(NodeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = (long)blockIdx.x * (128 * NB) + (long)wave * (32 * NB);

    long n[NB], nc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        n[nb] = n0 + 32 * nb + c;
        nc[nb] = n[nb] < a.n_nodes ? n[nb] : a.n_nodes - 1;
    }

    // acc init = b3 + W3[:, 256:256+nv] vars_n  (the variables columns of the concatenation, K = nv <= 8)
    f32x16 z[4][NB];
    acc_init_bias<NB>(a.b3, hh, z);
    for (int v = 0; v < a.nv; ++v) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float xv = a.vars[(size_t)nc[nb] * a.nv + v];
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    z[T][nb][r] = fmaf(a.w3v[(32 * T + acc_row(r, hh)) * MSMP_MAX_VARS + v], xv, z[T][nb][r]);
        }
    }

    WStage ws;
    f32x4 bcur[NB][4], bnext[NB][4];
    wstage_load(ws, a.w3, tid);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float* p = a.h + (size_t)nc[nb] * H + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) bcur[nb][q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
    }
    wstage_store(ws, lds, tid);
    __syncthreads();

#pragma unroll 1
    for (int ch = 0; ch < 8; ++ch) {
        // chunk 8 of this stream is W4 chunk 0 (w4 follows w3 in the blob)
        wstage_load(ws, a.w3 + (size_t)(ch + 1) * CHUNK_FLOATS, tid);
        if (ch + 1 < 8) {
            const int cn = ch + 1;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float* p = (cn < 4 ? a.h : a.agg) + (size_t)nc[nb] * H + 32 * (cn & 3) + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) bnext[nb][q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
            }
        }
        mma_chunk<NB>(lds + (ch & 1) * H * LDW, c, hh, bcur, z);
        wstage_store(ws, lds + ((ch + 1) & 1) * H * LDW, tid);
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) bcur[nb][q] = bnext[nb][q];
    }

#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[T][nb][r] = swishf(z[T][nb][r]);

    f32x16 y[4][NB];
    acc_init_bias<NB>(a.b4, hh, y);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) wstage_load(ws, a.w4 + (size_t)(t + 1) * CHUNK_FLOATS, tid);
        mma_chunk_from_acc<NB>(lds + (t & 1) * H * LDW, c, hh, z[t], y);
        if (t < 3) {
            wstage_store(ws, lds + ((t + 1) & 1) * H * LDW, tid);
            __syncthreads();
        }
    }

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (n[nb] < a.n_nodes) {
            float* o = a.out + (size_t)n[nb] * H + 4 * hh;
            const float* hx = a.h + (size_t)n[nb] * H + 4 * hh;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
                    if (a.mode == MSMP_LAYER_LIN) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = y[T][nb][4 * q + m];
                    } else {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(hx + 32 * T + 8 * q);
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = x[m] + swishf(y[T][nb][4 * q + m]);
                    }
                    *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = v;
                }
        }
    }
}

}  // namespace msmp

using namespace msmp;

extern "C" int msmp_edge_mlp_f32(const float* h, const float* u, const float* pos, const float* vars,
                                 const int32_t* tgt, const int32_t* col, int64_t n_nodes, int64_t n_edges,
                                 int tw, int nv, const float* packed, float* msg_out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && tgt && col && packed && msg_out, MSMP_ERR_ARG, "msmp_edge_mlp_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_edges >= 0 && tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG,
                 "msmp_edge_mlp_f32: bad sizes N=%ld E=%ld tw=%d nv=%d", (long)n_nodes, (long)n_edges, tw, nv);
    MSMP_REQUIRE(n_edges < (1L << 31) && n_nodes < (1L << 31), MSMP_ERR_UNSUPPORTED, "msmp_edge_mlp_f32: int32 index range");
    if (n_edges == 0) return MSMP_OK;
    const PackedLayout L = packed_layout(tw, nv);
    EdgeArgs a{h, u, pos, vars, tgt, col, (long)n_edges, tw, nv, L.nc1,
               packed + L.w1, packed + L.w2, packed + L.b1, packed + L.b2, msg_out};
    constexpr int NB = 2;
    const unsigned grid = (unsigned)((n_edges + 128 * NB - 1) / (128 * NB));
    timing_begin(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    hipLaunchKernelGGL(edge_mlp_kernel<NB>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    return check_launch("edge_mlp_kernel");
}

extern "C" int msmp_node_update_f32(const float* h, const float* agg, const float* vars, int64_t n_nodes, int nv,
                                    const float* packed, int mode, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && agg && vars && packed && out, MSMP_ERR_ARG, "msmp_node_update_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG, "msmp_node_update_f32: bad sizes");
    MSMP_REQUIRE(mode == MSMP_LAYER_LIN || mode == MSMP_LAYER_RESIDUAL_SWISH, MSMP_ERR_ARG, "msmp_node_update_f32: bad mode %d", mode);
    MSMP_REQUIRE(out != h, MSMP_ERR_ARG, "msmp_node_update_f32: out may not alias h");
    const PackedLayout L = packed_layout(1, nv);   // w3/w4/b3/b4/w3v offsets do not depend on tw
    NodeArgs a{h, agg, vars, (long)n_nodes, nv, mode, packed + L.w3, packed + L.w4,
               packed + L.b3, packed + L.b4, packed + L.w3v, out};
    constexpr int NB = 1;
    const unsigned grid = (unsigned)((n_nodes + 128 * NB - 1) / (128 * NB));
    timing_begin(MSMP_K_NODE_UPDATE, (hipStream_t)stream);
    hipLaunchKernelGGL(node_update_kernel<NB>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_NODE_UPDATE, (hipStream_t)stream);
    return check_launch("node_update_kernel");
}
