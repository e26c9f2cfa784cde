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
#include <string.h>
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
    const int* rowptr;   // FUSE only
    long n_edges, n_nodes;
    int tile_nodes;      // FUSE only: nodes per workgroup tile (tile_nodes * max in-degree <= 128*NB)
    int tw, nv, nc1;
    const float* w1;   // nc1 chunks
    const float* w2;   // 4 chunks
    const float* w2s;  // 4 split chunks (acc order)
    const float* scales;  // [8] 2^s, 2^-s of w1..w4
    const float* b1;
    const float* b2;
    const float* P;      // FACT: [N,128] target-side projection  W1[:, h_i|u|p|v] . + b1   (node_proj_kernel)
    const float* Q;      // FACT: [N,128] source-side projection  W1[:, h_j] h - W1[:, u|p] [u, p]
    float* msg;          // !FUSE: [E,128] messages
    float* agg;          // FUSE:  [N,128] mean of the messages of each target
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

// FUSE = false: one workgroup per 128*NB consecutive CSR edges, writes msg [E,128] (row L1 alone).
// FUSE = true : one workgroup per tile of `tile_nodes` consecutive target nodes (all their in-edges, at
//               most 128*NB); after GEMM2 the messages are staged through LDS one 32-channel tile at a
//               time (re-using the weight buffers) and reduced per target in CSR order -> agg [N,128]
//               (rows L1 + L2): the [E,128] message tensor never touches HBM.
// FACT = true : message_net_1 is linear in the concatenation, so its pre-activation is P[i] + Q[j] with the
//               per-NODE projections P, Q made once by node_proj_kernel (5.3x fewer FLOPs on that term);
//               GEMM1 disappears from the edge kernel, Swish(P_i + Q_j) is formed straight in the B-operand
//               registers of GEMM2.  Rounding differs from the dense form only in where the partial sums
//               are rounded (validated against the float64 oracle with the same bars).
// SPLIT (with FACT): message_net_2 on the fp16 matrix pipe with the 2-way fp16 split of mfma_tiles.h (fp32-class
//               accuracy, 5.3x fewer matrix-pipe cycles); weights from the split chunks `w2s` of the blob.
template <int NB, bool FUSE, bool FACT, bool SPLIT>
__device__ __forceinline__ void edge_mlp_body(const EdgeArgs& a, float* lds) {
    static_assert(!SPLIT || FACT, "the split path is built for the factorised kernel");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    long tile_e0, tile_e1;     // CSR edge range of this workgroup
    int tile_n0 = 0, tile_n1 = 0;
    if (FUSE) {
        tile_n0 = blockIdx.x * a.tile_nodes;
        tile_n1 = min((long)tile_n0 + a.tile_nodes, a.n_nodes);
        tile_e0 = a.rowptr[tile_n0];
        tile_e1 = a.rowptr[tile_n1];
    } else {
        tile_e0 = (long)blockIdx.x * (128 * NB);
        tile_e1 = min(tile_e0 + 128 * NB, a.n_edges);
    }
    const long e0 = tile_e0 + (long)wave * (32 * NB);

    long e[NB];
    int ni[NB], nj[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        e[nb] = e0 + 32 * nb + c;
        const long ec = e[nb] < tile_e1 ? e[nb] : a.n_edges - 1;
        ni[nb] = a.tgt[ec];
        nj[nb] = a.col[ec];
    }

    f32x16 z[4][NB];
    WStage ws;
    int par;                     // LDS buffer holding W2 chunk 0
    if (FACT && SPLIT) {
        // lazy form: the P/Q pieces of channel tile t are gathered one chunk ahead inside the GEMM2 loop below, so the
        // 64 registers of the whole Swish(P_i + Q_j) row are never live at once (more workgroups per CU)
        wstage_load(ws, a.w2s, tid);
        wstage_store_linear(ws, lds, tid);
        __syncthreads();
        par = 0;
    } else if (FACT) {
        wstage_load(ws, SPLIT ? a.w2s : a.w2, tid);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float* pp = a.P + (size_t)ni[nb] * H + 4 * hh;
            const float* qp = a.Q + (size_t)nj[nb] * H + 4 * hh;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + 32 * T + 8 * q);
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 32 * T + 8 * q);
#pragma unroll
                    for (int m = 0; m < 4; ++m) z[T][nb][4 * q + m] = swishf(pv[m] + qv[m]);
                }
        }
        if (SPLIT) wstage_store_linear(ws, lds, tid);
        else wstage_store(ws, lds, tid);
        __syncthreads();
        par = 0;
    } else {
        acc_init_bias<NB>(a.b1, hh, z);
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
        par = a.nc1 & 1;
    }

    f32x16 y[4][NB];
    if (SPLIT) {
        acc_init_bias_scaled<NB>(a.b2, a.scales[1], hh, y);
        f32x4 pq[NB][8];                 // P (0..3) and Q (4..7) pieces of one 32-channel tile: channels 8q + 4hh .. +3
        auto gather_tile = [&](int t) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float* pp = a.P + (size_t)ni[nb] * H + 32 * t + 4 * hh;
                const float* qp = a.Q + (size_t)nj[nb] * H + 32 * t + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pq[nb][q] = *reinterpret_cast<const f32x4*>(pp + 8 * q);
                    pq[nb][4 + q] = *reinterpret_cast<const f32x4*>(qp + 8 * q);
                }
            }
        };
        gather_tile(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < 3) wstage_load(ws, a.w2s + (size_t)(t + 1) * SPLIT_CHUNK_FLOATS, tid);
            f32x16 zt[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < 4; ++m) zt[nb][4 * q + m] = swishf(pq[nb][q][m] + pq[nb][4 + q][m]);
            if (t < 3) gather_tile(t + 1);          // in flight during this chunk's matrix work
            half8 bhi[NB][2], blo[NB][2];
            split_acc_tile<NB>(zt, bhi, blo);
            mma_chunk_split<NB>(lds + (t & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, y);
            if (t < 3) {
                wstage_store_linear(ws, lds + ((t + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
                __syncthreads();
            }
        }
        const float inv2 = a.scales[5];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[T][nb][r] *= inv2;
    } else {
        acc_init_bias<NB>(a.b2, hh, y);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < 3) wstage_load(ws, a.w2 + (size_t)(t + 1) * CHUNK_FLOATS, tid);
            mma_chunk_from_acc<NB>(lds + ((par + t) & 1) * H * LDW, c, hh, z[t], y);
            if (t < 3) {
                wstage_store(ws, lds + ((par + t + 1) & 1) * H * LDW, tid);
                __syncthreads();
            }
        }
    }

    if (!FUSE) {
        // msg[e][32T + 8q + 4hh + m] = Swish(y)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (e[nb] < tile_e1) {
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
    } else {
        // Mean over the in-edges of each target.  The messages are staged through LDS (re-using the weight buffers)
        // CPR channels at a time as [local edge][CPR] (row stride CPR + 4), then thread (slot, cq) sums the rows of
        // node tile_n0 + slot (+ NSLOT, ...) for channels 4cq..4cq+3 in CSR order.  128-edge tiles stage 64 channels per
        // round (2 rounds, 4 barriers), 256-edge tiles 32 (4 rounds).  The CSR row bounds are fetched once, up front.
        constexpr int TPR = NB == 1 ? 2 : 1;              // 32-channel tiles per round
        constexpr int CPR = 32 * TPR, LDR = CPR + 4;      // channels per round, LDS row stride (dwords; 4*odd)
        constexpr int NCQ = CPR / 4, NSLOT = 256 / NCQ;
        const int cq = tid & (NCQ - 1), slot = tid / NCQ;
        int r0a[4], r1a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int node = tile_n0 + slot + k * NSLOT;
            const bool in = node < tile_n1;
            r0a[k] = in ? a.rowptr[node] - (int)tile_e0 : 0;
            r1a[k] = in ? a.rowptr[node + 1] - (int)tile_e0 : 0;
        }
#pragma unroll
        for (int R = 0; R < 4 / TPR; ++R) {
            __syncthreads();     // previous readers of lds (W2 chunk 3 / previous round) are done
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float* o = lds + (wave * 32 * NB + 32 * nb + c) * LDR + 4 * hh;
#pragma unroll
                for (int tt = 0; tt < TPR; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = swishf(y[R * TPR + tt][nb][4 * q + m]);
                        *reinterpret_cast<f32x4*>(o + 32 * tt + 8 * q) = v;
                    }
            }
            __syncthreads();
            int k = 0;
            for (int node = tile_n0 + slot; node < tile_n1; node += NSLOT, ++k) {
                int r0, r1;
                if (k < 4) { r0 = r0a[k < 4 ? k : 0]; r1 = r1a[k < 4 ? k : 0]; }
                else { r0 = a.rowptr[node] - (int)tile_e0; r1 = a.rowptr[node + 1] - (int)tile_e0; }
                // rows are fetched four at a time (independent LDS reads in flight) and added in CSR order; rows past the
                // end contribute an exact +0, so the result equals the sequential sum bit for bit
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                for (int r = r0; r < r1; r += 4) {
                    f32x4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = r + i < r1 ? *reinterpret_cast<const f32x4*>(lds + (r + i) * LDR + 4 * cq) : zero;
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum += v[i];
                }
                const float inv = 1.0f / (float)max(r1 - r0, 1);
                *reinterpret_cast<f32x4*>(a.agg + (size_t)node * H + CPR * R + 4 * cq) = sum * inv;
            }
        }
    }
}

template <int NB, bool FUSE, bool FACT, bool SPLIT = false>
__global__ __launch_bounds__(256) void edge_mlp_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    edge_mlp_body<NB, FUSE, FACT, SPLIT>(a, lds);
}

// Same body held to <= 256 registers so that two workgroups share a CU (one's gathers / Swish / reduce overlap the
// other's matrix work).
template <int NB, bool FUSE, bool FACT, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void edge_mlp_kernel_occ2(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    edge_mlp_body<NB, FUSE, FACT, SPLIT>(a, lds);
}

template <int NB, bool FUSE, bool FACT, bool SPLIT = false>
__global__ __launch_bounds__(256, 4) void edge_mlp_kernel_occ4(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    edge_mlp_body<NB, FUSE, FACT, SPLIT>(a, lds);
}

// ----------------------------------------------------------------------------------------------
// Per-node projections of message_net_1 (FACT path).  For node n:
//   P[n] = W1[:, 0:128] h_n + W1[:, 256:] [u_n, p_n, v_n] + b1      (what the node contributes as a TARGET i)
//   Q[n] = W1[:, 128:256] h_n - W1[:, 256:] [u_n, p_n, 0]           (what it contributes as a SOURCE j)
// so that W1 [h_i, h_j, u_i-u_j, p_i-p_j, v_i] + b1 = P[i] + Q[j].  Same channel-major scheme, one node per
// lane; P and Q share every B fragment (h_n chunk) and, in the tail chunk, every A fragment.
// ----------------------------------------------------------------------------------------------
struct ProjArgs {
    const float* h;
    const float* u;
    const float* pos;
    const float* vars;
    long n_nodes;
    int tw, nv, nc1;
    const float* w1;   // nc1 chunks
    const float* b1;
    float* P;
    float* Q;
};

__global__ __launch_bounds__(256) void node_proj_kernel(ProjArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;

    f32x16 p[4][1], qa[4][1];
    acc_init_bias<1>(a.b1, hh, p);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) qa[T][0][r] = 0.f;

    // chunk stream: 0..3 (h_i columns -> P), 4..7 (h_j columns -> Q), 8.. (tail -> both); B = h_n chunk (ch & 3)
    WStage ws;
    wstage_load(ws, a.w1, tid);
    wstage_store(ws, lds, tid);
    __syncthreads();
    const float* hp = a.h + (size_t)nc * H + 4 * hh;
    for (int ch = 0; ch < 8; ++ch) {
        wstage_load(ws, a.w1 + (size_t)(ch + 1) * CHUNK_FLOATS, tid);
        f32x4 b[1][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[0][q] = *reinterpret_cast<const f32x4*>(hp + 32 * (ch & 3) + 8 * q);
        if (ch < 4) mma_chunk<1>(lds + (ch & 1) * H * LDW, c, hh, b, p);
        else mma_chunk<1>(lds + (ch & 1) * H * LDW, c, hh, b, qa);
        wstage_store(ws, lds + ((ch + 1) & 1) * H * LDW, tid);
        __syncthreads();
    }
    const float* un = a.u + (size_t)nc * a.tw;
    for (int ch = 8; ch < a.nc1; ++ch) {
        if (ch + 1 < a.nc1) wstage_load(ws, a.w1 + (size_t)(ch + 1) * CHUNK_FLOATS, tid);
        f32x4 bp[1][4], bq[1][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = 32 * (ch - 8) + 8 * q + 4 * hh + m;
                float vp = 0.f, vq = 0.f;
                if (k < a.tw) { vp = un[k]; vq = -vp; }
                else if (k == a.tw) { vp = a.pos[nc]; vq = -vp; }
                else if (k <= a.tw + a.nv) vp = a.vars[(size_t)nc * a.nv + (k - a.tw - 1)];
                bp[0][q][m] = vp;
                bq[0][q][m] = vq;
            }
        mma_chunk<1>(lds + (ch & 1) * H * LDW, c, hh, bp, p);
        mma_chunk<1>(lds + (ch & 1) * H * LDW, c, hh, bq, qa);
        if (ch + 1 < a.nc1) wstage_store(ws, lds + ((ch + 1) & 1) * H * LDW, tid);
        __syncthreads();
    }
    if (n < a.n_nodes) {
        float* po = a.P + (size_t)n * H + 4 * hh;
        float* qo = a.Q + (size_t)n * H + 4 * hh;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v, w;
#pragma unroll
                for (int m = 0; m < 4; ++m) { v[m] = p[T][0][4 * q + m]; w[m] = qa[T][0][4 * q + m]; }
                *reinterpret_cast<f32x4*>(po + 32 * T + 8 * q) = v;
                *reinterpret_cast<f32x4*>(qo + 32 * T + 8 * q) = w;
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

// ----------------------------------------------------------------------------------------------
// fp16-split editions of the two node kernels (mfma_tiles.h).
// B operand: the workgroup's 128 node rows are read COALESCED (8 threads x 16 B = one 128-B line per node and
// 32-channel chunk), split into hi/lo fp16 by the loading thread and staged in LDS as
//   [node 128][hi 32 halfs | lo 32 halfs] (+16 B pad: row stride 144 B = 36 dwords -> conflict-free ds_read_b128),
// double buffered beside the weight chunks; lane (n, h) then reads its four fragments (hi/lo x two K=16 steps,
// natural k order) with 16-B LDS reads.  (Letting every lane fetch its own 512-B row straight from memory touches
// each cache line four times with 32-B pieces: measured, that row gather was 34 % of the kernel.)
// ----------------------------------------------------------------------------------------------
constexpr int BROW = 72;                         // halfs per staged node row (64 used)
constexpr int BTILE_FLOATS = 128 * BROW / 2;     // one staged B tile in floats (18 KB)

struct BStage {
    f32x4 r[4];
};

// thread t covers (node = (t + 256 i) >> 3, 4 channels at 4 ((t + 256 i) & 7)) for i = 0..3
__device__ __forceinline__ void bstage_load(BStage& b, const float* __restrict__ src, long n0, long n_nodes, int ch32, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        long node = n0 + (idx >> 3);
        node = node < n_nodes ? node : n_nodes - 1;
        b.r[i] = *reinterpret_cast<const f32x4*>(src + (size_t)node * H + ch32 + 4 * (idx & 7));
    }
}

__device__ __forceinline__ void bstage_store(const BStage& b, _Float16* tile, int tid) {
    using half4 = __attribute__((ext_vector_type(4))) _Float16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        half4 hi, lo;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const _Float16 h = (_Float16)b.r[i][m];
            hi[m] = h;
            lo[m] = (_Float16)(b.r[i][m] - (float)h);
        }
        _Float16* row = tile + (idx >> 3) * BROW + 4 * (idx & 7);
        *reinterpret_cast<half4*>(row) = hi;
        *reinterpret_cast<half4*>(row + 32) = lo;
    }
}

__device__ __forceinline__ void bfrag_read(const _Float16* tile, int node_local, int hh, half8 (&bhi)[1][2], half8 (&blo)[1][2]) {
    const _Float16* row = tile + node_local * BROW + 8 * hh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bhi[0][s] = *reinterpret_cast<const half8*>(row + 16 * s);
        blo[0][s] = *reinterpret_cast<const half8*>(row + 32 + 16 * s);
    }
}

struct ProjSplitArgs {
    ProjArgs b;
    const float* w1s;     // nc1 split chunks (natural order)
    const float* scales;  // [8]
};

__global__ __launch_bounds__(256, 2) void node_proj_split_kernel(ProjSplitArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS + 2 * BTILE_FLOATS];
    float* wbuf = lds;
    _Float16* bbuf = reinterpret_cast<_Float16*>(lds + 2 * SPLIT_CHUNK_FLOATS);
    const ProjArgs& a = sa.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = (long)blockIdx.x * 128;
    const long n = n0 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const int nl = wave * 32 + c;
    const float sc = sa.scales[0], inv = sa.scales[4];

    f32x16 p[4][1], qa[4][1];
    acc_init_bias_scaled<1>(a.b1, sc, hh, p);
    acc_zero<1>(qa);
    // step st = 0..7 uses weight chunk (st & 1 ? 4 : 0) + (st >> 1) (P then Q of the same h chunk st >> 1)
    WStage ws;
    BStage bs;
    wstage_load(ws, sa.w1s, tid);
    bstage_load(bs, a.h, n0, a.n_nodes, 0, tid);
    wstage_store_linear(ws, wbuf, tid);
    bstage_store(bs, bbuf, tid);
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int nxt = st + 1;                                   // next step's weight chunk (8 = first tail chunk)
        const int nchunk = nxt < 8 ? ((nxt & 1) ? 4 : 0) + (nxt >> 1) : 8;
        wstage_load(ws, sa.w1s + (size_t)nchunk * SPLIT_CHUNK_FLOATS, tid);
        if ((st & 1) && st < 7) bstage_load(bs, a.h, n0, a.n_nodes, 32 * ((st >> 1) + 1), tid);
        half8 bhi[1][2], blo[1][2];
        bfrag_read(bbuf + ((st >> 1) & 1) * 128 * BROW, nl, hh, bhi, blo);
        if (st & 1) mma_chunk_split<1>(wbuf + (st & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, qa);
        else mma_chunk_split<1>(wbuf + (st & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, p);
        wstage_store_linear(ws, wbuf + ((st + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        if ((st & 1) && st < 7) bstage_store(bs, bbuf + (((st >> 1) + 1) & 1) * 128 * BROW, tid);
        __syncthreads();
    }
    // tail chunks 8.. : [u_n, p_n, v_n] for P and [-u_n, -p_n, 0] for Q (per-lane scalar loads; 1-2 chunks)
    const float* un = a.u + (size_t)nc * a.tw;
    for (int ch = 8; ch < a.nc1; ++ch) {
        if (ch + 1 < a.nc1) wstage_load(ws, sa.w1s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS, tid);
        half8 phi[1][2], plo[1][2], qhi[1][2], qlo[1][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vp[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * (ch - 8) + 16 * s + 8 * hh + j;
                float x = 0.f, y = 0.f;
                if (k < a.tw) { x = un[k]; y = -x; }
                else if (k == a.tw) { x = a.pos[nc]; y = -x; }
                else if (k <= a.tw + a.nv) x = a.vars[(size_t)nc * a.nv + (k - a.tw - 1)];
                vp[j] = x;
                vq[j] = y;
            }
            split8(vp, phi[0][s], plo[0][s]);
            split8(vq, qhi[0][s], qlo[0][s]);
        }
        mma_chunk_split<1>(wbuf + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, phi, plo, p);
        mma_chunk_split<1>(wbuf + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, qhi, qlo, qa);
        if (ch + 1 < a.nc1) wstage_store_linear(ws, wbuf + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __syncthreads();
    }
    if (n < a.n_nodes) {
        float* po = a.P + (size_t)n * H + 4 * hh;
        float* qo = a.Q + (size_t)n * H + 4 * hh;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v, w;
#pragma unroll
                for (int m = 0; m < 4; ++m) { v[m] = p[T][0][4 * q + m] * inv; w[m] = qa[T][0][4 * q + m] * inv; }
                *reinterpret_cast<f32x4*>(po + 32 * T + 8 * q) = v;
                *reinterpret_cast<f32x4*>(qo + 32 * T + 8 * q) = w;
            }
    }
}

// node_update keeps the per-lane row gather: with 50 KB less LDS three workgroups share a CU, which measured
// faster (1.28 vs 1.39 ms per step) than the coalesced staging that pays off in node_proj (1.34 vs 1.46).
__device__ __forceinline__ void gather_split_row(const float* __restrict__ row32, int hh, half8 (&bhi)[1][2], half8 (&blo)[1][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(row32 + 16 * s + 8 * hh);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(row32 + 16 * s + 8 * hh + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(v, bhi[0][s], blo[0][s]);
    }
}

struct NodeSplitArgs {
    NodeArgs b;
    const float* w3s;     // 8 split chunks (natural), followed by w4s: 4 split chunks (acc order)
    const float* scales;  // [8]
};

__global__ __launch_bounds__(256, 2) void node_update_split_kernel(NodeSplitArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS];
    const NodeArgs& a = sa.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const float sc3 = sa.scales[2], inv3 = sa.scales[6], sc4 = sa.scales[3], inv4 = sa.scales[7];

    // acc init = (b3 + W3[:, 256:256+nv] vars_n) * 2^s3
    f32x16 z[4][1];
    acc_init_bias<1>(a.b3, hh, z);
    for (int v = 0; v < a.nv; ++v) {
        const float xv = a.vars[(size_t)nc * a.nv + v];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                z[T][0][r] = fmaf(a.w3v[(32 * T + acc_row(r, hh)) * MSMP_MAX_VARS + v], xv, z[T][0][r]);
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] *= sc3;

    WStage ws;
    wstage_load(ws, sa.w3s, tid);
    wstage_store_linear(ws, lds, tid);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < 8; ++ch) {
        wstage_load(ws, sa.w3s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS, tid);      // chunk 8 = w4s chunk 0
        half8 bhi[1][2], blo[1][2];
        gather_split_row((ch < 4 ? a.h : a.agg) + (size_t)nc * H + 32 * (ch & 3), hh, bhi, blo);
        mma_chunk_split<1>(lds + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, z);
        wstage_store_linear(ws, lds + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __syncthreads();
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] = swishf(z[T][0][r] * inv3);

    f32x16 y[4][1];
    acc_init_bias_scaled<1>(a.b4, sc4, hh, y);
    const float* w4s = sa.w3s + 8 * SPLIT_CHUNK_FLOATS;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) wstage_load(ws, w4s + (size_t)(t + 1) * SPLIT_CHUNK_FLOATS, tid);
        half8 bhi[1][2], blo[1][2];
        split_acc_tile<1>(z[t], bhi, blo);
        mma_chunk_split<1>(lds + (t & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, y);
        if (t < 3) {
            wstage_store_linear(ws, lds + ((t + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
            __syncthreads();
        }
    }
    if (n < a.n_nodes) {
        float* o = a.out + (size_t)n * H + 4 * hh;
        const float* hx = a.h + (size_t)n * H + 4 * hh;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
                if (a.mode == MSMP_LAYER_LIN) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = y[T][0][4 * q + m] * inv4;
                } else {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(hx + 32 * T + 8 * q);
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = x[m] + swishf(y[T][0][4 * q + m] * inv4);
                }
                *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = v;
            }
    }
}

}  // namespace msmp

using namespace msmp;

extern int g_lem_split;
static int g_split = 1;      // fp16-split matrix path (default); msmp_tune("split", 0) selects the fp32-MFMA kernels

extern "C" int msmp_edge_mlp_f32(const float* h, const float* u, const float* pos, const float* vars,
                                 const int32_t* tgt, const int32_t* col, int64_t n_nodes, int64_t n_edges,
                                 int tw, int nv, const float* packed, float* msg_out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && tgt && col && packed && msg_out, MSMP_ERR_ARG, "msmp_edge_mlp_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_edges >= 0 && tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG,
                 "msmp_edge_mlp_f32: bad sizes N=%ld E=%ld tw=%d nv=%d", (long)n_nodes, (long)n_edges, tw, nv);
    MSMP_REQUIRE(n_edges < (1L << 31) && n_nodes < (1L << 31), MSMP_ERR_UNSUPPORTED, "msmp_edge_mlp_f32: int32 index range");
    if (n_edges == 0) return MSMP_OK;
    const PackedLayout L = packed_layout(tw, nv);
    EdgeArgs a{h, u, pos, vars, tgt, col, nullptr, (long)n_edges, (long)n_nodes, 0, tw, nv, L.nc1,
               packed + L.w1, packed + L.w2, packed + L.w2s, packed + L.scales, packed + L.b1, packed + L.b2, nullptr, nullptr, msg_out, nullptr};
    constexpr int NB = 2;
    const unsigned grid = (unsigned)((n_edges + 128 * NB - 1) / (128 * NB));
    timing_begin(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    hipLaunchKernelGGL((edge_mlp_kernel<NB, false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    return check_launch("edge_mlp_kernel");
}

extern "C" int msmp_node_project_f32(const float* h, const float* u, const float* pos, const float* vars, int64_t n_nodes,
                                     int tw, int nv, const float* packed, float* p_out, float* q_out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && packed && p_out && q_out, MSMP_ERR_ARG, "msmp_node_project_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG,
                 "msmp_node_project_f32: bad sizes");
    const PackedLayout L = packed_layout(tw, nv);
    ProjArgs a{h, u, pos, vars, (long)n_nodes, tw, nv, L.nc1, packed + L.w1, packed + L.b1, p_out, q_out};
    const unsigned grid = (unsigned)((n_nodes + 127) / 128);
    timing_begin(MSMP_K_NODE_PROJ, (hipStream_t)stream);
    if (g_split) {
        ProjSplitArgs sa{a, packed + L.w1s, packed + L.scales};
        hipLaunchKernelGGL(node_proj_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa);
    } else
        hipLaunchKernelGGL(node_proj_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_NODE_PROJ, (hipStream_t)stream);
    return check_launch("node_proj_kernel");
}

static int g_edge_occ = 4;     // 4 waves per SIMD (128 registers) measured 5 % faster than 3 (134 registers)
static int g_edge_nb = 0;    // tuning override (msmp_tune): 0 = automatic, 1 / 2 = force the tile size of the factorised kernel

extern "C" int msmp_tune(const char* key, int value) {
    if (key && !strcmp(key, "edge_nb")) { g_edge_nb = value; return MSMP_OK; }
    if (key && !strcmp(key, "edge_occ")) { g_edge_occ = value; return MSMP_OK; }
    if (key && !strcmp(key, "lem")) { g_lem_split = value; return MSMP_OK; }
    if (key && !strcmp(key, "split")) { g_split = value; g_lem_split = value ? 3 : 0; return MSMP_OK; }
    msmp::set_error("msmp_tune: unknown key");
    return MSMP_ERR_ARG;
}

static int edge_aggregate(const float* h, const float* u, const float* pos, const float* vars, const float* P, const float* Q,
                          const int32_t* rowptr, const int32_t* col, const int32_t* tgt, int64_t n_nodes, int64_t n_edges,
                          int max_in_degree, int tw, int nv, const float* packed, float* agg_out, msmp_stream_t stream,
                          const char* who) {
    MSMP_REQUIRE(rowptr && col && tgt && packed && agg_out, MSMP_ERR_ARG, "%s: null pointer", who);
    MSMP_REQUIRE(n_nodes > 0 && n_edges >= 0 && tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS && max_in_degree >= 0, MSMP_ERR_ARG,
                 "%s: bad sizes", who);
    MSMP_REQUIRE(n_edges < (1L << 31) && n_nodes < (1L << 31), MSMP_ERR_UNSUPPORTED, "%s: int32 index range", who);
    if (n_edges == 0) {     // every node has an empty neighbourhood: the mean is 0
        const hipError_t me = hipMemsetAsync(agg_out, 0, (size_t)n_nodes * H * sizeof(float), (hipStream_t)stream);
        MSMP_REQUIRE(me == hipSuccess, MSMP_ERR_HIP, "%s: memset: %s", who, hipGetErrorString(me));
        return MSMP_OK;
    }
    MSMP_REQUIRE(max_in_degree <= 256, MSMP_ERR_UNSUPPORTED,
                 "%s: max in-degree %d > 256 (use msmp_edge_mlp_f32 + msmp_scatter_mean_f32)", who, max_in_degree);
    // Tile = 128 edges (NB = 1: <= 256 registers, two workgroups per CU so one's gathers / Swish / reduce overlap the
    // other's MFMAs) when the factorised form is used and the degrees allow, else 256 edges (NB = 2).
    const int edges_per_tile = (P && max_in_degree <= 128 && g_edge_nb != 2) ? 128 : 256;
    const PackedLayout L = packed_layout(tw, nv);
    int tile_nodes = max_in_degree > 0 ? edges_per_tile / max_in_degree : edges_per_tile;
    if (tile_nodes > 256) tile_nodes = 256;      // keeps the per-tile node loop short when degrees are tiny
    EdgeArgs a{h, u, pos, vars, tgt, col, rowptr, (long)n_edges, (long)n_nodes, tile_nodes, tw, nv, L.nc1,
               packed + L.w1, packed + L.w2, packed + L.w2s, packed + L.scales, packed + L.b1, packed + L.b2, P, Q, nullptr, agg_out};
    const unsigned grid = (unsigned)((n_nodes + tile_nodes - 1) / tile_nodes);
    timing_begin(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    if (P && edges_per_tile == 128 && g_split && g_edge_occ == 4) hipLaunchKernelGGL((edge_mlp_kernel_occ4<1, true, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (P && edges_per_tile == 128 && g_split) hipLaunchKernelGGL((edge_mlp_kernel_occ2<1, true, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (P && edges_per_tile == 128) hipLaunchKernelGGL((edge_mlp_kernel<1, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (P) hipLaunchKernelGGL((edge_mlp_kernel<2, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((edge_mlp_kernel<2, true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    return check_launch("edge_mlp_kernel<fused mean>");
}

extern "C" int msmp_edge_aggregate_f32(const float* h, const float* u, const float* pos, const float* vars,
                                       const int32_t* rowptr, const int32_t* col, const int32_t* tgt, int64_t n_nodes,
                                       int64_t n_edges, int max_in_degree, int tw, int nv, const float* packed,
                                       float* agg_out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars, MSMP_ERR_ARG, "msmp_edge_aggregate_f32: null pointer");
    return edge_aggregate(h, u, pos, vars, nullptr, nullptr, rowptr, col, tgt, n_nodes, n_edges, max_in_degree, tw, nv, packed,
                          agg_out, stream, "msmp_edge_aggregate_f32");
}

extern "C" int msmp_edge_aggregate_projected_f32(const float* p, const float* q, const int32_t* rowptr, const int32_t* col,
                                                 const int32_t* tgt, int64_t n_nodes, int64_t n_edges, int max_in_degree,
                                                 int tw, int nv, const float* packed, float* agg_out, msmp_stream_t stream) {
    MSMP_REQUIRE(p && q, MSMP_ERR_ARG, "msmp_edge_aggregate_projected_f32: null pointer");
    return edge_aggregate(nullptr, nullptr, nullptr, nullptr, p, q, rowptr, col, tgt, n_nodes, n_edges, max_in_degree, tw, nv,
                          packed, agg_out, stream, "msmp_edge_aggregate_projected_f32");
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
    if (g_split) {
        NodeSplitArgs sa{a, packed + L.w3s, packed + L.scales};
        hipLaunchKernelGGL(node_update_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa);
    } else
        hipLaunchKernelGGL(node_update_kernel<NB>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_NODE_UPDATE, (hipStream_t)stream);
    return check_launch("node_update_kernel");
}
