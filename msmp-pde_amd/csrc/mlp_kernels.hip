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
#include "decoder_body.h"

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
    int xcd_map;         // FUSE only: workgroup b works on tile (b % 8) * ceil(tiles / 8) + b / 8 (consecutive tiles on one XCD)
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
    int* status;         // msmp_last_status word (or nullptr)
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
// Phase profile of the tail / edge kernels (build with MSMP_PROF=1 in the environment of build.py; scripts/prof_tail.py reads it):
// per-workgroup cycle sums of wave 0, kept in scalar registers and added to g_prof once at the end.
#if MSMP_PROF
__device__ unsigned long long g_prof[16];
#define PROF_DECL long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tp = __builtin_readcyclecounter();
#define PROF_ARGS , long long& tp, long long (&pacc)[12]
#define PROF_PASS , tp, pacc
#define PROF_MARK(i) do { const long long t_ = __builtin_readcyclecounter(); pacc[i] += t_ - tp; tp = t_; } while (0)
#define PROF_FLUSH if (tid == 0 && (blockIdx.x & 15) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_prof[i_], (unsigned long long)pacc[i_]); atomicAdd(&g_prof[15], 1ull); }   /* one workgroup in 16 reports: see tile_kernels.hip */
#if MSMP_PROF_EDGE
#define PROF_EDGE(i) PROF_MARK(i)
#define PROF_EDGE_FLUSH PROF_FLUSH
#endif
#else
#define PROF_DECL
#define PROF_ARGS
#define PROF_PASS
#define PROF_MARK(i)
#define PROF_FLUSH
#endif
#ifndef PROF_EDGE
#define PROF_EDGE(i)
#define PROF_EDGE_FLUSH
#endif
#if MSMP_PROF_PROJ
#define PROF_PROJ(i) PROF_MARK(i)
#define PROF_PROJ_DECL PROF_DECL
#define PROF_PROJ_FLUSH PROF_FLUSH
#else
#define PROF_PROJ(i)
#define PROF_PROJ_DECL
#define PROF_PROJ_FLUSH
#endif

template <int NB, bool FUSE, bool FACT, bool SPLIT>
__device__ __forceinline__ void edge_mlp_body(const EdgeArgs& a, float* lds) {
    static_assert(!SPLIT || FACT, "the split path is built for the factorised kernel");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    long tile_e0, tile_e1;     // CSR edge range of this workgroup
    int tile_n0 = 0, tile_n1 = 0;
    if (FUSE) {
        const unsigned tile = blockIdx.x;
        tile_n0 = tile * a.tile_nodes;
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

    PROF_DECL
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

    // CSR row bounds of the nodes this thread reduces in the mean epilogue: fetched here, before the matrix work, so their
    // latency is hidden (issued after the GEMM loop they were 6-12 % of the kernel); unconditional loads from clamped indices
    constexpr int EP_NSLOT = 256 / ((NB == 1 ? 64 : 32) / 4);
    int r0a[4] = {0, 0, 0, 0}, r1a[4] = {0, 0, 0, 0};
    if (FUSE) {
        const int ep_slot = tid / (256 / EP_NSLOT);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int node = tile_n0 + ep_slot + k * EP_NSLOT;
            const int nodec = node < tile_n1 ? node : tile_n1 - 1;
            const int v0 = a.rowptr[nodec], v1 = a.rowptr[nodec + 1];
            r0a[k] = node < tile_n1 ? v0 - (int)tile_e0 : 0;
            r1a[k] = node < tile_n1 ? v1 - (int)tile_e0 : 0;
        }
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
        PROF_EDGE(0);
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
            PROF_EDGE(1);
            if (t < 3) gather_tile(t + 1);          // in flight during this chunk's matrix work
            half8 bhi[NB][2], blo[NB][2];
            split_acc_tile<NB>(zt, bhi, blo);
            PROF_EDGE(2);
            mma_chunk_split<NB>(lds + (t & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, y);
            PROF_EDGE(3);
            if (t < 3) {
                wstage_store_linear(ws, lds + ((t + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
                __syncthreads();
            }
            PROF_EDGE(4);
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
        PROF_EDGE(5);
#pragma unroll
        for (int R = 0; R < 4 / TPR; ++R) {
            __syncthreads();     // previous readers of lds (W2 chunk 3 / previous round) are done
            PROF_EDGE(6);
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
            PROF_EDGE(7);
            __syncthreads();
            PROF_EDGE(8);
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
                    for (int i = 0; i < 4; ++i) {         // unconditional read of a clamped row, zeroed afterwards (no predicated blocks)
                        const f32x4 t = *reinterpret_cast<const f32x4*>(lds + min(r + i, r1 - 1) * LDR + 4 * cq);
                        v[i] = r + i < r1 ? t : zero;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum += v[i];
                }
                const float inv = 1.0f / (float)max(r1 - r0, 1);
                const f32x4 res = sum * inv;
                *reinterpret_cast<f32x4*>(a.agg + (size_t)node * H + CPR * R + 4 * cq) = res;
                if (out_of_range(res)) status_raise(a.status, MSMP_STATUS_NODE_SATURATED);
            }
            PROF_EDGE(9);
        }
    }
    PROF_EDGE_FLUSH
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

struct EdgeArgs2 {
    EdgeArgs head[2];
};
// both heads of a gated pair in one launch (blockIdx.y = head); see node_proj_split_pair_kernel
__global__ __launch_bounds__(256, 2) void edge_mlp_pair_kernel_occ2(EdgeArgs2 a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    edge_mlp_body<1, true, true, true>(a.head[blockIdx.y], lds);
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
    const float* w1s;     // nc1 split chunks, natural k order, rows dealt round-robin (packed_layout().w1t)
    const float* scales;  // [8]
};

// acc[T] += A B for the 32 k of one split chunk with the operands swapped: A = the caller's fragments (a node / edge per lane),
// B = the weight chunk in LDS, so acc is [32 items][channel = lane] (the transposed form of mma_chunk_split<1>)
__device__ __forceinline__ void mma_chunk_split_t(const float* wl, int lane, const half8 (&ahi)[1][2], const half8 (&alo)[1][2],
                                                  f32x16 (&acc)[4]) {
    const half8* w = reinterpret_cast<const half8*>(wl) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const half8 whi = w[((s * 4 + T) * 2 + 0) * 64];
            const half8 wlo = w[((s * 4 + T) * 2 + 1) * 64];
            MSMP_MFMA_LOLO(2, acc[T], alo[0][s], wlo);
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0][s], wlo, acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[0][s], whi, acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0][s], whi, acc[T], 0, 0, 0);
        }
}

// P and Q are computed TRANSPOSED (node fragments as A operand, the row-dealt w1t chunks as B): the accumulators hold
// [32 nodes of the wave][channel 4 c + T], so a lane stores 16 consecutive bytes of whole 512-byte rows (16 stores per
// matrix instead of 32 quarter-line pieces).  The tail chunk ([u, pos, vars] columns) is read with unconditional, clamped
// loads and selected afterwards: predicated per-element loads compiled into serially waited-for blocks (27 % of the kernel).
__device__ __forceinline__ void node_proj_split_body(const ProjSplitArgs& sa, float* lds) {
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
    PROF_PROJ_DECL

    f32x16 p[4], qa[4];
    {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + 4 * c) * sc;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[T][r] = bv[T]; qa[T][r] = 0.f; }
    }
    // step st = 0..7 uses weight chunk (st & 1 ? 4 : 0) + (st >> 1) (P then Q of the same h chunk st >> 1)
    // Issue order = order of first use (the memory counter retires in order): weight chunk 0 and the first h chunk, THEN the
    // tail features, which are consumed after the eight main steps -- issued first, they made the first LDS store wait for 33
    // scattered 4-byte loads (30 % of the kernel).
    WStage ws;
    BStage bs;
    wstage_load(ws, sa.w1s, tid);
    bstage_load(bs, a.h, n0, a.n_nodes, 0, tid);
    // tail features of this lane's node, k = 16 s + 8 hh + j: [u (tw), pos, vars (nv), 0...]; loads first, selection later
    float tu[16], tv[16];
    const float* un = a.u + (size_t)nc * a.tw;
    const float tpos = a.pos[nc];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = 16 * (i >> 3) + 8 * hh + (i & 7);
        tu[i] = un[min(k, a.tw - 1)];
        tv[i] = a.vars[(size_t)nc * a.nv + min(max(k - a.tw - 1, 0), a.nv - 1)];
    }
    wstage_store_linear(ws, wbuf, tid);
    bstage_store(bs, bbuf, tid);
    __syncthreads();
    PROF_PROJ(0);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int nxt = st + 1;                                   // next step's weight chunk (8 = first tail chunk)
        const int nchunk = nxt < 8 ? ((nxt & 1) ? 4 : 0) + (nxt >> 1) : 8;
        wstage_load(ws, sa.w1s + (size_t)nchunk * SPLIT_CHUNK_FLOATS, tid);
        if ((st & 1) && st < 7) bstage_load(bs, a.h, n0, a.n_nodes, 32 * ((st >> 1) + 1), tid);
        half8 bhi[1][2], blo[1][2];
        bfrag_read(bbuf + ((st >> 1) & 1) * 128 * BROW, nl, hh, bhi, blo);
        PROF_PROJ(1);
        if (st & 1) mma_chunk_split_t(wbuf + (st & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, qa);
        else mma_chunk_split_t(wbuf + (st & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, p);
        PROF_PROJ(2);
        wstage_store_linear(ws, wbuf + ((st + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        if ((st & 1) && st < 7) bstage_store(bs, bbuf + (((st >> 1) + 1) & 1) * 128 * BROW, tid);
        PROF_PROJ(3);
        __syncthreads();
        PROF_PROJ(4);
    }
    // tail chunks 8.. : [u_n, p_n, v_n] for P and [-u_n, -p_n, 0] for Q (1 chunk for tw <= 25 + ..., 2 for tw = 50)
    for (int ch = 8; ch < a.nc1; ++ch) {
        if (ch + 1 < a.nc1) wstage_load(ws, sa.w1s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS, tid);
        if (ch > 8) {                   // second tail chunk (tw = 50): fetch its columns now
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = 32 * (ch - 8) + 16 * (i >> 3) + 8 * hh + (i & 7);
                tu[i] = un[min(k, a.tw - 1)];
                tv[i] = a.vars[(size_t)nc * a.nv + min(max(k - a.tw - 1, 0), a.nv - 1)];
            }
        }
        half8 phi[1][2], plo[1][2], qhi[1][2], qlo[1][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vp[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * (ch - 8) + 16 * s + 8 * hh + j;
                const float x = k < a.tw ? tu[8 * s + j] : k == a.tw ? tpos : k <= a.tw + a.nv ? tv[8 * s + j] : 0.f;
                vp[j] = x;
                vq[j] = k <= a.tw ? -x : 0.f;
            }
            split8(vp, phi[0][s], plo[0][s]);
            split8(vq, qhi[0][s], qlo[0][s]);
        }
        PROF_PROJ(5);
        mma_chunk_split_t(wbuf + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, phi, plo, p);
        mma_chunk_split_t(wbuf + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, qhi, qlo, qa);
        if (ch + 1 < a.nc1) wstage_store_linear(ws, wbuf + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __syncthreads();
        PROF_PROJ(6);
    }
    // rows of this lane: node n0 + 32 wave + acc_row(r, hh), channels 4 c .. 4 c + 3
    const size_t base = ((size_t)n0 + wave * 32 + 4 * hh) * H + 4 * c;
    const long lim = a.n_nodes - n0 - wave * 32 - 4 * hh;
    if (n0 + wave * 32 + 32 <= a.n_nodes) {         // wave-uniform common path: no per-row predication
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t o = base + (size_t)((r & 3) + 8 * (r >> 2)) * H;
            *reinterpret_cast<f32x4*>(a.P + o) = f32x4{p[0][r], p[1][r], p[2][r], p[3][r]} * inv;
            *reinterpret_cast<f32x4*>(a.Q + o) = f32x4{qa[0][r], qa[1][r], qa[2][r], qa[3][r]} * inv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);
            if (row < lim) {
                const size_t o = base + (size_t)row * H;
                *reinterpret_cast<f32x4*>(a.P + o) = f32x4{p[0][r], p[1][r], p[2][r], p[3][r]} * inv;
                *reinterpret_cast<f32x4*>(a.Q + o) = f32x4{qa[0][r], qa[1][r], qa[2][r], qa[3][r]} * inv;
            }
        }
    }
    PROF_PROJ(7);
    PROF_PROJ_FLUSH
}

__global__ __launch_bounds__(256, 2) void node_proj_split_kernel(ProjSplitArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS + 2 * BTILE_FLOATS];
    node_proj_split_body(sa, lds);
}

// Both heads of a gated pair in ONE launch (blockIdx.y = head): same body, so the results are bit-identical to two launches.
// Used for small batches, where a rollout step is a chain of ~60 dependent launches and each one is ~10 us of latency.
struct ProjSplitArgs2 {
    ProjSplitArgs head[2];
};
__global__ __launch_bounds__(256, 2) void node_proj_split_pair_kernel(ProjSplitArgs2 a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS + 2 * BTILE_FLOATS];
    node_proj_split_body(a.head[blockIdx.y], lds);
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

// One update head on the split path: y = (W4 Swish(W3 [h ; agg ; vars] + b3) + b4) in accumulator layout (one node per lane),
// already multiplied by 2^-s4.  Ends with the weight buffers free (no barrier pending).
__device__ __forceinline__ void node_head_split(const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ vars,
                                                long nc, int nv, const float* b3, const float* b4, const float* w3v, const float* w3s,
                                                const float* scales, float* lds, int tid, int lane, int hh, f32x16 (&y)[4][1]) {
    const float sc3 = scales[2], inv3 = scales[6], sc4 = scales[3], inv4 = scales[7];
    // acc init = (b3 + W3[:, 256:256+nv] vars_n) * 2^s3
    f32x16 z[4][1];
    acc_init_bias<1>(b3, hh, z);
    for (int v = 0; v < nv; ++v) {
        const float xv = vars[(size_t)nc * nv + v];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                z[T][0][r] = fmaf(w3v[(32 * T + acc_row(r, hh)) * MSMP_MAX_VARS + v], xv, z[T][0][r]);
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] *= sc3;

    // The lane's slices of its h / agg rows (B operand of the 8 k-chunks) are prefetched FOUR chunks ahead: the loop is
    // a chain of short MFMA bursts between barriers, and a row load issued right before its use exposed the HBM
    // latency once per chunk (the kernel ran at a third of its matrix-pipe time).
    f32x4 pf[4][4];
    auto row_load = [&](int ch, f32x4 (&dst)[4]) {
        const float* row32 = (ch < 4 ? h : agg) + (size_t)nc * H + 32 * (ch & 3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            dst[2 * s] = *reinterpret_cast<const f32x4*>(row32 + 16 * s + 8 * hh);
            dst[2 * s + 1] = *reinterpret_cast<const f32x4*>(row32 + 16 * s + 8 * hh + 4);
        }
    };
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) row_load(ch, pf[ch]);
    WStage ws;
    wstage_load(ws, w3s, tid);
    wstage_store_linear(ws, lds, tid);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        wstage_load(ws, w3s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS, tid);      // chunk 8 = w4s chunk 0
        half8 bhi[1][2], blo[1][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 v0 = pf[ch & 3][2 * s], v1 = pf[ch & 3][2 * s + 1];
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split8(v, bhi[0][s], blo[0][s]);
        }
        if (ch + 4 < 8) row_load(ch + 4, pf[ch & 3]);
        mma_chunk_split<1>(lds + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, z);
        wstage_store_linear(ws, lds + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __syncthreads();
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] = swishf(z[T][0][r] * inv3);

    acc_init_bias_scaled<1>(b4, sc4, hh, y);
    const float* w4s = w3s + 8 * SPLIT_CHUNK_FLOATS;
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
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[T][0][r] *= inv4;
}

__global__ __launch_bounds__(256, 2) void node_update_split_kernel(NodeSplitArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS];
    const NodeArgs& a = sa.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    f32x16 y[4][1];
    node_head_split(a.h, a.agg, a.vars, nc, a.nv, a.b3, a.b4, a.w3v, sa.w3s, sa.scales, lds, tid, lane, hh, y);
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
                    for (int m = 0; m < 4; ++m) v[m] = y[T][0][4 * q + m];
                } else {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(hx + 32 * T + 8 * q);
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = x[m] + swishf(y[T][0][4 * q + m]);
                }
                *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = v;
            }
    }
}

// ----------------------------------------------------------------------------------------------
// Node tail of one layer in ONE launch (graphs of at most 128 nodes, one graph per workgroup):
//   gated pair (experiments/models_gnn.py:1366-1368):  h' = (1 - tau) h + tau Swish(IN(update_main)),  tau = sigmoid(IN(update_gate))
//   plain layer (:61-67 / :124-130):                    h' = IN(update)   (update = h + Swish(..) or the Lin form)
// IN = PyG InstanceNorm over the graph's nodes (biased variance, eps).  The pre-norm tensors never leave registers.
// update_net_1 is computed channel-major as in the other kernels (a node per lane); update_net_2 is computed TRANSPOSED
// (the Swish output fragments become the A operand, the W4 fragments the B operand), so its accumulator holds
// [32 nodes of the wave][channel = lane]: the per-channel sums over the graph are 16 in-lane adds per tile plus eight
// partials through LDS, each lane needs the statistics of four channels only, and h / h' move as 128-B rows.
// The variables columns are two fp16 slot MFMAs (var_slot_frags).  Against node_update x2 + gate_blend this reads
// h, agg_main, agg_gate and writes h' (420 MB instead of 1050 MB at 2048 x 100 nodes).
// ----------------------------------------------------------------------------------------------
struct TailArgs {
    const float* h;
    const float* agg[2];     // main, gate
    const float* vars;
    const int* graph_ptr;
    int nv, mode;
    float eps;
    const float* b3[2];
    const float* b4[2];
    const float* w3vh[2];
    const float* w3s[2];
    const float* w4t[2];
    const float* scales[2];
    float* out;
    int* status;             // msmp_last_status word (or nullptr)
    DecW dec;                // dec.w1 != nullptr: the 1-D decoder (time_window 25) runs as this launch's epilogue on the rows it wrote
};

// One update head, update_net_2 transposed: yT[T][r] = 2^s4 (W4 Swish(W3 [h ; agg ; vars] + b3) + b4)[channel 4 c + T]
// of node acc_row(r, hh) of the wave's 32-node tile (the w4t fragments deal the output channels round-robin over the
// four tiles, so a lane owns four CONSECUTIVE channels and h / h' move as 16-byte pieces of 512-byte rows).
// tail_rows_issue starts the LDS-DMA of the first 32-column chunk of the graph's rows (the gated kernel issues the main head's
// during the gate head's update_net_2); head_compute ends with the weight buffers free.
constexpr float TAIL_ACT_SCALE = 64.0f;       // hidden units of the update net (Swish outputs)
constexpr float TAIL_NODE_SCALE = 256.0f;     // node rows (h, aggregate, variables): saturating, see tile_kernels.hip
__device__ __forceinline__ float tail_node_scaled(float x) { return __builtin_amdgcn_fmed3f(x * TAIL_NODE_SCALE, -65504.0f, 65504.0f); }

// The graph's node rows reach the update GEMM through LDS, one 32-column chunk (128 rows x 128 B = 16 KB) at a time, filled by
// LDS-DMA (global_load_lds_dwordx4: no registers, no VALU): a wave-instruction copies 8 whole 128-byte lines, so every line of h /
// agg is fetched ONCE per head, coalesced.  Round 2 loaded the B fragments straight into registers, a row per lane: every
// instruction touched 32 different lines and every line was touched by four instructions (rocprofv3: 41 % of the wave cycles
// waiting on memory; replacing the row gather by a broadcast load removed 16 % of the kernel).  The 16-byte pieces of a row
// are XOR-swizzled on the SOURCE address (the LDS image of an LDS-DMA is lane-linear), so that the ds_read_b128 of a 16-lane
// group (16 rows, the same piece) hits 16 different 4-bank groups: piece p of row r sits in slot p ^ ((r >> 1) & 7).
constexpr int ROWBUF_FLOATS = 128 * 32;
__device__ __forceinline__ void tail_rows_issue(const float* __restrict__ h, const float* __restrict__ agg, int n0, int n1, int ch, float* buf, int tid) {
    const float* src = (ch < 4 ? h : agg) + 32 * (ch & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = tid + 256 * i, r = P >> 3;
        const int node = min(n0 + r, n1 - 1);
        const float* g = src + (size_t)node * H + 4 * ((P & 7) ^ ((r >> 1) & 7));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + 4 * P), 16, 0, 0);
    }
}
// this lane's 16 values of its row for the staged chunk: k = 16 s + 8 hh + 0..7
__device__ __forceinline__ void tail_rows_read(const float* buf, int row, int hh, f32x4 (&dst)[4]) {
    const float* rb = buf + row * 32;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dst[2 * s] = *reinterpret_cast<const f32x4*>(rb + 4 * ((4 * s + 2 * hh) ^ sw));
        dst[2 * s + 1] = *reinterpret_cast<const f32x4*>(rb + 4 * ((4 * s + 2 * hh + 1) ^ sw));
    }
}

// a 16-KB weight chunk global -> LDS by LDS-DMA (lane-linear image, as wstage_store_linear writes it)
__device__ __forceinline__ void wstage_dma(const float* __restrict__ chunk, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(chunk + 4 * (tid + 256 * i)),
                                         (__attribute__((address_space(3))) void*)(buf + 4 * (tid + 256 * i)), 16, 0, 0);
}

template <typename MidHook>
__device__ __forceinline__ void head_compute(float* rowbuf, int n0, int n1, const float* __restrict__ h, const float* __restrict__ agg,
                                             const float* __restrict__ vars, long nc, int nv, const float* b3, const float* b4,
                                             const float* w3vh, const float* w3s, const float* w4t, const float* scales, float* lds, const float* xvl,
                                             int tid, int lane, int c, int hh, f32x16 (&yT)[4], MidHook mid_hook, bool center, float* zref PROF_ARGS) {
    // ACT_SCALE: the node rows and the Swish output enter the split GEMMs multiplied by 2^6, so that the fp16 low halves of small
    // activations stay normal (see tile_kernels.hip); every factor is a power of two folded into an existing constant.
    const float sc3 = uniform_ro(scales, 2) * TAIL_NODE_SCALE, inv3 = uniform_ro(scales, 6) * (TAIL_ACT_SCALE / TAIL_NODE_SCALE), sc4 = uniform_ro(scales, 3) * TAIL_ACT_SCALE;
    // Weight chunks reach LDS by LDS-DMA (round 4; the staged image is lane-linear, i.e. exactly what global_load_lds_dwordx4 writes):
    // no register round trip (16 VGPRs, 4 ds_write_b128 per thread and chunk); a chunk is requested into the buffer the PREVIOUS
    // chunk's MFMAs read, free since the barrier that ended that iteration, and the barrier at the end of this one waits for it.
    wstage_dma(w3s, lds, tid);
    float xv[8];
    half8 wvf[2][4];
    f32x16 z[4][1];
    // acc_init_bias_scaled<1>(b3, sc3, hh, z) with all sixteen 16-byte loads in flight at once, INTO the accumulator registers (pinned
    // there by the empty asm), scaled in place: left to the compiler they went through one four-register temporary, four round trips
    // to L2 one after the other at the head of every update head (the phase profile's 3.7 k cycles of "head prologue")
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b3 + 32 * T + 8 * q + 4 * hh);
#pragma unroll
            for (int m = 0; m < 4; ++m) z[T][0][4 * q + m] = bv[m];
        }
#pragma unroll
    for (int T = 0; T < 4; ++T) asm volatile("" : "+v"(z[T][0]));
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] *= sc3;
    __syncthreads();                // (with the vmcnt(0) of the LDS-DMA in flight: chunk 0 of the rows has landed)
    PROF_MARK(5);
    const int myrow = (tid >> 6) * 32 + c;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        if (ch < 7) tail_rows_issue(h, agg, n0, n1, ch + 1, rowbuf + ((ch + 1) & 1) * ROWBUF_FLOATS, tid);
        wstage_dma(ch < 7 ? w3s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS : w4t, lds + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        half8 bhi[1][2], blo[1][2];
        {
            f32x4 pf[4];
            tail_rows_read(rowbuf + (ch & 1) * ROWBUF_FLOATS, myrow, hh, pf);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 v0 = pf[2 * s], v1 = pf[2 * s + 1];
                // hi = fp16(256 x), lo = fp16(256 x - hi) as four mixed-precision FMAs per pair (split_node_pair; no clamp: |x| >= 256 becomes
                // an fp16 infinity, the norm's statistics are not finite and the status word is raised -- tile_kernels.hip).  Round 4 first
                // measured this form as nondeterministic: the inline asm's results reached an MFMA in the next slot, which gfx950 does not
                // interlock (profiles/r04N_mfma_operand_hazard.md); split8_node ends with the guard.
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                split8_node(v, bhi[0][s], blo[0][s]);
            }
        }
        if (ch == 5) {      // the variables' slot fragments are consumed after the k loop: issued two chunks ahead
            const half8* wv = reinterpret_cast<const half8*>(w3vh) + lane;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int T = 0; T < 4; ++T) wvf[m][T] = wv[(m * 4 + T) * 64];
        }
        PROF_MARK(6);
        mma_chunk_split<1>(lds + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, z);
        PROF_MARK(7);
        PROF_MARK(8);
        __syncthreads();
        PROF_MARK(9);
    }
    {
        half8 bx[2];
        // the node's variables from the LDS table the kernel filled at its start (as `f < nv ? vars[...] : 0` inside the chunk loop every
        // load sat in a branch of its own with a wait right behind it: up to eight exposed round trips per head)
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xvl + 8 * myrow), x1 = *reinterpret_cast<const f32x4*>(xvl + 8 * myrow + 4);
        const float xr[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int f = 0; f < 8; ++f) xv[f] = tail_node_scaled(xr[f]);
        var_slot_frags(xv, hh, bx);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int T = 0; T < 4; ++T) z[T][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wvf[m][T], bx[m], z[T][0], 0, 0, 0);
    }
    mid_hook();             // the row registers are free from here on
    __builtin_amdgcn_sched_barrier(0);      // (the requests stay HERE: sunk to the barrier below, its fence waits for them where they are issued)
    PROF_MARK(10);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {         // z' = ACT_SCALE Swish(x), x = acc 2^-s3 / ACT_SCALE: same instruction count as Swish(x)
#if MSMP_PRECISE_ACT
            z[T][0][r] = TAIL_ACT_SCALE * swishf(z[T][0][r] * inv3 * (1.0f / TAIL_ACT_SCALE));
#else
            const float xs = z[T][0][r] * inv3;
            const float e = __builtin_amdgcn_exp2f(xs * (-1.44269504088896340736f / TAIL_ACT_SCALE));
            z[T][0][r] = xs * __builtin_amdgcn_rcpf(1.0f + e);
#endif
        }

    // `center` (the head's output goes straight into an InstanceNorm: GNN_LayerLin, experiments/models_gnn.py:129): the norm removes
    // every per-graph, per-channel constant, so update_net_2 is evaluated on z_n - z_ref (z_ref = the hidden units of the graph's
    // first node) and without its bias: y_n - y_ref = W4 (z_n - z_ref).  The accumulator then carries the VARIATION of y over the
    // graph instead of its value, and the 24 fp32 roundings of the K = 128 accumulation are relative to what the norm divides by.
    // (scripts/diag_quant.py: with untrained weights |y| is 10-100 x its spread over a graph, and this accumulation was 90 % of a
    // layer's error after the norm: 5.2e-7 of 5.4e-7 rms.)
    if (center) {
        if ((tid >> 6) == 0 && c == 0) {
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(zref + 32 * T + 8 * q + 4 * hh) = f32x4{z[T][0][4 * q], z[T][0][4 * q + 1], z[T][0][4 * q + 2], z[T][0][4 * q + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(zref + 32 * T + 8 * q + 4 * hh);
#pragma unroll
                for (int m = 0; m < 4; ++m) z[T][0][4 * q + m] -= rv[m];
            }
    }
    {
        const f32x4 bv = center ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(b4 + 4 * c) * sc4;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) yT[T][r] = bv[T];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) wstage_dma(w4t + (size_t)(t + 1) * SPLIT_CHUNK_FLOATS, lds + ((t + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __builtin_amdgcn_sched_barrier(0);  // (as above: the next chunk is requested at the top of the iteration, not in front of its barrier)
        half8 zhi[1][2], zlo[1][2];
        split_acc_tile<1>(z[t], zhi, zlo);
        const half8* w = reinterpret_cast<const half8*>(lds + (t & 1) * SPLIT_CHUNK_FLOATS) + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const half8 whi = w[((s * 4 + T) * 2 + 0) * 64], wlo = w[((s * 4 + T) * 2 + 1) * 64];
                MSMP_MFMA_LOLO(1, yT[T], zlo[0][s], wlo);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi[0][s], wlo, yT[T], 0, 0, 0);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zlo[0][s], whi, yT[T], 0, 0, 0);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi[0][s], whi, yT[T], 0, 0, 0);
            }
        if (t < 3) __syncthreads();
    }
}

// per-channel total over the workgroup of one per-lane value per channel tile (lane = channel 32 T + c; the two hh halves
// and the four waves hold partial sums): returns the totals of this lane's four channels
__device__ __forceinline__ void tile_t_total(const float (&v)[4], float* part, float* tot, int tid, int wave, int c, int hh,
                                             float (&out)[4]) {
#pragma unroll
    for (int T = 0; T < 4; ++T) part[(wave * 2 + hh) * H + 32 * T + c] = v[T];
    __syncthreads();
    if (tid < H) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) s += part[p * H + tid];
        tot[tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int T = 0; T < 4; ++T) out[T] = tot[32 * T + c];
}

// x <- (x - mean) / sqrt(var + eps) per channel over the graph's cnt nodes; x is in units of `unit` (a power of two)
__device__ __forceinline__ void tile_t_instance_norm(f32x16 (&x)[4], int wave, int cnt, float unit, float eps, float* part, float* tot,
                                                     int tid, int c, int hh, int* status) {
    const float inv = 1.0f / (float)max(cnt, 1);
    const bool ragged = wave * 32 + 32 > cnt;       // wave-uniform: some of this wave's 32 nodes lie past the graph
    if (ragged) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[T][r] = wave * 32 + acc_row(r, hh) < cnt ? x[T][r] : 0.f;
    }
    float s[4], m[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) { a0 += x[T][r]; a1 += x[T][r + 1]; }
        s[T] = a0 + a1;
    }
    tile_t_total(s, part, tot, tid, wave, c, hh, m);
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const float mean = m[T] * inv;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            x[T][r] -= mean;
            x[T][r + 1] -= mean;
            if (ragged) {
                x[T][r] = wave * 32 + acc_row(r, hh) < cnt ? x[T][r] : 0.f;
                x[T][r + 1] = wave * 32 + acc_row(r + 1, hh) < cnt ? x[T][r + 1] : 0.f;
            }
            a0 = fmaf(x[T][r], x[T][r], a0);
            a1 = fmaf(x[T][r + 1], x[T][r + 1], a1);
        }
        s[T] = a0 + a1;
    }
    tile_t_total(s, part, tot, tid, wave, c, hh, m);
    // range sentinel: an activation beyond the fp16 range upstream (|Swish| > 1023) arrives here as inf / NaN statistics
    if (wave == 0 && !(m[0] + m[1] + m[2] + m[3] < 3.0e38f)) status_raise(status, MSMP_STATUS_NONFINITE);
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const float f = unit * msmp_rsq(m[T] * inv * unit * unit + eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[T][r] *= f;
    }
}

template <bool GATED>
__global__ __launch_bounds__(256, 2) void node_tail_split_kernel(TailArgs a) {
    // ONE LDS object: with several __shared__ arrays the compiler tags every access with its array and then makes each ds_read of an
    // array wait (vmcnt) for the LDS-DMA into that array issued just before it -- the next chunk's, i.e. the prefetch was waited for at once
    __shared__ __attribute__((aligned(16))) float smem[2 * SPLIT_CHUNK_FLOATS + 2 * ROWBUF_FLOATS + 9 * H + 8 * 128];
    float* const lds = smem;
    float* const rowbuf = smem + 2 * SPLIT_CHUNK_FLOATS;        // two staged 32-column chunks of the graph's node rows
    float* const part = rowbuf + 2 * ROWBUF_FLOATS;
    float* const tot = part + 8 * H;
    float* const xvl = tot + H;                                 // the graph's variables, [row 128][8] (zero past nv and past the graph)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int n0 = a.graph_ptr[blockIdx.x], n1 = a.graph_ptr[blockIdx.x + 1];
    const int cnt = n1 - n0;
    if (cnt <= 0) return;
    const long n = (long)n0 + wave * 32 + c;
    const long nc = n < n1 ? n : n1 - 1;

    PROF_DECL
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (visible to every wave behind the first head's first barrier)
        const int e = tid + 256 * i, row = e >> 3, f = e & 7;
        const float v = a.vars[(size_t)min(n0 + row, n1 - 1) * a.nv + min(f, a.nv - 1)];
        xvl[e] = v * (row < cnt && f < a.nv ? 1.0f : 0.f);        // (a product, not a select of the loaded value: the compiler turns that into a branch around the load, with the wait behind it)
    }
    f32x16 tau[4];
    if (GATED) {
        tail_rows_issue(a.h, a.agg[1], n0, n1, 0, rowbuf, tid);
        head_compute(rowbuf, n0, n1, a.h, a.agg[1], a.vars, nc, a.nv, a.b3[1], a.b4[1], a.w3vh[1], a.w3s[1], a.w4t[1], a.scales[1], lds, xvl, tid, lane,
                     c, hh, tau, [&] { tail_rows_issue(a.h, a.agg[0], n0, n1, 0, rowbuf, tid); }, true, tot PROF_PASS);
        PROF_MARK(0);
        tile_t_instance_norm(tau, wave, cnt, uniform_ro(a.scales[1], 7) * (1.0f / TAIL_ACT_SCALE), a.eps, part, tot, tid, c, hh, a.status);
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) tau[T][r] = sigmoidf_(tau[T][r]);
        PROF_MARK(1);
    } else {
        tail_rows_issue(a.h, a.agg[0], n0, n1, 0, rowbuf, tid);
    }
    f32x16 y[4];
    head_compute(rowbuf, n0, n1, a.h, a.agg[0], a.vars, nc, a.nv, a.b3[0], a.b4[0], a.w3vh[0], a.w3s[0], a.w4t[0], a.scales[0], lds, xvl, tid, lane, c, hh,
                 y, [] {}, GATED || a.mode == MSMP_LAYER_LIN, tot PROF_PASS);
    PROF_MARK(2);
    // this lane's piece of the transposed tiles: nodes n0 + 32 wave + acc_row(r, hh), channels 4 c .. 4 c + 3 (tile T = channel 4 c + T)
    const size_t base = ((size_t)n0 + wave * 32 + 4 * hh) * H + 4 * c;
    const int lim = cnt - wave * 32 - 4 * hh;          // register r is a node of the graph iff (r & 3) + 8 (r >> 2) < lim
    const bool full = wave * 32 + 32 <= cnt;           // wave-uniform: all 32 nodes of this wave belong to the graph
    const bool need_h = GATED || a.mode != MSMP_LAYER_LIN;
    // h for the residual / blend: all 16 loads are issued here, ahead of the norm's barriers, and unpredicated on the common path
    // (loads and stores predicated per row became separately waited-for blocks: 40 % of the kernel)
    f32x4 hx[16];
    if (need_h) {
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hx[r] = *reinterpret_cast<const f32x4*>(a.h + base + (size_t)((r & 3) + 8 * (r >> 2)) * H);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                hx[r] = row < lim ? *reinterpret_cast<const f32x4*>(a.h + base + (size_t)row * H) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    float unit = uniform_ro(a.scales[0], 7) * (1.0f / TAIL_ACT_SCALE);
    if (!GATED && a.mode != MSMP_LAYER_LIN) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[T][r] = hx[r][T] + swishf(y[T][r] * unit);
        unit = 1.0f;
    }
    tile_t_instance_norm(y, wave, cnt, unit, a.eps, part, tot, tid, c, hh, a.status);
    if (need_h) {       // range sentinel on the rows of the blend (checked HERE, behind the norm's barriers: right behind the loads it made them wait)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(fmaxf(fmaxf(fabsf(hx[r][0]), fabsf(hx[r][1])), fmaxf(fabsf(hx[r][2]), fabsf(hx[r][3]))), mx);
        if (mx > NODE_RANGE) status_raise(a.status, MSMP_STATUS_NODE_SATURATED);
    }
    PROF_MARK(3);
    if (GATED) {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[T][r] = fmaf(tau[T][r], swishf(y[T][r]) - hx[r][T], hx[r][T]);
    }
    if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *reinterpret_cast<f32x4*>(a.out + base + (size_t)((r & 3) + 8 * (r >> 2)) * H) = f32x4{y[0][r], y[1][r], y[2][r], y[3][r]};
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);
            if (row < lim) *reinterpret_cast<f32x4*>(a.out + base + (size_t)row * H) = f32x4{y[0][r], y[1][r], y[2][r], y[3][r]};
        }
    }
    PROF_MARK(4);
    // Fused decoder (models_gnn.py:1371-1375 on the last pair's output): the graph's h' rows have just been stored by this
    // workgroup; after the barrier (its release waits for the stores) they are read back past the L1 -- from the L2 they were written
    // to, not from HBM -- by the position-split decoder, 32 nodes per pass, eight lanes per node, the per-node tables in the dead weight /
    // row buffers (16 nodes each).  Same arithmetic as decoder_split_kernel: the same bits.
    if (a.dec.w1) {
        using G = DecSplit<25, 16, 3, 14>;
        static_assert(16 * 8 * G::LP <= 2 * SPLIT_CHUNK_FLOATS && 16 * 8 * G::LP <= 2 * ROWBUF_FLOATS, "decoder tables must fit the dead buffers");
        __syncthreads();
        const int q = tid & 7, nl = tid >> 3;
        float* mrow = (nl < 16 ? lds : rowbuf) + (nl & 15) * 8 * G::LP;
        for (int p = 0; p < cnt; p += 32) {
            const bool live = p + nl < cnt;
            const long nn = live ? (long)n0 + p + nl : (long)n1 - 1;
            decoder_split_node<25, 16, 3, 14, true>(a.out + (size_t)nn * H, mrow, q, live, nn, a.dec, [] { __syncthreads(); });
            __syncthreads();
        }
    }
    PROF_FLUSH
}

}  // namespace msmp

using namespace msmp;

extern int g_lem_split;
extern int g_lem_nodes;
extern int g_lem_tail;
extern int g_lem_share;
// the sticky range status is the fp16-split path's: the exact-fp32 kernels have no range to leave (they run the data the split path could not)
static int* split_status() { return msmp_tune_get("split") ? msmp::status_ptr() : nullptr; }
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
    EdgeArgs a{h, u, pos, vars, tgt, col, nullptr, (long)n_edges, (long)n_nodes, 0, 0, tw, nv, L.nc1,
               packed + L.w1, packed + L.w2, packed + L.w2s, packed + L.scales, packed + L.b1, packed + L.b2, nullptr, nullptr, msg_out, nullptr, split_status()};
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
        ProjSplitArgs sa{a, packed + L.w1t, packed + L.scales};
        hipLaunchKernelGGL(node_proj_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa);
    } else
        hipLaunchKernelGGL(node_proj_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_NODE_PROJ, (hipStream_t)stream);
    return check_launch("node_proj_kernel");
}

static int g_edge_nb = 0;    // tuning override (msmp_tune): 0 = automatic, 1 / 2 = force the tile size of the factorised kernel

static int g_pair = 1;       // gated pair: both heads' projection / message kernels in one launch each: 0 never, 1 up to PAIR_MAX_NODES nodes, 2 always
constexpr int64_t PAIR_MAX_NODES = 65536;     // measured (E2, ms per rollout step, per-head vs paired): 256 graphs 1.33 / 1.14, 512: 2.06 / 1.95, 1024: 3.63 / 3.65, 2048: 6.95 / 7.07
static int g_decoder = 1;     // 1-D decoder: 1 = eight lanes per node, split by position; 0 = one lane per node (decoder_kernel.hip)
static int g_tile_arith = 1;  // ranged tiles: slot -> node arithmetically (tile_halo) instead of through the node list
static int g_tile_align = 0;  // host layer: cut node tiles at graph boundaries also where tile_nodes does not divide the graph size (bitwise graph-order / sharding equivariance on knn graphs, ~11-20 % more tiles there)
static int g_tile = 2;       // node tiles (tile_kernels.hip): 2 fold the projections into the message kernel, 1 staged P / Q rows, 0 off
static int g_bwd_gemm = 1;   // layer backward: row GEMMs on rows_gemm_kernel (bf16x3 MFMA, fused epilogues); 0: rocblas_sgemm + separate passes
static int g_dec_fuse = 0;   // host layer: the 1-D decoder as the epilogue of the last layer's node tail (msmp_mp_layer_decode_f32); measured, not the default
static int g_tail = 1;       // fused node tail (msmp_node_tail_f32) inside msmp_mp_layer_f32; msmp_tune("tail", 0) chains the pieces
int msmp_tune_get(const char* key) {
    if (!strcmp(key, "split")) return g_split;
    if (!strcmp(key, "tail")) return g_tail;
    if (!strcmp(key, "bwd_gemm")) return g_bwd_gemm;
    if (!strcmp(key, "pair")) return g_pair;
    if (!strcmp(key, "tile")) return g_tile;
    if (!strcmp(key, "tile_arith")) return g_tile_arith;
    if (!strcmp(key, "tile_align")) return g_tile_align;
    if (!strcmp(key, "decoder")) return g_decoder;
    if (!strcmp(key, "dec_fuse")) return g_dec_fuse;
    if (!strcmp(key, "lem_tail")) return g_lem_tail;
    if (!strcmp(key, "lem_share")) return g_lem_share;
    return 0;
}

extern "C" int msmp_tune_query(const char* key) { return key ? msmp_tune_get(key) : 0; }

extern "C" int msmp_tune(const char* key, int value) {
    if (key && !strcmp(key, "tail")) { g_tail = value; return MSMP_OK; }
    if (key && !strcmp(key, "bwd_gemm")) { g_bwd_gemm = value; return MSMP_OK; }
    if (key && !strcmp(key, "pair")) { g_pair = value; return MSMP_OK; }
    if (key && !strcmp(key, "tile")) { g_tile = value; return MSMP_OK; }
    if (key && !strcmp(key, "tile_arith")) { g_tile_arith = value; return MSMP_OK; }
    if (key && !strcmp(key, "tile_align")) { g_tile_align = value; return MSMP_OK; }
    if (key && !strcmp(key, "decoder")) { g_decoder = value != 0; return MSMP_OK; }
    if (key && !strcmp(key, "dec_fuse")) { g_dec_fuse = value != 0; return MSMP_OK; }
    if (key && !strcmp(key, "lem_tail")) { g_lem_tail = value != 0; return MSMP_OK; }
    if (key && !strcmp(key, "lem_share") && value >= 1 && value <= 16) { g_lem_share = value; return MSMP_OK; }
    if (key && !strcmp(key, "edge_nb")) { g_edge_nb = value; return MSMP_OK; }
    if (key && !strcmp(key, "lem")) { g_lem_split = value; return MSMP_OK; }
    if (key && !strcmp(key, "lem_nodes")) { g_lem_nodes = value; return MSMP_OK; }
    if (key && !strcmp(key, "split")) { g_split = value; g_lem_split = value ? 4 : 0; return MSMP_OK; }
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
    EdgeArgs a{h, u, pos, vars, tgt, col, rowptr, (long)n_edges, (long)n_nodes, tile_nodes, 0, tw, nv, L.nc1,
               packed + L.w1, packed + L.w2, packed + L.w2s, packed + L.scales, packed + L.b1, packed + L.b2, P, Q, nullptr, agg_out, split_status()};
    const unsigned grid = (unsigned)((n_nodes + tile_nodes - 1) / tile_nodes);
    timing_begin(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    if (P && edges_per_tile == 128 && g_split) hipLaunchKernelGGL((edge_mlp_kernel_occ2<1, true, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (P && edges_per_tile == 128) hipLaunchKernelGGL((edge_mlp_kernel<1, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (P) hipLaunchKernelGGL((edge_mlp_kernel<2, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((edge_mlp_kernel<2, true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_EDGE_MLP, (hipStream_t)stream);
    return check_launch("edge_mlp_kernel<fused mean>");
}

// Projection + message + mean of BOTH heads of a gated pair with one launch per stage (library-internal; msmp_mp_layer_f32).
// Returns MSMP_ERR_UNSUPPORTED (without setting an error text) when the configuration is not the default kernel path or the
// batch is large: the caller then issues the per-head launches.  Same kernels bodies, bit-identical results.
int msmp_pair_project_aggregate(const float* h, const float* u, const float* pos, const float* vars, const int32_t* rowptr,
                                const int32_t* col, const int32_t* tgt, int64_t n_nodes, int64_t n_edges, int max_in_degree, int tw,
                                int nv, const float* packed_a, const float* packed_b, float* p_a, float* q_a, float* p_b, float* q_b,
                                float* agg_a, float* agg_b, msmp_stream_t stream) {
    if (!g_pair || (g_pair == 1 && n_nodes > PAIR_MAX_NODES)) return MSMP_ERR_UNSUPPORTED;
    if (!g_split || g_edge_nb == 2) return MSMP_ERR_UNSUPPORTED;
    if (n_edges <= 0 || max_in_degree <= 0 || max_in_degree > 128 || n_edges >= (1L << 31) || n_nodes >= (1L << 31)) return MSMP_ERR_UNSUPPORTED;
    const PackedLayout L = packed_layout(tw, nv);
    hipStream_t st = (hipStream_t)stream;
    ProjSplitArgs2 pa;
    const float* packed[2] = {packed_a, packed_b};
    float* pp[2] = {p_a, p_b};
    float* qq[2] = {q_a, q_b};
    float* agg[2] = {agg_a, agg_b};
    for (int i = 0; i < 2; ++i)
        pa.head[i] = ProjSplitArgs{ProjArgs{h, u, pos, vars, (long)n_nodes, tw, nv, L.nc1, packed[i] + L.w1, packed[i] + L.b1, pp[i], qq[i]},
                                   packed[i] + L.w1t, packed[i] + L.scales};
    timing_begin(MSMP_K_NODE_PROJ, st);
    hipLaunchKernelGGL(node_proj_split_pair_kernel, dim3((unsigned)((n_nodes + 127) / 128), 2), dim3(256), 0, st, pa);
    timing_end(MSMP_K_NODE_PROJ, st);
    int tile_nodes = 128 / max_in_degree;
    if (tile_nodes > 256) tile_nodes = 256;
    EdgeArgs2 ea;
    for (int i = 0; i < 2; ++i)
        ea.head[i] = EdgeArgs{nullptr, nullptr, nullptr, nullptr, tgt, col, rowptr, (long)n_edges, (long)n_nodes, tile_nodes, 0, tw, nv, L.nc1,
                              packed[i] + L.w1, packed[i] + L.w2, packed[i] + L.w2s, packed[i] + L.scales, packed[i] + L.b1,
                              packed[i] + L.b2, pp[i], qq[i], nullptr, agg[i], split_status()};
    timing_begin(MSMP_K_EDGE_MLP, st);
    hipLaunchKernelGGL(edge_mlp_pair_kernel_occ2, dim3((unsigned)((n_nodes + tile_nodes - 1) / tile_nodes), 2), dim3(256), 0, st, ea);
    timing_end(MSMP_K_EDGE_MLP, st);
    return check_launch("edge_mlp_pair_kernel_occ2");
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

#if MSMP_PROF
extern "C" __attribute__((visibility("default"))) int msmp_debug_prof(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int msmp_node_tail_f32(const float* h, const float* agg_main, const float* agg_gate, const float* vars,
                                  const int32_t* graph_ptr, int64_t n_nodes, int64_t n_graphs, int max_graph_nodes, int nv,
                                  const float* packed_main, const float* packed_gate, int mode, float eps, float* out,
                                  msmp_stream_t stream) {
    return msmp_node_tail_impl(h, agg_main, agg_gate, vars, graph_ptr, n_nodes, n_graphs, max_graph_nodes, nv, packed_main, packed_gate, mode,
                               eps, out, nullptr, stream);
}

// The same with the 1-D decoder as the launch's epilogue (library-internal: msmp_mp_layer_decode_f32)
int msmp_node_tail_impl(const float* h, const float* agg_main, const float* agg_gate, const float* vars, const int32_t* graph_ptr,
                        int64_t n_nodes, int64_t n_graphs, int max_graph_nodes, int nv, const float* packed_main, const float* packed_gate,
                        int mode, float eps, float* out, const msmp_decoder_t* dec, msmp_stream_t stream) {
    MSMP_REQUIRE(h && agg_main && vars && graph_ptr && packed_main && out, MSMP_ERR_ARG, "msmp_node_tail_f32: null pointer");
    MSMP_REQUIRE((agg_gate != nullptr) == (packed_gate != nullptr), MSMP_ERR_ARG, "msmp_node_tail_f32: give both gate arguments or none");
    MSMP_REQUIRE(n_nodes > 0 && n_graphs > 0 && n_graphs < (1L << 31) && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG,
                 "msmp_node_tail_f32: bad sizes");
    MSMP_REQUIRE(mode == MSMP_LAYER_LIN || mode == MSMP_LAYER_RESIDUAL_SWISH, MSMP_ERR_ARG, "msmp_node_tail_f32: bad mode %d", mode);
    MSMP_REQUIRE(!packed_gate || mode == MSMP_LAYER_LIN, MSMP_ERR_ARG, "msmp_node_tail_f32: the gated pair uses GNN_LayerLin layers");
    MSMP_REQUIRE(out != h, MSMP_ERR_ARG, "msmp_node_tail_f32: out may not alias h");
    MSMP_REQUIRE(max_graph_nodes > 0 && max_graph_nodes <= 128, MSMP_ERR_UNSUPPORTED,
                 "msmp_node_tail_f32: graphs of up to 128 nodes (got max_graph_nodes=%d); use the piecewise entry points", max_graph_nodes);
    MSMP_REQUIRE(g_split, MSMP_ERR_UNSUPPORTED, "msmp_node_tail_f32: only on the fp16-split matrix path");
    const PackedLayout L = packed_layout(1, nv);   // w3/w4/b3/b4/w3v/w3s/scales offsets do not depend on tw
    const float* pg = packed_gate ? packed_gate : packed_main;
    TailArgs a{h, {agg_main, agg_gate}, vars, graph_ptr, nv, mode, eps,
               {packed_main + L.b3, pg + L.b3}, {packed_main + L.b4, pg + L.b4}, {packed_main + L.w3vh, pg + L.w3vh},
               {packed_main + L.w3s, pg + L.w3s}, {packed_main + L.w4t, pg + L.w4t}, {packed_main + L.scales, pg + L.scales}, out, split_status(), DecW{}};
    if (dec) {
        MSMP_REQUIRE(dec->w1 && dec->b1 && dec->w2 && dec->b2 && dec->out, MSMP_ERR_ARG, "msmp_mp_layer_decode_f32: null pointer in the decoder description");
        MSMP_REQUIRE(dec->time_window == 25, MSMP_ERR_UNSUPPORTED, "msmp_mp_layer_decode_f32: the fused decoder is built for time_window 25 (got %d)", dec->time_window);
        a.dec = DecW{dec->w1, dec->b1, dec->w2, dec->b2, dec->u, dec->dt, dec->out};
    }
    timing_begin(MSMP_K_NODE_UPDATE, (hipStream_t)stream);
    if (packed_gate) hipLaunchKernelGGL(node_tail_split_kernel<true>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(node_tail_split_kernel<false>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, a);
    timing_end(MSMP_K_NODE_UPDATE, (hipStream_t)stream);
    return check_launch("node_tail_split_kernel");
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
