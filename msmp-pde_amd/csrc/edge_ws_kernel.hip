// Weight-stationary message + mean kernel (rows L1 + L2, factorised message_net_1, fp16-split matrix path).
//
//   agg[i] = mean_{j in N(i)} Swish(W2 Swish(P[i] + Q[j]) + b2)          (experiments/models_gnn.py:69-78, 132-141, aggr='mean')
//
// message_net_2 (128 x 128, fp16 hi + lo = 64 KB) stays on the CU for the whole kernel: every wave keeps the hi halves in its
// registers (128 of the 512 a lone wave owns), the lo halves sit once per workgroup in LDS (32 KB).  A 256-thread workgroup is
// four independent waves that loop over 32-edge blocks: no weight staging and no barrier after the prologue.  A block is `npb` consecutive target nodes (npb * max
// in-degree <= 32) with their in-edges (CSR order), one edge per lane pair (c, hh):
//   * the GEMM is computed TRANSPOSED (A = Swish(P_i + Q_j) fragments, one edge per lane, k = 16 s + 8 hh + j; B = the
//     stationary W2 fragments), so the accumulator holds [32 edges of the block][channel = lane]; the w2t copy deals the
//     output channels round-robin over the four tiles (tile T, lane c = channel 4 c + T), so a lane owns four
//     consecutive channels;
//   * the mean over the in-edges of each target is ONE MORE matrix product on that tile: agg = S M with the 0/1 matrix
//     S[target][edge] (exact in fp16) as A operand and the message tile (hi + lo halves, k = edge in accumulator order,
//     straight from registers) as B operand, scaled by 1 / deg afterwards; zero in-degree nodes get an all-zero row;
//   * the index loads and the first P/Q pieces of the next block are in flight while the current block computes.
// Same arithmetic as the streamed-weight edge kernel up to summation order (the mean is formed from hi + lo halves of the
// messages: 2^-22 relative).  Used when max in-degree <= 32; larger degrees take the streamed kernel.
#include "mfma_tiles.h"

#if MSMP_PROF_EDGE      // phase counters (wave 0 of every workgroup; scripts/prof_edge.py)
extern __device__ unsigned long long g_prof_ws[16];
__device__ unsigned long long g_prof_ws[16];
#define EWS_DECL long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tp = __builtin_readcyclecounter();
#define EWS_MARK(i) do { const long long t_ = __builtin_readcyclecounter(); pacc[i] += t_ - tp; tp = t_; } while (0)
#define EWS_FLUSH if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_prof_ws[i_], (unsigned long long)pacc[i_]);
#else
#define EWS_DECL
#define EWS_MARK(i)
#define EWS_FLUSH
#endif

namespace msmp {

struct EdgeWsArgs {
    const float* P;        // [N,128]
    const float* Q;        // [N,128]
    const int* rowptr;     // [N+1]
    const int* col;        // [E] source of CSR edge
    const int* tgt;        // [E] target of CSR edge
    long n_nodes, n_blocks;
    int npb;               // nodes per block
    const float* w2t;      // 4 split chunks, natural k order, rows dealt round-robin (packed_layout().w2t)
    const float* b2;       // [128]
    const float* scales;   // [8]
    float* agg;            // [N,128]
};

// pieces of k-chunk t for this lane's edge: floats 32 t + 16 s + 8 hh .. + 7 of a [N,128] row
__device__ __forceinline__ void ews_row_piece(const float* __restrict__ base, int node, int hh, int t, f32x4 (&dst)[4]) {
    const float* p = base + (size_t)node * H + 32 * t + 8 * hh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dst[2 * s] = *reinterpret_cast<const f32x4*>(p + 16 * s);
        dst[2 * s + 1] = *reinterpret_cast<const f32x4*>(p + 16 * s + 4);
    }
}

struct EwsBlock {           // one 32-edge block: npb consecutive targets and their in-edges
    int rp0, rp1;           // rowptr[n_a + c], rowptr[n_a + c + 1] (clamped to the last entry)
    int cnt, e0, e1;        // nodes of the block, its CSR edge range
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void edge_ws_kernel(EdgeWsArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long gw = (long)blockIdx.x * WAVES + wave, nw = (long)gridDim.x * WAVES;

    // stationary operand: message_net_2 as hi / lo B fragments, once per workgroup in LDS (64 KB), read-only after the prologue
    __shared__ half8 wf[4 * 1024];          // [k chunk t][k step s][channel tile T][hi, lo][lane]
    {
        const half8* ws = reinterpret_cast<const half8*>(a.w2t);
        for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) wf[i] = ws[i];
    }
    __syncthreads();
    if (gw >= a.n_blocks) return;
    const float sc2 = a.scales[1], inv2 = a.scales[5];
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b2 + 4 * c) * sc2;

    // The wave has its SIMD to itself, so every load must be issued a whole block ahead of its use.  Pipeline per wave
    // (block index i counts this wave's blocks gw + i nw):  rowptr words of i + 3 -> tgt / col of i + 2 -> Q rows of
    // i + 1 (all four k-chunks, each reissued as soon as the previous block's chunk has been consumed) -> compute i.
    // P rows are shared by the in-edges of a target (L1 hits after the first lane): fetched one k-chunk ahead.
    auto load_rp = [&](long blk, EwsBlock& m) {
        m.rp0 = 0; m.rp1 = 0; m.cnt = 0; m.e0 = 0; m.e1 = 0;
        if (blk < a.n_blocks) {
            const long n_a = blk * a.npb;
            m.rp0 = a.rowptr[min(n_a + c, a.n_nodes)];
            m.rp1 = a.rowptr[min(n_a + c + 1, a.n_nodes)];
            m.cnt = (int)min((long)a.npb, a.n_nodes - n_a);
        }
    };
    auto finish_rp = [&](EwsBlock& m) {      // edge range once the rowptr words have arrived
        if (m.cnt > 0) {
            m.e0 = __builtin_amdgcn_readfirstlane(m.rp0);
            m.e1 = __builtin_amdgcn_readlane(m.rp1, m.cnt - 1);
        }
    };
    auto load_idx = [&](const EwsBlock& m, int& ni, int& nj) {
        const int ec = m.e1 == m.e0 ? 0 : min(m.e0 + c, m.e1 - 1);      // lanes past the block repeat its last edge
        ni = a.tgt[ec];
        nj = a.col[ec];
    };

    EwsBlock m0, m1, m2, m3;        // blocks i, i + 1, i + 2, i + 3
    int ni0, nj0, ni1, nj1, ni2 = 0, nj2 = 0;
    load_rp(gw, m0); load_rp(gw + nw, m1); load_rp(gw + 2 * nw, m2); load_rp(gw + 3 * nw, m3);
    finish_rp(m0); finish_rp(m1); finish_rp(m2);
    load_idx(m0, ni0, nj0); load_idx(m1, ni1, nj1); load_idx(m2, ni2, nj2);
    f32x4 qc[4], pc[4];        // Q / P pieces of the next k-chunk (the other wave of the SIMD covers their latency)
    ews_row_piece(a.Q, nj0, hh, 0, qc);
    ews_row_piece(a.P, ni0, hh, 0, pc);

    EWS_DECL
    for (long blk = gw; blk < a.n_blocks; blk += nw) {
        EWS_MARK(6);
        const long n_a = blk * a.npb;
        const int cnt = m0.cnt;
        // advance the index pipeline: m3's rowptr words (issued one block ago) -> its edge range; tgt / col of block i + 3
        // are issued at the bottom, after the rotation
        finish_rp(m3);
        const bool edge_valid = m0.e0 + c < m0.e1;
        const int tl = edge_valid ? ni0 - (int)n_a : -1;                // local target of this lane's edge
        const float invdeg = 1.0f / (float)max(m0.rp1 - m0.rp0, 1);     // of node n_a + c

        // ---- message_net_2, transposed: yT[T][r] = 2^s2 (W2 Swish(P_i + Q_j) + b2)[channel 4 c + T] of edge acc_row(r, hh)
        f32x16 yT[4];
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) yT[T][r] = bias[T];
        EWS_MARK(0);
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            float z[16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < 4; ++m) z[4 * q + m] = swishf(pc[q][m] + qc[q][m]);
            if (t < 3) {
                ews_row_piece(a.Q, nj0, hh, t + 1, qc);
                ews_row_piece(a.P, ni0, hh, t + 1, pc);
            } else {                                                    // first pieces of the next block
                ews_row_piece(a.Q, nj1, hh, 0, qc);
                ews_row_piece(a.P, ni1, hh, 0, pc);
            }
            EWS_MARK(1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = z[8 * s + j];
                half8 zhi, zlo;
                split8(v, zhi, zlo);
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    const half8 whi = wf[t * 1024 + ((s * 4 + T) * 2 + 0) * 64 + lane], wlo = wf[t * 1024 + ((s * 4 + T) * 2 + 1) * 64 + lane];
                    yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi, wlo, yT[T], 0, 0, 0);
                    yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zlo, whi, yT[T], 0, 0, 0);
                    yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi, whi, yT[T], 0, 0, 0);
                }
            }
            EWS_MARK(2);
        }

        // ---- S[target row = lane & 31][edge k]: k = 16 s + 8 (j >> 2) + 4 hh + (j & 3) (the accumulator row order)
        half8 S[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
                const int tk = __builtin_amdgcn_ds_bpermute(4 * k, tl);
                S[s][j] = tk == c ? (_Float16)1.0f : (_Float16)0.0f;
            }
        // 1 / deg of the rows this lane stores (node n_a + (r & 3) + 8 (r >> 2) + 4 hh): fetched before the matrix work so
        // that the cross-lane latency is not paid once per store
        float fr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = (r & 3) + 8 * (r >> 2);
            fr[r] = 0.f;
            if (row0 < cnt) fr[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (row0 + 4 * hh), __builtin_bit_cast(int, invdeg)));
        }
        EWS_MARK(3);
        // ---- mean: D[T] = S (Swish(yT[T]) as hi + lo fragments), rows = local targets.  Straight-line over the four channel
        // tiles (no stores in between), so the Swish / split of one tile overlaps the matrix work of the previous one.
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x16 m;
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = swishf(yT[T][r] * inv2);
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = m[8 * s + j];
                half8 mhi, mlo;
                split8(v, mhi, mlo);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(S[s], mlo, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(S[s], mhi, d, 0, 0, 0);
            }
            yT[T] = d;              // the tile's registers now hold its aggregated rows
        }
        EWS_MARK(4);
        float* o = a.agg + ((size_t)n_a + 4 * hh) * H + 4 * c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = (r & 3) + 8 * (r >> 2);
            if (row0 < cnt) {                                          // wave-uniform
                const f32x4 v = f32x4{yT[0][r], yT[1][r], yT[2][r], yT[3][r]} * fr[r];
                if (row0 + 4 * hh < cnt) *reinterpret_cast<f32x4*>(o + (size_t)row0 * H) = v;
            }
        }
        EWS_MARK(5);
        // rotate the pipeline
        m0 = m1; m1 = m2; m2 = m3;
        ni0 = ni1; nj0 = nj1; ni1 = ni2; nj1 = nj2;
        load_idx(m2, ni2, nj2);
        load_rp(blk + 4 * nw, m3);
    }
    EWS_FLUSH
}

}  // namespace msmp

using namespace msmp;

#if MSMP_PROF_EDGE
extern "C" __attribute__((visibility("default"))) int msmp_debug_prof_ws(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_ws), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof_ws), 16 * sizeof(unsigned long long));
}
#endif

int g_edge_ws_waves = 8;        // msmp_tune("edge_ws_waves"): waves per workgroup = per CU (8, 12, 16)
// launcher used by edge_aggregate (mlp_kernels.hip)
int msmp_launch_edge_ws(const float* P, const float* Q, const int32_t* rowptr, const int32_t* col, const int32_t* tgt, int64_t n_nodes,
                        int max_in_degree, const float* w2t, const float* b2, const float* scales, float* agg, hipStream_t stream) {
    const int npb = max_in_degree > 0 ? 32 / max_in_degree : 32;
    const long n_blocks = (n_nodes + npb - 1) / npb;
    EdgeWsArgs a{P, Q, rowptr, col, tgt, (long)n_nodes, n_blocks, npb, w2t, b2, scales, agg};
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
    }
    const int waves = g_edge_ws_waves == 12 ? 12 : g_edge_ws_waves == 16 ? 16 : 8;
    const long want = (n_blocks + waves - 1) / waves;
    const unsigned grid = (unsigned)(want < cus ? want : cus);     // one workgroup per CU: each wave keeps W2 in its registers
    if (waves == 16) hipLaunchKernelGGL(edge_ws_kernel<16>, dim3(grid), dim3(1024), 0, stream, a);
    else if (waves == 12) hipLaunchKernelGGL(edge_ws_kernel<12>, dim3(grid), dim3(768), 0, stream, a);
    else hipLaunchKernelGGL(edge_ws_kernel<8>, dim3(grid), dim3(512), 0, stream, a);
    return 0;
}
