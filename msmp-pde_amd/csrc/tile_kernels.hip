// LDS-staged node tiles for the message kernel (rows L1 + L2; north_star "LDS staging of node tiles for edge gather").
//
// The reference gathers x_i = x[edge_index[1]], x_j = x[edge_index[0]] per EDGE (PyG propagate, experiments/models_gnn.py:65,128).
// Here a workgroup owns a tile of `tile_nodes` consecutive target nodes with all their in-edges (<= 128) and the <= 32 distinct
// nodes those edges touch (msmp_tiles_t, built once per graph structure by msmp_build_tiles).  Each node row of the tile is
// loaded ONCE, coalesced, and every edge then reads its two operands from LDS:
//   FOLD = false: the tile's P / Q rows (msmp_node_project_f32) are staged, [slot][128] fp32, row stride 132 floats
//                 (conflict-free ds_read_b128: a 16-lane group reads one 16-byte piece of 16 different or equal rows);
//   FOLD = true : the tile's h rows and [u, pos, vars] columns are staged as fp16 hi/lo fragments, P and Q of the tile's
//                 nodes are computed in the workgroup (one 32-node MFMA block; wave w owns output channels 32w..32w+31 of
//                 both, its weight fragments come straight from the L2-resident packed blob) and written to the same LDS
//                 rows: message_net_1's per-node projections never touch HBM, halo nodes are recomputed per tile.
// After that the kernel is the factorised message kernel of mlp_kernels.hip: Swish(P_i + Q_j) formed in the B-operand registers
// of message_net_2 (fp16-split MFMA, weights streamed through LDS), Swish, per-target mean in CSR order through LDS.
#include "mfma_tiles.h"

#define TILE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)

namespace msmp {

// Phase profile (build with MSMP_PROF=tile in the environment of build.py; scripts/prof_tile.py reads it): cycle sums of wave 0 of
// every workgroup per phase, kept in registers and added to g_prof_tile once at the end.
#if MSMP_PROF_TILE
__device__ unsigned long long g_prof_tile[16];
#define TPROF_DECL unsigned pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned tp = (unsigned)__builtin_readcyclecounter();
#define TPROF(i) do { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); pacc[i] += t_ - tp; tp = t_; } while (0)
// one workgroup in 64 reports (every workgroup adding to the same 12 words turns the counters' L2 channel into a hot spot that
// slows the weight loads of ALL workgroups: the profile of round 2's first edition showed 2-3 x inflated load phases)
#define TPROF_FLUSH if (tid == 0 && (blockIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_prof_tile[i_], (unsigned long long)pacc[i_]); atomicAdd(&g_prof_tile[15], 1ull); }
#else
#define TPROF_DECL
#define TPROF(i)
#define TPROF_FLUSH
#endif

constexpr int TILE_NCAP = MSMP_TILE_NCAP;
constexpr int TILE_EDGES = MSMP_TILE_EDGES;
constexpr int PQLD = H + 4;                       // LDS row stride of a staged P / Q row (floats): 33 x 16 B
constexpr int BROW_T = 72;                        // halfs per staged fragment row: [hi 32 | lo 32] + 16 B pad (as node_proj's B tile)
// chunk c of the fragment tile starts FRAG_SKEW halfs (16 banks) later than a plain [chunk][slot] array would put it: a thread's 16-lane
// write group covers the 32-byte pieces of TWO consecutive chunks of one slot (32 rows = 1 152 dwords apart: the same banks without the skew)
constexpr int FRAG_SKEW = 32;
__device__ __forceinline__ int frag_row(int chunk, int slot) { return (chunk * 32 + slot) * BROW_T + chunk * FRAG_SKEW; }
// Activations enter the fp16-split GEMMs multiplied by 2^6.  The low half of a value x is fp16(x - fp16(x)) ~ 2^-11 x: for
// |x| < 0.25 it falls into the fp16 subnormals (quantum 6e-8), i.e. the split then carries x with an ABSOLUTE error of 3e-8
// instead of a relative 2^-23.  Hidden states right behind the encoder and the pre-activations of the first layers are that
// small, and the InstanceNorm that follows divides by their (equally small) per-graph spread (measured: layer pair 0 of E2, 2.4x
// the error of a float32 evaluation; scripts/diag_layer.py).  With 2^6 every |x| > 4e-3 keeps a normal low half; all factors
// are powers of two folded into constants that were there anyway (no extra instruction); the fp16 range then covers |x| < 1023 (2^8 was measured too: 3.2x instead of 3.6x the float32 floor on the worst case, not worth the lost range).
constexpr float ACT_SCALE = 64.0f;
// The NODE rows (hidden state and the [u | pos | vars] columns) take 2^8: hidden states are bounded by the InstanceNorm that made
// them (|h| <= sqrt(nodes per graph - 1) < 11.4 for graphs of up to 128 nodes), and the staging (per node, not per edge) can
// afford one v_med3 per value that saturates instead of overflowing (|x| > 255 saturates; irrelevant for PDE data of order one).
constexpr float NODE_SCALE = 256.0f;
constexpr float NODE_MAX = 65504.0f;
__device__ __forceinline__ float node_scaled(float x) { return __builtin_amdgcn_fmed3f(x * NODE_SCALE, -NODE_MAX, NODE_MAX); }

// ------------------------------------------------------------------------------------------------------------------------
// Tile metadata: one 128-thread workgroup per tile.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void build_tiles_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n_nodes,
                                                          int tile_nodes, int group_nodes, int* __restrict__ tile_node,
                                                          int* __restrict__ tile_count, int* __restrict__ tile_halo,
                                                          int* __restrict__ edge_slot, int* __restrict__ stats) {
    __shared__ int s_col[TILE_EDGES];       // source of the edge at lane k, -1: no edge at this lane
    __shared__ int s_first[TILE_EDGES];     // 1: first occurrence of an out-of-range source
    __shared__ int s_slot[TILE_EDGES];
    __shared__ int s_halo[5];               // lo start, lo count, hi start, hi count, ranged
    __shared__ int s_stat[6];
    const int t = blockIdx.x, k = threadIdx.x;
    const int n0 = t * tile_nodes, n1 = min(n0 + tile_nodes, n_nodes);
    const int nt = n1 - n0;
    // Wave groups: wave g of the message kernel owns the targets [gf, gl) and ALL their in-edges, at lanes 32 g ...; a target's
    // edges never straddle two waves, so the per-target sum is wave-local (one more MFMA on the message tile, no LDS staging).
    const int g = k >> 5, kk = k & 31;
    const int gf = min(n0 + g * group_nodes, n1), gl = min(gf + group_nodes, n1);
    const int ge0 = rowptr[gf], ge1 = rowptr[gl];
    const bool valid = kk < ge1 - ge0 && kk < 32;
    const int e = ge0 + kk;
    int j = -1;
    if (valid) j = col[e];
    s_col[k] = j;
    __syncthreads();
    int first = 0, owner = k;
    if (valid && (j < n0 || j >= n1)) {
        first = 1;
        for (int i = 0; i < k; ++i)
            if (s_col[i] == j) { first = 0; owner = i; break; }
    }
    s_first[k] = first;
    if (k < 6) s_stat[k] = (k == 0 || k == 3) ? 0x7fffffff : (k == 1 || k == 4) ? -1 : 0;     // lo min, lo max, lo count, hi min, hi max, hi count
    __syncthreads();
    // "Ranged" tile: the sources outside the tile form at most one run of consecutive nodes below it and one above it (every
    // interior tile of a banded 1-D graph).  Their slots are then numbered in ascending node order and the message kernel
    // computes the node of a slot ARITHMETICALLY from four integers it reads with one scalar load (tile_halo), instead of reading
    // the node list first and the rows second (two dependent global reads at the head of every tile).
    if (first) {
        int* st3 = s_stat + (j < n0 ? 0 : 3);
        atomicMin(st3, j);
        atomicMax(st3 + 1, j);
        atomicAdd(st3 + 2, 1);
    }
    __syncthreads();
    if (k == 0) {
        const int lo_min = s_stat[0], lo_max = s_stat[1], lo_cnt = s_stat[2], hi_min = s_stat[3], hi_max = s_stat[4], hi_cnt = s_stat[5];
        const bool ranged = (lo_cnt == 0 || lo_max - lo_min + 1 == lo_cnt) && (hi_cnt == 0 || hi_max - hi_min + 1 == hi_cnt);
        s_halo[0] = lo_cnt ? lo_min : 0; s_halo[1] = lo_cnt; s_halo[2] = hi_cnt ? hi_min : 0; s_halo[3] = hi_cnt; s_halo[4] = ranged;
    }
    __syncthreads();
    const bool ranged = s_halo[4] != 0;
    int slot = 0;
    if (valid) {
        if (j >= n0 && j < n1) slot = j - n0;
        else if (ranged) slot = j < n0 ? nt + (j - s_halo[0]) : nt + s_halo[1] + (j - s_halo[2]);
        else if (first) {
            int before = 0;
            for (int i = 0; i < k; ++i) before += s_first[i];
            slot = nt + before;
        }
    }
    s_slot[k] = slot;
    __syncthreads();
    if (valid && !(j >= n0 && j < n1) && !first && !ranged) slot = s_slot[owner];
    // outputs
    if (valid) {
        // target slot of edge e: the CSR row it lies in (rows of a group are short: linear search over <= group_nodes rows)
        int ts = gf - n0;
        while (ts + 1 < gl - n0 && rowptr[n0 + ts + 1] <= e) ++ts;
        edge_slot[(size_t)t * TILE_EDGES + k] = ts | (min(slot, 255) << 8);
        if (first && slot < TILE_NCAP) tile_node[(size_t)t * TILE_NCAP + slot] = j;
    } else
        edge_slot[(size_t)t * TILE_EDGES + k] = 0;           // lanes without an edge read a valid slot pair; their messages are never summed
    if (k < TILE_NCAP && k < nt) tile_node[(size_t)t * TILE_NCAP + k] = n0 + k;
    __syncthreads();
    const int total = nt + s_stat[2] + s_stat[5];           // targets + distinct outside sources
    if (k == 0) {
        tile_count[t] = min(total, TILE_NCAP);
        const bool use = ranged && total <= TILE_NCAP;
        tile_halo[4 * t + 0] = use ? s_halo[0] : 0;
        tile_halo[4 * t + 1] = use ? s_halo[1] : -1;       // -1: not ranged, the node list decides
        tile_halo[4 * t + 2] = use ? s_halo[2] : 0;
        tile_halo[4 * t + 3] = use ? s_halo[3] : 0;
        atomicMax(&stats[0], total);
        if (!use) atomicAdd(&stats[2], 1);                    // tiles that need the node list (not two runs): the launch then takes the LISTED kernels
    }
    if (kk == 0) atomicMax(&stats[1], ge1 - ge0);             // edges of one wave group: must fit 32 lanes
    // unused slots repeat the first node (valid addresses for unconditional loads): slots >= total were never written above
    __syncthreads();
    if (k < TILE_NCAP && k >= total) tile_node[(size_t)t * TILE_NCAP + k] = n0;
}

// Per-node rows of the columns of message_net_1 that are not hidden state: [u (tw) | pos | vars (nv) | 0 ...], padded to whole
// 32-column chunks.  They do not change over the layers of a forward, so they are packed once and every layer's tile staging
// reads them with 16-byte loads (instead of tw + 1 + nv scalar loads with index arithmetic per node and layer).
__global__ __launch_bounds__(256) void pack_features_kernel(const float* __restrict__ u, const float* __restrict__ pos,
                                                            const float* __restrict__ vars, long n, int tw, int nv, int stride,
                                                            float* __restrict__ out, int* status) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * stride) return;
    const long node = i / stride;
    const int k = (int)(i - node * stride);
    float v = 0.f;
    if (k < tw) v = u[node * tw + k];
    else if (k == tw) v = pos[node];
    else if (k <= tw + nv) v = vars[node * nv + (k - tw - 1)];
    out[i] = v;
    if (out_of_range(v)) status_raise(status, MSMP_STATUS_INPUT_RANGE);
}

// Feature preparation of a forward in ONE launch (experiments/models_gnn.py:1325-1352, models_gnn2D.py:104-116): from the graph's
// x [N,Tw], pos [N,2] = (t, x) and the per-node parameter columns it writes u = float(x), pos_x = float(pos[:,1] / L),
// pos_t = float(pos[:,0] / tmax), variables = float([pos_t | col_k / div_k]) and (optionally) the packed feature rows above.
// Divisions are done in the INPUT's dtype (float64 graphs: float64 division, then rounded to float32), exactly what the
// PyTorch expressions they replace did (~20 small launches per forward before).
struct PrepArgs {
    const void* x;
    const void* pos;
    const void* col[MSMP_MAX_VARS];
    double col_div[MSMP_MAX_VARS];
    int col_f64[MSMP_MAX_VARS];
    int x_f64, pos_f64, n_cols;
    double L, tmax;
    long n;
    int tw, stride;            // stride = 32 * tail chunks >= tw + 1 + (1 + n_cols)
    float *u, *pos_x, *pos_t, *vars, *feat;
    int* status;
};
__device__ __forceinline__ float prep_load(const void* p, int f64, long i) {
    return f64 ? (float)reinterpret_cast<const double*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ float prep_div(const void* p, int f64, long i, double d) {
    return f64 ? (float)(reinterpret_cast<const double*>(p)[i] / d) : reinterpret_cast<const float*>(p)[i] / (float)d;
}
__global__ __launch_bounds__(256) void prepare_nodes_kernel(PrepArgs a) {
    // thread = (node, four consecutive columns of the packed row): one 16-byte store of the feature row per thread
    const int groups = a.stride >> 2;                                  // 8, 16 or 24
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n * groups) return;
    const long node = groups == 8 ? i >> 3 : (groups == 16 ? i >> 4 : i / groups);
    const int k0 = 4 * (int)(i - node * groups);
    const int nv = 1 + a.n_cols;
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    bool bad = false;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = k0 + m;
        float v = 0.f;
        if (k < a.tw) {
            v = prep_load(a.x, a.x_f64, node * a.tw + k);
            a.u[node * a.tw + k] = v;
        } else if (k == a.tw) {
            v = prep_div(a.pos, a.pos_f64, 2 * node + 1, a.L);
            a.pos_x[node] = v;
        } else if (k == a.tw + 1) {
            v = prep_div(a.pos, a.pos_f64, 2 * node, a.tmax);
            a.pos_t[node] = v;
            a.vars[node * nv] = v;
        } else if (k <= a.tw + nv) {
            const int c = k - a.tw - 2;
            v = prep_div(a.col[c], a.col_f64[c], node, a.col_div[c]);
            a.vars[node * nv + 1 + c] = v;
        }
        out[m] = v;
        bad |= out_of_range(v);
    }
    if (a.feat) *reinterpret_cast<f32x4*>(a.feat + node * a.stride + k0) = out;
    if (bad) status_raise(a.status, MSMP_STATUS_INPUT_RANGE);        // the split path saturates node rows at |x| = 255.87
}

// ------------------------------------------------------------------------------------------------------------------------
struct TileArgs {
    const float* h;        // FOLD
    const float* u;
    const float* pos;
    const float* vars;
    const float* feat;     // FOLD, optional: [N][32 * tail chunks] = [u | pos | vars | 0...] per node (msmp_pack_node_features_f32)
    const float* P;        // !FOLD
    const float* Q;
    const int* rowptr;
    const int* tile_node;
    const int* tile_halo;
    const int* edge_slot;
    long n_nodes, n_edges;
    int tile_nodes, group_nodes;      // group_nodes = tile_nodes / 4: targets of one wave
    int period_tiles, period_nodes;   // > 0: the descriptor arrays hold ONE period (a graph); tile t uses entry t mod period_tiles, node ids + (t div period_tiles) period_nodes
    unsigned period_magic;            // ceil(2^32 / period_tiles): t div period_tiles = umulhi(t, magic) (host checks n_tiles * period_tiles < 2^32)
    int tw, nv, nc1;
    const float* w1s;      // FOLD: nc1 split chunks, natural k order, fragment row (T, lane) = W1 row 32 T + lane
    const float* w2s;      // 4 split chunks (acc order)
    const float* scales;   // [8]
    const float* b1;
    const float* b2;
    float* agg;
    int* status;           // msmp_last_status word (or nullptr)
};

// ------------------------------------------------------------------------------------------------------------------------
// The tile body.  MODE 0: staged P / Q rows;  1: folded projections, features read from u / pos / vars;  2: folded, packed feature rows.
// LISTED: the structure has tiles whose halo is not two runs of consecutive nodes (knn on a scattered periodic grid): every slot's node
// comes from the tile's node list (one batch of loads ahead of the row loads).  Otherwise the node of a slot is ARITHMETIC in four
// integers of the tile (one 16-byte scalar load, cache-resident when the descriptor is periodic) and nothing is waited for ahead of
// the row loads: round 3's single instantiation decided per tile with a uniform branch, and the compiler's conservative
// s_waitcnt vmcnt(0) at each join serialised the four row loads of a thread behind each other and behind the weight prefetches
// (six dependent memory round trips at the head of every tile; profiles/r04b_*).
// Cut for THREE workgroups per CU (50 304 B of LDS, <= 168 registers):
//   * message_net_2's weights stream through LDS in HALF chunks (one K = 16 step: 8 KB), double-buffered, by LDS-DMA;
//   * the fp16 fragment tile of the folded projections overlays that buffer AND the head of the P / Q rows (dead until the
//     projections are done: one barrier more than a private region would need).
constexpr int WHALF_FLOATS = SPLIT_CHUNK_FLOATS / 2;                            // 8 KB: one K = 16 step of a split chunk
constexpr int WBUF_FLOATS = 2 * WHALF_FLOATS;
constexpr int TILE_MAIN_FLOATS = WBUF_FLOATS + 2 * TILE_NCAP * PQLD;      // 12 544 floats = 50 176 B
constexpr int TILE_LUT_FLOATS = 32;                                       // 16 entries x 8 B: nibble -> four fp16 0 / 1 (the mean's selection matrix)
constexpr int TILE_LDS_FLOATS = TILE_MAIN_FLOATS + TILE_LUT_FLOATS;       // 50 304 B: three workgroups per CU
static_assert((8 * 32 * BROW_T + 8 * FRAG_SKEW) * 2 <= TILE_MAIN_FLOATS * 4, "fragment tile of the folded projections must fit");

struct WHalf {
    f32x4 r[2];
};
__device__ __forceinline__ void whalf_load(WHalf& w, const float* __restrict__ half_chunk, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) w.r[i] = *reinterpret_cast<const f32x4*>(half_chunk + 4 * (tid + 256 * i));
}
__device__ __forceinline__ void whalf_store(const WHalf& w, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(buf + 4 * (tid + 256 * i)) = w.r[i];
}
// The same copy by LDS-DMA (global_load_lds_dwordx4: the lane-linear LDS image IS the fragment layout; no registers, no ds_write,
// the workgroup barrier's vmcnt(0) waits for it).
__device__ __forceinline__ void whalf_dma(const float* __restrict__ half_chunk, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(half_chunk + 4 * (tid + 256 * i)),
                                         (__attribute__((address_space(3))) void*)(buf + 4 * (tid + 256 * i)), 16, 0, 0);
}

using half4v = __attribute__((ext_vector_type(4))) _Float16;

template <int MODE, bool LISTED>
__device__ __forceinline__ void edge_tile_body(const TileArgs& a, float* lds) {
    constexpr bool FOLD = MODE != 0;
    float* wbuf = lds;                               // W2 half chunks (2 x 8 KB)
    float* pl = lds + WBUF_FLOATS;             // P rows [32][PQLD]
    float* ql = pl + TILE_NCAP * PQLD;               // Q rows [32][PQLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    TPROF_DECL
    constexpr bool packed_feat = MODE == 2;
    // tile -> (descriptor entry, node offset): a periodic descriptor holds one graph's tiles
    int tm = blockIdx.x, node_off = 0;
    long n_end = a.n_nodes;
    if (a.period_tiles > 0) {
        const int q = a.period_tiles == 1 ? (int)blockIdx.x : (int)__umulhi((unsigned)blockIdx.x, a.period_magic);
        tm = (int)blockIdx.x - q * a.period_tiles;
        node_off = q * a.period_nodes;
        n_end = (long)node_off + a.period_nodes;
    }
    const int tile_n0 = node_off + tm * a.tile_nodes;
    const int tile_n1 = (int)min((long)tile_n0 + a.tile_nodes, n_end);
    const int nt_ = tile_n1 - tile_n0;
    int h_lo = 0, h_nlo = 0, h_hi = 0, h_nhi = 0;
    if (!LISTED) {                                   // one 16-byte scalar load
        h_lo = a.tile_halo[4 * tm] + node_off; h_nlo = a.tile_halo[4 * tm + 1]; h_hi = a.tile_halo[4 * tm + 2] + node_off; h_nhi = a.tile_halo[4 * tm + 3];
    }
    auto node_of = [&](int slot) -> int {            // ranged tiles: targets | run below | run above | (unused slots: the first node)
        const int r = slot - nt_;
        int node = r < h_nlo + h_nhi ? h_hi + (r - h_nlo) : tile_n0;
        node = r < h_nlo ? h_lo + r : node;
        node = r < 0 ? tile_n0 + slot : node;
        return node;
    };

    // slot pair of this lane's edge: stored per TILE ([tile][128], zero at lanes without an edge), so the load depends on nothing but
    // the workgroup's index
    const int sl = a.edge_slot[(size_t)tm * TILE_EDGES + wave * 32 + c];
    // this wave's targets [gf, gl) and, for lane c < gl - gf, the lanes [er0, er0 + edeg) of this wave that hold target gf + c's in-edges
    // (consumed by the mean at the very end: requested here, off the critical path).  Only DIFFERENCES of rowptr are used, so a
    // periodic structure reads its first period's entries (the same few cache lines for every tile of the launch).
    const int gf = min(tile_n0 + wave * a.group_nodes, tile_n1), gl = min(gf + a.group_nodes, tile_n1);
    const int* rp = a.rowptr - node_off;
    const int rp_base = rp[__builtin_amdgcn_readfirstlane(gf)];      // wave-uniform: a scalar load
    const int rp_row = gf + min(c, max(gl - gf - 1, 0));
    int rp_lo = rp[rp_row], rp_hi = rp[rp_row + 1];
    int er0 = 0, edeg = 0;
    auto edge_ranges = [&]() {           // called once the stage's own loads are out (and about to be waited for anyway)
        __builtin_amdgcn_sched_barrier(0);       // (the copies into the asm's operands carry the wait for rp: they must not be hoisted above the loads)
        asm volatile("" : "+v"(rp_lo), "+v"(rp_hi));
        er0 = rp_lo - rp_base;
        edeg = c < gl - gf ? rp_hi - rp_lo : 0;
    };
    // nibble -> four fp16 values 0 / 1: the building block of the mean's selection matrix (read after many barriers)
    if (tid < 16) {
        unsigned* lut = reinterpret_cast<unsigned*>(lds + TILE_MAIN_FLOATS);
        lut[2 * tid] = ((tid & 1) ? 0x3C00u : 0u) | ((tid & 2) ? 0x3C000000u : 0u);
        lut[2 * tid + 1] = ((tid & 4) ? 0x3C00u : 0u) | ((tid & 8) ? 0x3C000000u : 0u);
    }
    const float* prow = pl + (sl & 255) * PQLD + 4 * hh;
    const float* qrow = ql + ((sl >> 8) & 255) * PQLD + 4 * hh;

    float b2v[4];
    // nodes of the slots this thread stages: rows (tid >> 5) + 8 i of h (or P / Q), row tid >> 3 of the feature columns
    int rnode[4], fnode;
    if (LISTED) {
        const int* tnode = a.tile_node + (size_t)tm * TILE_NCAP;
#pragma unroll
        for (int i = 0; i < 4; ++i) rnode[i] = tnode[(tid >> 5) + 8 * i];
        fnode = tnode[tid >> 3];
#pragma unroll
        for (int i = 0; i < 4; ++i) rnode[i] += node_off;
        fnode += node_off;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) rnode[i] = node_of((tid >> 5) + 8 * i);
        fnode = node_of(tid >> 3);
    }

    WHalf ws;
    whalf_load(ws, a.w2s, tid);                      // W2 chunk 0, K step 0: stored once the buffer is free

    if (!FOLD) {
        const int piece = tid & 31;
        f32x4 pv[4], qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pv[i] = *reinterpret_cast<const f32x4*>(a.P + (size_t)rnode[i] * H + 4 * piece) * ACT_SCALE;
            qv[i] = *reinterpret_cast<const f32x4*>(a.Q + (size_t)rnode[i] * H + 4 * piece) * ACT_SCALE;
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) b2v[T] = a.b2[32 * T + c];
        edge_ranges();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = (tid >> 5) + 8 * i;
            *reinterpret_cast<f32x4*>(pl + slot * PQLD + 4 * piece) = pv[i];
            *reinterpret_cast<f32x4*>(ql + slot * PQLD + 4 * piece) = qv[i];
        }
        whalf_store(ws, wbuf, tid);
        __syncthreads();
        TPROF(0);
    } else {
        // fragment tile at the START of the LDS (it overlays the weight buffer and the head of the P rows)
        _Float16* bt = reinterpret_cast<_Float16*>(lds);
        const int ntail = a.nc1 - 8;
        unsigned woff = (unsigned)((wave * 2 * 64 + lane) * 16);
        asm volatile("" : "+v"(woff));
        auto wfrag = [&](int ch, int s, int plane) {
            const char* base = reinterpret_cast<const char*>(a.w1s) + (size_t)ch * (SPLIT_CHUNK_FLOATS * 4) + (s * 8 + plane) * 1024;
            return *reinterpret_cast<const half8*>(base + woff);
        };
        half8 wp[3][2][2], wq[3][2][2];
        f32x4 b1v[4];
        auto wload = [&](int i) {
            const int chp = i < 4 ? i : 8, chq = i < 4 ? 4 + i : (ntail > 1 ? 9 : 8);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) { wp[i % 3][s][p] = wfrag(chp, s, p); wq[i % 3][s][p] = wfrag(chq, s, p); }
        };
        {
            f32x4 hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const f32x4*>(a.h + (size_t)rnode[i] * H + 4 * (tid & 31));
            const int g = tid & 7;
            float tx[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (packed_feat) {
                // loaded on BOTH sides of `jc < ntail` (a second copy of piece 0 where there is no second chunk) and selected afterwards: as a
                // load inside the branch, the value was merged with the zeros of the other side by a copy right behind it, and that copy
                // waited (vmcnt(0)) for every row load issued above before the weight prefetches below could be requested
#pragma unroll
                for (int jc = 0; jc < 2; ++jc) {
                    const bool on = jc < ntail;
                    const f32x4 fv = *reinterpret_cast<const f32x4*>(a.feat + (size_t)fnode * (32 * ntail) + 32 * (on ? jc : 0) + 4 * g);
#pragma unroll
                    for (int m = 0; m < 4; ++m) tx[jc][m] = on ? fv[m] : 0.f;
                }
            } else {
#pragma unroll
                for (int jc = 0; jc < 2; ++jc)
                    if (jc < ntail) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const int k = 32 * jc + 4 * g + m;
                            const float uv = a.u[(size_t)fnode * a.tw + min(k, a.tw - 1)];
                            const float vv = a.vars[(size_t)fnode * a.nv + min(max(k - a.tw - 1, 0), a.nv - 1)];
                            tx[jc][m] = k < a.tw ? uv : (k == a.tw ? a.pos[fnode] : (k <= a.tw + a.nv ? vv : 0.f));
                        }
                    }
            }
            // message_net_1's bias for the P accumulators: requested with this batch, AHEAD of the weight chunks (loads complete in order:
            // the first MFMA then waits for the bias and chunk 0 only).  Read where the projection starts, behind the stage's barrier,
            // it was waited for ten instructions behind its request, in the middle of the MFMA stream.
#pragma unroll
            for (int q = 0; q < 4; ++q) b1v[q] = *reinterpret_cast<const f32x4*>(a.b1 + 32 * wave + 8 * q + 4 * hh);
            wload(0);                                // the projection's first two weight chunks: behind the row loads
            wload(1);
            edge_ranges();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                half2v h01, l01, h23, l23;
                split_node_pair(hv[i][0], hv[i][1], h01, l01);
                split_node_pair(hv[i][2], hv[i][3], h23, l23);
                const int piece = tid & 31;
                _Float16* row = bt + frag_row(piece >> 3, (tid >> 5) + 8 * i) + 4 * (piece & 7);
                *reinterpret_cast<half4v*>(row) = half4v{h01[0], h01[1], h23[0], h23[1]};
                *reinterpret_cast<half4v*>(row + 32) = half4v{l01[0], l01[1], l23[0], l23[1]};
            }
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) {
                if (jc < ntail) {
                    // Q takes -(u, pos) and no variables: the fp16 halves of -x are the negated halves of x
                    half2v ph[2], pl2[2], qh[2], ql2[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        split_node_pair(tx[jc][2 * m], tx[jc][2 * m + 1], ph[m], pl2[m]);
                        const int k = 32 * jc + 4 * g + 2 * m;
                        const half2v nm = {(_Float16)(k <= a.tw ? -1.f : 0.f), (_Float16)(k + 1 <= a.tw ? -1.f : 0.f)};
                        qh[m] = ph[m] * nm;
                        ql2[m] = pl2[m] * nm;
                    }
                    _Float16* rp_ = bt + frag_row(4 + 2 * jc, tid >> 3) + 4 * g;
                    _Float16* rq_ = bt + frag_row(5 + 2 * jc, tid >> 3) + 4 * g;
                    *reinterpret_cast<half4v*>(rp_) = half4v{ph[0][0], ph[0][1], ph[1][0], ph[1][1]};
                    *reinterpret_cast<half4v*>(rp_ + 32) = half4v{pl2[0][0], pl2[0][1], pl2[1][0], pl2[1][1]};
                    *reinterpret_cast<half4v*>(rq_) = half4v{qh[0][0], qh[0][1], qh[1][0], qh[1][1]};
                    *reinterpret_cast<half4v*>(rq_ + 32) = half4v{ql2[0][0], ql2[0][1], ql2[1][0], ql2[1][1]};
                }
            }
        }
        __syncthreads();
        TPROF(0);

        // Projections TRANSPOSED (weight fragments as the A operand, node fragments as B): D[channel 32 wave + acc_row(r, hh)][slot c],
        // i.e. a lane holds four CONSECUTIVE channels of its slot per register quad: the P / Q rows go to LDS as 16-byte pieces
        // (8 ds_write_b128; round 3: 32 ds_write_b32 + 32 multiplications) and keep the accumulators' scale 2^s 2^8, which the
        // activation folds into its constants.
        const float sc = uniform_ro(a.scales, 0) * NODE_SCALE;
        f32x16 accP, accQ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = b1v[q];
#pragma unroll
            for (int m = 0; m < 4; ++m) { accP[4 * q + m] = bv[m] * sc; accQ[4 * q + m] = 0.f; }
        }
        auto afrag = [&](int ch, half8 (&ahi)[2], half8 (&alo)[2]) {
            const _Float16* row = bt + frag_row(ch, c) + 8 * hh;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ahi[s] = *reinterpret_cast<const half8*>(row + 16 * s);
                alo[s] = *reinterpret_cast<const half8*>(row + 32 + 16 * s);
            }
        };
        auto mma3 = [&](f32x16& acc, const half8& ahi, const half8& alo, const half8& whi, const half8& wlo) {
            MSMP_MFMA_LOLO(2, acc, wlo, alo);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc, 0, 0, 0);
        };
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int cur = ch % 3;
            if (ch + 2 < 4 || (ch + 2 == 4 && ntail > 0)) wload(ch + 2);
            half8 ahi[2], alo[2];
            afrag(ch, ahi, alo);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mma3(accP, ahi[s], alo[s], wp[cur][s][0], wp[cur][s][1]);
                mma3(accQ, ahi[s], alo[s], wq[cur][s][0], wq[cur][s][1]);
            }
        }
        for (int jc = 0; jc < ntail; ++jc) {
            half8 phi[2], plo[2], qhi[2], qlo[2];
            afrag(4 + 2 * jc, phi, plo);
            afrag(5 + 2 * jc, qhi, qlo);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 whi = jc == 0 ? wp[1][s][0] : wq[1][s][0], wlo = jc == 0 ? wp[1][s][1] : wq[1][s][1];
                mma3(accP, phi[s], plo[s], whi, wlo);
                mma3(accQ, qhi[s], qlo[s], whi, wlo);
            }
        }
        TPROF(1);
#pragma unroll
        for (int T = 0; T < 4; ++T) b2v[T] = a.b2[32 * T + c];      // message_net_2's bias: requested two barriers ahead of the accumulator initialisation
        __syncthreads();                             // every wave is done with the fragment tile: the P / Q rows and the weight buffer may be written
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(pl + c * PQLD + 32 * wave + 8 * q + 4 * hh) = f32x4{accP[4 * q], accP[4 * q + 1], accP[4 * q + 2], accP[4 * q + 3]};
            *reinterpret_cast<f32x4*>(ql + c * PQLD + 32 * wave + 8 * q + 4 * hh) = f32x4{accQ[4 * q], accQ[4 * q + 1], accQ[4 * q + 2], accQ[4 * q + 3]};
        }
        whalf_store(ws, wbuf, tid);
        __syncthreads();
        TPROF(2);
    }

    // ---- message_net_2 on Swish(P_i + Q_j): eight K = 16 steps; the matrix work of step u (12 MFMAs = 4 groups of 3) is interleaved
    // with the activation of the NEXT step's operand (4 slices of two values).
    // The GEMM is computed TRANSPOSED (the activation fragments are the A operand, the W2 fragments the B operand: both have the
    // same lane / k structure): y[T][r] = message_net_2 of edge acc_row(r, hh) of this wave, channel 32 T + c.  Registers 8 s .. 8 s + 7
    // of a tile are then exactly the B fragment of K step s of one more MFMA over the wave's 32 edges: the per-target sum.
    f32x16 y[4];
    {
        const float s2 = uniform_ro(a.scales, 1) * ACT_SCALE;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            float bv = b2v[T] * s2;
            asm volatile("" : "+v"(bv));         // consumed HERE, ahead of the K loop's first LDS-DMA: a wait placed behind the DMA is vmcnt(0)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[T][r] = bv;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 pq[4];                 // P (0, 1) and Q (2, 3) pieces of the K step being activated: channels 32 t + 16 s + 8 j + 4 hh .. + 3
    auto gather_step = [&](int t, int s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pq[j] = *reinterpret_cast<const f32x4*>(prow + 32 * t + 16 * s + 8 * j);
            pq[2 + j] = *reinterpret_cast<const f32x4*>(qrow + 32 * t + 16 * s + 8 * j);
        }
    };
    float zt[8];
    // the rows carry x_raw = g x64 (g = 2^s 2^8 / 2^6, a power of two; 1 for staged rows):  z = 64 Swish(x) = x_raw / (g (1 + e^-x))
    const float act_g = FOLD ? uniform_ro(a.scales, 0) * (NODE_SCALE / ACT_SCALE) : 1.0f;
    const float act_ct = (-1.44269504088896340736f / ACT_SCALE) * (FOLD ? uniform_ro(a.scales, 4) * (ACT_SCALE / NODE_SCALE) : 1.0f);     // / act_g, a power of two: exact either way
    auto act_slice = [&](int i) {        // i = 0..3: piece j = i >> 1, elements 2 (i & 1), + 1
        const int j = i >> 1, m0 = 2 * (i & 1);
#if MSMP_PRECISE_ACT
        for (int m = m0; m < m0 + 2; ++m) zt[4 * j + m] = ACT_SCALE * swishf((pq[j][m] + pq[2 + j][m]) * (1.0f / (ACT_SCALE * act_g)));
        (void)act_ct;
#else
        const f32x2 x = f32x2{pq[j][m0], pq[j][m0 + 1]} + f32x2{pq[2 + j][m0], pq[2 + j][m0 + 1]};
        const f32x2 t = x * f32x2{act_ct, act_ct};
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} * f32x2{act_g, act_g} + f32x2{act_g, act_g};
        const f32x2 z = x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        zt[4 * j + m0] = z[0];
        zt[4 * j + m0 + 1] = z[1];
#endif
    };
    half8 bhi[2], blo[2];          // [parity of the step]
    gather_step(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) act_slice(i);
    split8(zt, bhi[0], blo[0]);
    TPROF(3);
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int par = u & 1;
        // the next K step's weight half chunk by LDS-DMA into the buffer the previous step read (free since that step's barrier); the
        // barrier at the end of this step waits for it
        whalf_dma(a.w2s + (size_t)(u + 1) * WHALF_FLOATS, wbuf + (par ^ 1) * WHALF_FLOATS, tid);
        const half8* w = reinterpret_cast<const half8*>(wbuf + par * WHALF_FLOATS) + lane;
        half8 ahi[4], alo[4];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            ahi[T] = w[(T * 2 + 0) * 64];
            alo[T] = w[(T * 2 + 1) * 64];
        }
        gather_step((u + 1) >> 1, (u + 1) & 1);
#pragma unroll
        for (int T = 2; T < 4; ++T) {
            ahi[T] = w[(T * 2 + 0) * 64];
            alo[T] = w[(T * 2 + 1) * 64];
        }
        TILE_SCHED_BARRIER();
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            MSMP_MFMA_LOLO(2, y[T], blo[par], alo[T]);
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[par], alo[T], y[T], 0, 0, 0);
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo[par], ahi[T], y[T], 0, 0, 0);
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[par], ahi[T], y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            act_slice(T);
            if (T == 3) split8(zt, bhi[par ^ 1], blo[par ^ 1]);
            TILE_SCHED_BARRIER();
        }
        TPROF(4);
        __syncthreads();
        TPROF(5);
    }

    // ---- last K step, Swish of the messages and the per-target mean as one software pipeline over the channel tiles T: the vector
    // work of tile T (Swish, fp16 split) is issued between the MFMAs of tile T + 1 (last K step) resp. of the means.  The mean of a
    // target is one more MFMA: S[target row n][edge k] (0 / 1, exact in fp16) times the wave's message tile as fp16 hi + lo fragments
    // straight from the accumulator registers, fp32 accumulation, then x 1 / (64 deg).  Same arithmetic in the same order per value
    // as the straight-line form of round 3: bit-identical.
    // m64 = 64 Swish(y).  The accumulators hold yy = 2^s 64 y; with kinv = 2^s:  m64 = yy / (kinv (1 + e^-y)).
    const float kinv = uniform_ro(a.scales, 1);
    const float inv2 = uniform_ro(a.scales, 5) * (1.0f / ACT_SCALE);
    const float cexp = -1.44269504088896340736f * inv2;
    auto swish2 = [&](int T, int r0, int r1) {
#pragma unroll
        for (int r = r0; r < r1; r += 2) {
#if MSMP_PRECISE_ACT
            y[T][r] = ACT_SCALE * swishf(y[T][r] * inv2);
            y[T][r + 1] = ACT_SCALE * swishf(y[T][r + 1] * inv2);
            (void)kinv; (void)cexp;
#else
            const f32x2 yy = {y[T][r], y[T][r + 1]};
            const f32x2 t = yy * f32x2{cexp, cexp};
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            const f32x2 d = e * f32x2{kinv, kinv} + f32x2{kinv, kinv};
            const f32x2 z = yy * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            y[T][r] = z[0];
            y[T][r + 1] = z[1];
#endif
        }
    };
    half8 sf[2];
    {
        constexpr int par = 1;
        const half8* w = reinterpret_cast<const half8*>(wbuf + par * WHALF_FLOATS) + lane;
        half8 ahi[4], alo[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            ahi[T] = w[(T * 2 + 0) * 64];
            alo[T] = w[(T * 2 + 1) * 64];
        }
        TILE_SCHED_BARRIER();
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            MSMP_MFMA_LOLO(2, y[T], blo[par], alo[T]);
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[par], alo[T], y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T > 0) swish2(T - 1, 0, 6);
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo[par], ahi[T], y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T > 0) swish2(T - 1, 6, 12);
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[par], ahi[T], y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T > 0) swish2(T - 1, 12, 16);
            if (T == 0) {
                // Selection matrix S[target row n][edge k] = 1 iff lane k of this wave holds an in-edge of target gf + n: the A operand,
                // built from the lane's edge range as a bit mask through the nibble table; K index j of step s is edge
                // 16 s + 8 (j >> 2) + 4 hh + (j & 3) (the accumulator's row order).
                const unsigned mask = edeg >= 32 ? 0xffffffffu : ((1u << edeg) - 1u) << (er0 & 31);
                const char* lut = reinterpret_cast<const char*>(lds + TILE_MAIN_FLOATS);
                using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(lut + 8 * ((mask >> (16 * s + 4 * hh)) & 15u));
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(lut + 8 * ((mask >> (16 * s + 8 + 4 * hh)) & 15u));
                    const u32x4 wv = {lo[0], lo[1], hi[0], hi[1]};
                    sf[s] = __builtin_bit_cast(half8, wv);
                }
            }
            TILE_SCHED_BARRIER();
        }
        TPROF(4);
    }
    // 1 / (64 deg) per accumulator row: row n's factor sits in lane n; the accumulator rows of a lane are (r & 3) + 8 (r >> 2) + 4 hh.
    // Through a per-wave table in the (dead) P rows: every wave has passed the last barrier after its last gather.
    f32x4 fr[4];
    {
        float* ftab = pl + wave * 32;
        if (hh == 0) ftab[c] = (1.0f / ACT_SCALE) / (float)max(edeg, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[q] = *reinterpret_cast<const f32x4*>(ftab + 8 * q + 4 * hh);
    }
    {
        half8 mh[2], ml[2];
        auto splitT = [&](int T) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = y[T][8 * s + j];
                split8(v, mh[s], ml[s]);
            }
        };
        splitT(0);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const half8 h0 = mh[0], l0 = ml[0], h1 = mh[1], l1 = ml[1];
            f32x16 z16;
#pragma unroll
            for (int r = 0; r < 16; ++r) z16[r] = 0.f;
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf[0], l0, z16, 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T == 0) swish2(3, 0, 8);
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf[0], h0, y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T == 0) swish2(3, 8, 16);
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf[1], l1, y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
            if (T < 3) splitT(T + 1);
            TILE_SCHED_BARRIER();
            y[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sf[1], h1, y[T], 0, 0, 0);
            TILE_SCHED_BARRIER();
        }
    }
    TPROF(7);
    // y[T][r] = 64 x (sum of the messages of target gf + acc_row(r, hh)), channel 32 T + c: scale and store 128-byte row pieces
    {
        const int ng = gl - gf;
        float* out = a.agg + (size_t)gf * H + c;
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (8 * q < ng) {                        // wave-uniform: rows 8 q .. 8 q + 7 hold targets
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int row = 8 * q + 4 * hh + m;
                    if (row < ng) {
#pragma unroll
                        for (int T = 0; T < 4; ++T) {
                            const float v = y[T][4 * q + m] * fr[q][m];
                            out[(size_t)row * H + 32 * T] = v;
                            bad |= out_of_range(v);      // also NaN / Inf: a node row or an activation beyond fp16 upstream
                        }
                    }
                }
            }
        }
        if (bad) status_raise(a.status, MSMP_STATUS_NODE_SATURATED);
    }
    TPROF(9);
    TPROF_FLUSH
}

template <int MODE, bool LISTED>
__global__ __launch_bounds__(256, 3) void edge_tile_kernel(TileArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[TILE_LDS_FLOATS];
    edge_tile_body<MODE, LISTED>(a, lds);
}

// Both heads of a gated pair in ONE launch (blockIdx.y = head; same body, bit-identical results): small batches are bound by the
// latency of their dependent launches, and with the projections folded in a gated pair is then two launches (this + the node tail).
struct TileArgs2 {
    TileArgs head[2];
};
template <int MODE, bool LISTED>
__global__ __launch_bounds__(256, 3) void edge_tile_pair_kernel(TileArgs2 a) {
    __shared__ __attribute__((aligned(16))) float lds[TILE_LDS_FLOATS];
    edge_tile_body<MODE, LISTED>(a.head[blockIdx.y], lds);
}

}  // namespace msmp

using namespace msmp;

#if MSMP_PROF_TILE
extern "C" __attribute__((visibility("default"))) int msmp_debug_prof_tile(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_tile), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof_tile), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int msmp_node_feature_stride(int tw, int nv) {
    if (tw <= 0 || nv < 1 || nv > MSMP_MAX_VARS) return -1;
    return 32 * tail_chunks(tw, nv);
}

extern "C" int msmp_pack_node_features_f32(const float* u, const float* pos, const float* vars, int64_t n_nodes, int tw, int nv,
                                           float* feat_out, msmp_stream_t stream) {
    MSMP_REQUIRE(u && pos && vars && feat_out, MSMP_ERR_ARG, "msmp_pack_node_features_f32: null pointer");
    const int stride = msmp_node_feature_stride(tw, nv);
    MSMP_REQUIRE(n_nodes > 0 && stride > 0, MSMP_ERR_ARG, "msmp_pack_node_features_f32: bad sizes");
    const long total = (long)n_nodes * stride;
    hipLaunchKernelGGL(pack_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, pos, vars,
                       (long)n_nodes, tw, nv, stride, feat_out, msmp_tune_get("split") ? status_ptr() : nullptr);
    return check_launch("pack_features_kernel");
}

extern "C" int msmp_prepare_nodes(const void* x, int x_f64, const void* pos, int pos_f64, int64_t n_nodes, int tw, double L, double tmax,
                                  int n_cols, const void* const* cols, const int* col_f64, const double* col_div, float* u_out,
                                  float* pos_x_out, float* pos_t_out, float* vars_out, float* feat_out, msmp_stream_t stream) {
    MSMP_REQUIRE(x && pos && u_out && pos_x_out && pos_t_out && vars_out && (n_cols == 0 || (cols && col_f64 && col_div)), MSMP_ERR_ARG,
                 "msmp_prepare_nodes: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && tw > 0 && n_cols >= 0 && n_cols < MSMP_MAX_VARS, MSMP_ERR_ARG, "msmp_prepare_nodes: bad sizes");
    const int stride = msmp_node_feature_stride(tw, 1 + n_cols);
    MSMP_REQUIRE(stride > 0 && stride <= 64 + 32, MSMP_ERR_ARG, "msmp_prepare_nodes: bad sizes");
    PrepArgs a{};
    a.x = x; a.pos = pos; a.x_f64 = x_f64; a.pos_f64 = pos_f64; a.n_cols = n_cols; a.L = L; a.tmax = tmax; a.n = (long)n_nodes; a.tw = tw;
    a.stride = stride; a.u = u_out; a.pos_x = pos_x_out; a.pos_t = pos_t_out; a.vars = vars_out; a.feat = feat_out;
    a.status = msmp_tune_get("split") ? status_ptr() : nullptr;       // the exact-fp32 kernels have no range limit
    for (int c = 0; c < n_cols; ++c) {
        MSMP_REQUIRE(cols[c] && col_div[c] != 0.0, MSMP_ERR_ARG, "msmp_prepare_nodes: bad column %d", c);
        a.col[c] = cols[c]; a.col_f64[c] = col_f64[c]; a.col_div[c] = col_div[c];
    }
    const long total = (long)n_nodes * (stride / 4);
    hipLaunchKernelGGL(prepare_nodes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("prepare_nodes_kernel");
}

extern "C" int msmp_build_tiles(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int64_t n_edges, int group_nodes,
                                int32_t* tile_node_out, int32_t* tile_count_out, int32_t* tile_halo_out, int32_t* edge_slot_out,
                                int32_t* stats_out, msmp_stream_t stream) {
    MSMP_REQUIRE(rowptr && col && tile_node_out && tile_count_out && tile_halo_out && edge_slot_out && stats_out, MSMP_ERR_ARG,
                 "msmp_build_tiles: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_edges >= 0 && n_nodes < (1L << 31) && n_edges < (1L << 31), MSMP_ERR_ARG, "msmp_build_tiles: bad sizes");
    MSMP_REQUIRE(group_nodes >= 1 && 4 * group_nodes <= MSMP_TILE_NCAP, MSMP_ERR_ARG, "msmp_build_tiles: group_nodes must be in 1..%d",
                 MSMP_TILE_NCAP / 4);
    hipStream_t st = (hipStream_t)stream;
    const hipError_t me = hipMemsetAsync(stats_out, 0, 3 * sizeof(int32_t), st);
    MSMP_REQUIRE(me == hipSuccess, MSMP_ERR_HIP, "msmp_build_tiles: memset: %s", hipGetErrorString(me));
    const int tile_nodes = 4 * group_nodes;
    const unsigned n_tiles = (unsigned)((n_nodes + tile_nodes - 1) / tile_nodes);
    hipLaunchKernelGGL(build_tiles_kernel, dim3(n_tiles), dim3(128), 0, st, rowptr, col, (int)n_nodes, tile_nodes, group_nodes, tile_node_out,
                       tile_count_out, tile_halo_out, edge_slot_out, stats_out);
    return check_launch("build_tiles_kernel");
}

// a caller-supplied descriptor is trusted for its pointers only: its geometry must cover exactly n_nodes
bool msmp_tiles_ok(const msmp_tiles_t* t, int64_t n_nodes) {
    if (!(t->tile_node && t->tile_count && t->tile_halo && t->edge_slot && t->group_nodes >= 1 && t->tile_nodes == 4 * t->group_nodes &&
          t->tile_nodes <= MSMP_TILE_NCAP && t->n_tiles >= 1))
        return false;
    if (t->period_tiles > 0) {       // one period's tiles, repeated: whole periods only, and the kernel's 32-bit division must hold
        return t->period_nodes >= 1 && n_nodes % t->period_nodes == 0 && (int64_t)t->period_tiles * t->tile_nodes >= t->period_nodes &&
               (int64_t)(t->period_tiles - 1) * t->tile_nodes < t->period_nodes && (int64_t)t->n_tiles == n_nodes / t->period_nodes * t->period_tiles &&
               (int64_t)t->n_tiles * t->period_tiles < (1LL << 32);
    }
    return (int64_t)t->n_tiles * t->tile_nodes >= n_nodes && (int64_t)(t->n_tiles - 1) * t->tile_nodes < n_nodes;
}

static TileArgs tile_args(const float* h, const float* u, const float* pos, const float* vars, const float* feat, const float* p, const float* q,
                          const int32_t* rowptr, const msmp_tiles_t* t, int64_t n_nodes, int64_t n_edges, int tw, int nv, const PackedLayout& L,
                          const float* packed, float* agg) {
    TileArgs a{};
    a.h = h; a.u = u; a.pos = pos; a.vars = vars; a.feat = feat; a.P = p; a.Q = q; a.rowptr = rowptr;
    a.tile_node = t->tile_node; a.tile_halo = t->tile_halo; a.edge_slot = t->edge_slot;
    a.n_nodes = (long)n_nodes; a.n_edges = (long)n_edges; a.tile_nodes = t->tile_nodes; a.group_nodes = t->group_nodes;
    a.period_tiles = t->period_tiles > 0 ? t->period_tiles : 0;
    a.period_nodes = t->period_tiles > 0 ? t->period_nodes : 0;
    a.period_magic = t->period_tiles > 1 ? (unsigned)(((1ULL << 32) + t->period_tiles - 1) / t->period_tiles) : 0u;
    a.tw = tw; a.nv = nv; a.nc1 = L.nc1; a.w1s = packed + L.w1s; a.w2s = packed + L.w2s; a.scales = packed + L.scales; a.b1 = packed + L.b1;
    a.b2 = packed + L.b2; a.agg = agg; a.status = status_ptr();      // (these kernels exist on the split path only)
    return a;
}

template <int MODE>
static void launch_tiles(const TileArgs& a, bool listed, unsigned n_tiles, hipStream_t st) {
    if (listed) hipLaunchKernelGGL((edge_tile_kernel<MODE, true>), dim3(n_tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((edge_tile_kernel<MODE, false>), dim3(n_tiles), dim3(256), 0, st, a);
}

extern "C" int msmp_edge_aggregate_tiled_f32(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                                             const float* p, const float* q, const int32_t* rowptr, const msmp_tiles_t* tiles, int64_t n_nodes,
                                             int64_t n_edges, int tw, int nv, const float* packed, float* agg_out,
                                             msmp_stream_t stream) {
    MSMP_REQUIRE(rowptr && tiles && packed && agg_out, MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_f32: null pointer");
    MSMP_REQUIRE((p != nullptr) == (q != nullptr), MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_f32: give both of p, q or neither");
    const bool fold = p == nullptr;
    MSMP_REQUIRE(!fold || (h && u && pos && vars), MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_f32: null pointer (h, u, pos, vars)");
    MSMP_REQUIRE(msmp_tiles_ok(tiles, n_nodes), MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_f32: the tile descriptor does not cover %ld nodes", (long)n_nodes);
    MSMP_REQUIRE(n_nodes > 0 && n_edges >= 0 && n_nodes < (1L << 31) && n_edges < (1L << 31) && tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS,
                 MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_f32: bad sizes");
    MSMP_REQUIRE(msmp_tune_get("split"), MSMP_ERR_UNSUPPORTED, "msmp_edge_aggregate_tiled_f32: only on the fp16-split matrix path");
    const PackedLayout L = packed_layout(tw, nv);
    MSMP_REQUIRE(!fold || L.nc1 - 8 <= 2, MSMP_ERR_UNSUPPORTED, "msmp_edge_aggregate_tiled_f32: tw + 1 + nv <= 64");
    const TileArgs a = tile_args(h, u, pos, vars, feat, p, q, rowptr, tiles, n_nodes, n_edges, tw, nv, L, packed, agg_out);
    hipStream_t st = (hipStream_t)stream;
    timing_begin(MSMP_K_EDGE_MLP, st);
    const bool listed = tiles->listed != 0 || !msmp_tune_get("tile_arith");
    if (fold && feat) launch_tiles<2>(a, listed, (unsigned)tiles->n_tiles, st);
    else if (fold) launch_tiles<1>(a, listed, (unsigned)tiles->n_tiles, st);
    else launch_tiles<0>(a, listed, (unsigned)tiles->n_tiles, st);
    timing_end(MSMP_K_EDGE_MLP, st);
    return check_launch("edge_tile_kernel");
}

// Rows L1 + L2 of BOTH heads of a gated pair in one launch (library-internal; msmp_mp_layer_f32 at small batches).
int msmp_edge_aggregate_tiled_pair(const float* h, const float* u, const float* pos, const float* vars, const float* feat, const int32_t* rowptr,
                                   const msmp_tiles_t* tiles, int64_t n_nodes, int64_t n_edges, int tw, int nv, const float* packed_a,
                                   const float* packed_b, float* agg_a, float* agg_b, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && rowptr && tiles && packed_a && packed_b && agg_a && agg_b, MSMP_ERR_ARG,
                 "msmp_edge_aggregate_tiled_pair: null pointer");
    MSMP_REQUIRE(msmp_tiles_ok(tiles, n_nodes), MSMP_ERR_ARG, "msmp_edge_aggregate_tiled_pair: the tile descriptor does not cover %ld nodes", (long)n_nodes);
    const PackedLayout L = packed_layout(tw, nv);
    MSMP_REQUIRE(L.nc1 - 8 <= 2 && msmp_tune_get("split"), MSMP_ERR_UNSUPPORTED, "msmp_edge_aggregate_tiled_pair: unsupported configuration");
    TileArgs2 a2;
    a2.head[0] = tile_args(h, u, pos, vars, feat, nullptr, nullptr, rowptr, tiles, n_nodes, n_edges, tw, nv, L, packed_a, agg_a);
    a2.head[1] = tile_args(h, u, pos, vars, feat, nullptr, nullptr, rowptr, tiles, n_nodes, n_edges, tw, nv, L, packed_b, agg_b);
    hipStream_t st = (hipStream_t)stream;
    timing_begin(MSMP_K_EDGE_MLP, st);
    const dim3 grid((unsigned)tiles->n_tiles, 2);
    const bool listed = tiles->listed != 0 || !msmp_tune_get("tile_arith");
    if (feat && listed) hipLaunchKernelGGL((edge_tile_pair_kernel<2, true>), grid, dim3(256), 0, st, a2);
    else if (feat) hipLaunchKernelGGL((edge_tile_pair_kernel<2, false>), grid, dim3(256), 0, st, a2);
    else if (listed) hipLaunchKernelGGL((edge_tile_pair_kernel<1, true>), grid, dim3(256), 0, st, a2);
    else hipLaunchKernelGGL((edge_tile_pair_kernel<1, false>), grid, dim3(256), 0, st, a2);
    timing_end(MSMP_K_EDGE_MLP, st);
    return check_launch("edge_tile_pair_kernel");
}
