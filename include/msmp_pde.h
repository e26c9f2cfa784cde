/*
 * msmp_pde.h -- C-ABI of libmsmp_pde.so: the MI355X (gfx950) message-passing rollout path of MSMP-PDE.
 *
 * This is the drop-in boundary.  The reference (Leqr/MSMP-PDE, 100 % Python)
 * reaches native code for this path only through third-party Python extensions (torch_geometric,
 * torch_scatter, torch_cluster, lem_cuda); each entry point below names the reference call site it
 * replaces (paths relative to the reference root).  Conventions:
 *
 *   - plain C: raw DEVICE pointers + sizes, no torch / C++ types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is stream-ordered,
 *     never synchronises the device and never allocates: scratch comes from `workspace`;
 *   - all matrices row-major; float tensors are fp32 unless the name says f64; indices are int32 on
 *     the compute path (int64 only where the reference's `edge_index` is produced / consumed);
 *   - returns MSMP_OK (0) or a negative MSMP_ERR_*; msmp_last_error() gives the text; never throws;
 *   - hidden width is fixed at MSMP_HIDDEN = 128 (hidden_features, experiments/models_gnn.py:158).
 *
 * Layer weights are passed as a "packed layer" blob made by msmp_pack_layer_f32 from the reference's
 * eight nn.Linear tensors ([out,in] layout); see DESIGN.md section 3 for the blob layout.
 */
#ifndef MSMP_PDE_H
#define MSMP_PDE_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with hidden visibility and -Bsymbolic: only the msmp_* entry points are exported, and its
 * internal (rocPRIM, C++ runtime template) symbols can neither be interposed by nor leak into the host process
 * (PyTorch ships its own rocPRIM build). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MSMP_HIDDEN 128
#define MSMP_MAX_VARS 8          /* len(eq_variables) + 1 <= 8 */

#define MSMP_OK               0
#define MSMP_ERR_ARG         -1  /* null pointer / bad size */
#define MSMP_ERR_UNSUPPORTED -2  /* shape outside what the kernels are built for */
#define MSMP_ERR_WORKSPACE   -3  /* workspace too small */
#define MSMP_ERR_HIP         -4  /* HIP runtime error (text in msmp_last_error) */

/* flags for msmp_node_update_f32 / msmp_mp_layer_f32 */
#define MSMP_LAYER_RESIDUAL_SWISH 0  /* GNN_Layer.update    experiments/models_gnn.py:77-86  : x + Swish(W4 . + b4) */
#define MSMP_LAYER_LIN            1  /* GNN_LayerLin.update experiments/models_gnn.py:140-149: W4 . + b4           */
/* OR-ed into `mode` of msmp_mp_layer_f32: evaluate message_net_1 on the materialised per-edge
 * concatenation (the reference's literal order of operations) instead of the per-node factorisation. */
#define MSMP_LAYER_DENSE_MESSAGE  16

typedef void* msmp_stream_t;

/* ABI version: 400 = round 4 (msmp_tiles_t: listed, period_tiles, period_nodes; msmp_build_tiles writes 3 statistics);
 * 300 = round 3 (msmp_last_status, msmp_tiles_t checked on every entry point that takes one).  Bumped whenever a
 * prototype or a blob layout changes; the ctypes host refuses a library whose version is not the one it was written for. */
#define MSMP_ABI_VERSION 400
int msmp_version(void);
const char* msmp_last_error(void);

/* Sticky range status of the fp16-split matrix path ("Range").  The default kernels carry node rows scaled
 * by 2^8 (saturating at +-65504, i.e. |x| <= 255) and hidden activations scaled by 2^6 (|x| < 1023, beyond that fp16 inf); the
 * reference has no such limit.  Kernels OR a bit into a host-visible word when a value leaves that range, so wrong-but-finite or
 * NaN output never goes unnoticed:
 *   MSMP_STATUS_INPUT_RANGE     an input feature (u, pos / L, a variables column) had |x| > 255 or was not finite
 *                               (msmp_prepare_nodes, msmp_pack_node_features_f32);
 *   MSMP_STATUS_NODE_SATURATED  a hidden-state row staged by the tile kernel, a row of the blend, or an aggregate written by a
 *                               message kernel had |x| > 255 (or was NaN: an activation above 1023 overflowed upstream);
 *   MSMP_STATUS_NONFINITE       the InstanceNorm statistics of a graph were not finite (node tail).
 * msmp_last_status reads the word WITHOUT synchronising (it lives in host-mapped memory the kernels update with system-scope
 * atomics: what it returns covers all work that has completed; after a stream synchronise, all work issued).  reset != 0 clears
 * it.  Remedy: rescale the data, or run the exact-fp32 kernels (msmp_tune("split", 0): no range limit). */
#define MSMP_STATUS_INPUT_RANGE    1
#define MSMP_STATUS_NODE_SATURATED 2
#define MSMP_STATUS_NONFINITE      4
int msmp_last_status(int* flags_out, int reset);
/* Knobs for A/B measurements and validation (not part of the data contract):
 *   "split"   1 (default): the GEMMs of the node / edge / LEM kernels run on the fp16 matrix pipe with a 2-way
 *             fp16 split of both operands (fp32-class accuracy); 0: the fp32-MFMA kernels.
 *   "edge_nb" 0 auto, 1 / 2 force the 128- / 256-edge tile of the factorised message kernel.
 *   "tile"    2 (default): with node tiles that are at least 60 % full (>= 76 edges per tile on average), project P / Q inside the message
 *             kernel, else the gather kernels; 3: the same regardless of the fill; 1: msmp_node_project_f32 + tile kernel on the
 *             staged P / Q rows; 0: ignore the tiles (gather kernels).
 *   "bwd_gemm" 1 (default) / 2: msmp_mp_layer_bwd_f32 runs its row GEMMs on its own bf16x3 MFMA kernels (fused bias / Swish / dSwish epilogues;
 *             128-row workgroups from 32 768 rows on, 32-row workgroups whose waves split the output channels below);
 *             0: rocblas_sgemm + separate epilogue passes (A/B runs only: librocblas is loaded on first use).
 *   "lem_share" k (default 1): k LEM launches share the GPU (sub-batches on k streams): each plans its rounds for CUs / k.
 *   "lem_tail" 1 (default): the LEM launch ends with a round of one-tile workgroups where that saves >= 0.3 of a round; 0: three-tile
 *             workgroups only (same bits either way).
 *   "tile_arith" 1 (default): ranged tiles take their node rows by arithmetic on tile_halo; 0: always through the node list.
 *   "tail"    1 (default): msmp_mp_layer_f32 uses msmp_node_tail_f32 for graphs of up to 128 nodes; 0: the piecewise kernels.
 *   "pair"    gated pair: both heads' projection / message kernels in one launch each (bit-identical results): 0 never,
 *             1 (default) for batches of up to 65 536 nodes, where a step is bound by the latency of its ~60 dependent launches, 2 always.
 *   "lem"     LEM encoder edition: 4 (default) weight-stationary, three node tiles, matrix / vector halves of a SIMD's two waves in
 *             anti-phase; 3 the two-tile weight-stationary kernel of round 2; 0 fp32 MFMA ("split" 1 / 0 selects 4 / 0). */
int msmp_tune(const char* key, int value);
int msmp_tune_query(const char* key);      /* current value of "split", "tail", "pair", "bwd_gemm", "tile", "tile_arith" (0 for other keys) */

/* ---------------------------------------------------------------------------------------------
 * Weights
 * ------------------------------------------------------------------------------------------- */
/* Number of floats of one packed layer for a given u-feature width `tw` (25; 50 for *2D) and
 * variable count `nv` = len(eq_variables)+1. */
int64_t msmp_packed_layer_floats(int tw, int nv);

/* Pack message_net_1/2 and update_net_1/2 of one GNN_Layer / GNN_LayerLin
 * (experiments/models_gnn.py:47-58, 112-121) into the kernel layout.  w1 [128, 256+tw+1+nv],
 * w2 [128,128], w3 [128, 256+nv], w4 [128,128], b* [128]; all device pointers, reference layout. */
int msmp_pack_layer_f32(const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, const float* w4, const float* b4,
                        int tw, int nv, float* packed_out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Graph structure (rows G1 of SURVEY section 8a)
 * ------------------------------------------------------------------------------------------- */
/* CSR-by-target from the reference's edge_index [2,E] int64 (row 0 source j, row 1 target i;
 * torch_geometric MessagePassing flow='source_to_target', experiments/models_gnn.py:65,128).
 * Writes rowptr [N+1], col [E] (source of each edge, grouped by ascending target, original relative
 * order kept inside a target) and tgt [E] (target of each CSR slot).  If the input is already grouped
 * by ascending target (what the graph builders emit) no sort is needed; otherwise a stable device
 * sort runs in `workspace` (size from msmp_build_csr_workspace_bytes). */
size_t msmp_build_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes);
int msmp_build_csr(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes,
                   int32_t* rowptr_out, int32_t* col_out, int32_t* tgt_out,
                   void* workspace, size_t workspace_bytes, msmp_stream_t stream);

/* torch_cluster.radius_graph(x, r, batch, loop=False, max_num_neighbors) as called at
 * common/utils.py:368, on `dim`-column float64 coordinates x [N,dim]; graph g owns nodes
 * graph_ptr[g]..graph_ptr[g+1]-1.  Pair (j -> i) kept iff same graph, i != j and squared distance
 * < r*r (float64, strict); at most max_neighbors sources per target, lowest index first.
 * Two calls: _count writes rowptr [N+1] (in-degree prefix sums; E = rowptr[N]); _fill writes
 * edge_index [2,E] int64 in canonical order (ascending target, then ascending source). */
int msmp_radius_graph_count_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs,
                                int64_t n_nodes, double r, int max_neighbors,
                                int32_t* rowptr_out, msmp_stream_t stream);
int msmp_radius_graph_fill_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs,
                               int64_t n_nodes, double r, int max_neighbors, const int32_t* rowptr,
                               int64_t n_edges, int64_t* edge_index_out, msmp_stream_t stream);

/* torch_cluster.knn_graph(x, k, batch, loop=False) as called at common/utils.py:377,380: per target
 * its min(k, n_g-1) nearest same-graph nodes, ascending float64 squared distance, ties -> lower
 * index.  rowptr [N+1] and edge_index [2,E] are both written; E = sum_g n_g*min(k, n_g-1) is known
 * to the caller from graph sizes and passed as n_edges (writes past it are suppressed). */
int msmp_knn_graph_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs,
                       int64_t n_nodes, int k, int64_t n_edges, int32_t* rowptr_out,
                       int64_t* edge_index_out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Message-passing layer pieces (rows L1-L5)
 * ------------------------------------------------------------------------------------------- */
/* L1  GNN_Layer.message / GNN_LayerLin.message, experiments/models_gnn.py:69-75 / 132-138, with the
 * PyG gathers of propagate (:65,128) fused in:
 *   msg[e] = Swish(W2 Swish(W1 [h_i, h_j, u_i-u_j, pos_i-pos_j, vars_i] + b1) + b2),
 * i = tgt[e], j = col[e].  h [N,128], u [N,tw], pos [N] (= pos_x), vars [N,nv], msg_out [E,128]. */
int msmp_edge_mlp_f32(const float* h, const float* u, const float* pos, const float* vars,
                      const int32_t* tgt, const int32_t* col, int64_t n_nodes, int64_t n_edges,
                      int tw, int nv, const float* packed, float* msg_out, msmp_stream_t stream);

/* L1 + L2 fused: agg[i] = mean over the in-edges of i of the message above, without materialising the
 * [E,128] message tensor (messages are reduced per target inside the workgroup, in CSR order).  Needs
 * every target's in-edges to fit one workgroup tile: max_in_degree <= 256, else MSMP_ERR_UNSUPPORTED
 * (then use msmp_edge_mlp_f32 + msmp_scatter_mean_f32). */
int msmp_edge_aggregate_f32(const float* h, const float* u, const float* pos, const float* vars,
                            const int32_t* rowptr, const int32_t* col, const int32_t* tgt,
                            int64_t n_nodes, int64_t n_edges, int max_in_degree, int tw, int nv,
                            const float* packed, float* agg_out, msmp_stream_t stream);

/* Per-node factorisation of message_net_1 (linear in its concatenated input):
 *   W1 [h_i, h_j, u_i-u_j, p_i-p_j, v_i] + b1 = P[i] + Q[j],
 *   P[n] = W1[:, 0:128] h_n + W1[:, 256:] [u_n, p_n, v_n] + b1,   Q[n] = W1[:, 128:256] h_n - W1[:, 256:] [u_n, p_n, 0].
 * msmp_node_project_f32 writes P and Q [N,128]; msmp_edge_aggregate_projected_f32 then computes the same
 * agg as msmp_edge_aggregate_f32 from them (5.3x fewer FLOPs in message_net_1; only the place where partial
 * sums are rounded differs).  This is the default inside msmp_mp_layer_f32. */
int msmp_node_project_f32(const float* h, const float* u, const float* pos, const float* vars,
                          int64_t n_nodes, int tw, int nv, const float* packed, float* p_out,
                          float* q_out, msmp_stream_t stream);
int msmp_edge_aggregate_projected_f32(const float* p, const float* q, const int32_t* rowptr,
                                      const int32_t* col, const int32_t* tgt, int64_t n_nodes,
                                      int64_t n_edges, int max_in_degree, int tw, int nv,
                                      const float* packed, float* agg_out, msmp_stream_t stream);

/* Node tiles for the LDS-staged message kernel (north_star: "LDS staging of node tiles for edge gather"; the gathers it
 * replaces are PyG propagate's x_i = x[edge_index[1]], x_j = x[edge_index[0]], experiments/models_gnn.py:65,128).
 * A tile = four WAVE GROUPS of `group_nodes` consecutive TARGET nodes each (tile_nodes = 4 group_nodes) with all their in-edges
 * (at most 32 per group: one lane per edge, a target's edges never straddle two waves, so the per-target mean is one more MFMA
 * on the wave's message tile) plus every node those edges read as a source: at most MSMP_TILE_NCAP distinct nodes, listed per
 * tile with the targets first (slot k < tile size is node tile * tile_nodes + k), so the kernel loads each node row of a tile
 * ONCE (coalesced, 512-byte rows) into LDS and every edge reads its two operands from there through the per-edge slot pair.
 * Works for any graph whose tiles fit (banded 1-D radius / knn graphs, periodic wrap-around included: the list is by node id, not
 * by window); msmp_build_tiles reports the largest list / group edge count it met so the caller can pick a smaller group_nodes
 * or fall back to the gather kernels (always for in-degrees above 32). */
#define MSMP_TILE_NCAP  32       /* distinct nodes of a tile: one 32-row MFMA block */
#define MSMP_TILE_EDGES 128      /* edge lanes of a tile: one 32-lane group per wave */
typedef struct {
    int32_t tile_nodes;          /* target nodes per tile = 4 * group_nodes */
    int32_t group_nodes;         /* target nodes per wave group */
    int32_t n_tiles;             /* ceil(n_nodes / tile_nodes) */
    const int32_t* tile_node;    /* [n_tiles][MSMP_TILE_NCAP] node ids (unused slots repeat the tile's first node) */
    const int32_t* tile_count;   /* [n_tiles] number of valid slots */
    const int32_t* tile_halo;    /* [n_tiles][4]: (lo start, lo count, hi start, hi count) when the tile's sources outside it are one
                                  * run of consecutive nodes below and one above (slots then follow node order and the kernel needs
                                  * no list lookup); lo count = -1 otherwise */
    const int32_t* edge_slot;    /* [n_tiles][MSMP_TILE_EDGES]: lane 32 g + k = k-th in-edge (CSR order) of wave group g: target slot | source slot << 8;
                                  * 0 at lanes without an edge */
    int32_t listed;              /* != 0: some tile is not "ranged" (msmp_build_tiles' stats[2] > 0): the kernels read every slot's node
                                  * from tile_node; 0: they compute it from tile_halo and never touch the list */
    int32_t period_tiles;        /* 0: the arrays above describe all n_tiles tiles of the batch.  > 0 (ABI 400): the batch is
                                  * n_nodes / period_nodes copies of ONE pattern (the reference's batches: the same grid for every
                                  * sample, common/utils.py:364-380), the arrays hold the period_tiles tiles of one copy (built from
                                  * its CSR, node ids 0 .. period_nodes - 1; tile_nodes need not divide period_nodes: the last tile of
                                  * a copy is short, so no tile straddles two graphs) and tile t of the launch uses entry
                                  * t mod period_tiles with node ids shifted by (t div period_tiles) * period_nodes.  n_tiles is still the
                                  * launch's tile count.  rowptr passed beside such a descriptor is the whole batch's. */
    int32_t period_nodes;
} msmp_tiles_t;
/* stats_out (device, 3 x int32): largest node list, largest edge count of a wave group over the tiles (lists longer than
 * MSMP_TILE_NCAP are truncated in the output, groups of more than 32 edges cut: the structure is then not usable with this
 * group_nodes), number of tiles that are not ranged (-> msmp_tiles_t.listed). */
int msmp_build_tiles(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int64_t n_edges, int group_nodes,
                     int32_t* tile_node_out, int32_t* tile_count_out, int32_t* tile_halo_out, int32_t* edge_slot_out,
                     int32_t* stats_out, msmp_stream_t stream);
/* L1 + L2 on node tiles: same result as msmp_edge_aggregate_projected_f32 (p, q given: the tile's P / Q rows are staged in
 * LDS) or, with p == q == NULL, as msmp_node_project_f32 + msmp_edge_aggregate_projected_f32 in ONE launch: the tile's h / u /
 * pos / vars rows are staged in LDS, P and Q of the tile's nodes are computed there (halo nodes recomputed per tile) and never
 * touch HBM.  feat (may be NULL): the packed [u | pos | vars] rows of msmp_pack_node_features_f32; they are the same for every
 * layer of a forward, so packing them once saves each layer's tile staging the scalar loads of those columns. */
int msmp_node_feature_stride(int tw, int nv);
/* Feature preparation of Solver.forward (experiments/models_gnn.py:1325-1352; models_gnn2D.py:104-116) in one launch: x [N,tw] and
 * pos [N,2] = (t, x) in float32 or float64 (*_f64 flags), n_cols per-node parameter columns [N] (`cols`, `col_f64`, host arrays)
 * each divided by col_div (1 for the boundary-condition flags) ->  u = float(x), pos_x = float(pos[:,1] / L),
 * pos_t = float(pos[:,0] / tmax), vars [N, 1 + n_cols] = [pos_t | cols / div], and, if feat_out != NULL, the packed rows of
 * msmp_pack_node_features_f32.  Divisions are evaluated in the input's dtype, like the tensor expressions they replace. */
int msmp_prepare_nodes(const void* x, int x_f64, const void* pos, int pos_f64, int64_t n_nodes, int tw, double L, double tmax,
                       int n_cols, const void* const* cols, const int* col_f64, const double* col_div, float* u_out,
                       float* pos_x_out, float* pos_t_out, float* vars_out, float* feat_out, msmp_stream_t stream);
int msmp_pack_node_features_f32(const float* u, const float* pos, const float* vars, int64_t n_nodes, int tw, int nv,
                                float* feat_out, msmp_stream_t stream);
int msmp_edge_aggregate_tiled_f32(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                                  const float* p, const float* q, const int32_t* rowptr, const msmp_tiles_t* tiles, int64_t n_nodes,
                                  int64_t n_edges, int tw, int nv, const float* packed, float* agg_out, msmp_stream_t stream);

/* L2  PyG aggr='mean' (torch_scatter scatter-mean; experiments/models_gnn.py:42,107):
 *   agg[i] = sum_{e in CSR row i} msg[e] / max(deg_i, 1), fixed summation order (CSR order). */
int msmp_scatter_mean_f32(const float* msg, const int32_t* rowptr, int64_t n_nodes,
                          float* agg_out, msmp_stream_t stream);

/* L3  update(): experiments/models_gnn.py:77-86 (MSMP_LAYER_RESIDUAL_SWISH) / 140-149 (MSMP_LAYER_LIN)
 * on [h, agg, vars]. */
int msmp_node_update_f32(const float* h, const float* agg, const float* vars, int64_t n_nodes, int nv,
                         const float* packed, int mode, float* out, msmp_stream_t stream);

/* L4  torch_geometric.nn.InstanceNorm(128) (affine=False, no running stats), experiments/models_gnn.py:
 * 59,66,122,129: per graph g and channel (x-mean)/sqrt(biased var + eps). */
/* Node tail of one layer in one launch: update head(s) + InstanceNorm (+ gated blend), for batches whose graphs have at
 * most 128 nodes (MSMP_ERR_UNSUPPORTED otherwise, and on the fp32-MFMA path: chain the piecewise entry points there).
 *   packed_gate == NULL: out = InstanceNorm(update(h, agg_main))                 GNN_Layer / GNN_LayerLin.forward tail,
 *                                                                                experiments/models_gnn.py:61-67,80-86 / 124-130,143-149
 *   packed_gate != NULL: out = (1 - tau) h + tau Swish(IN(update_main)),  tau = sigmoid(IN(update_gate))       :1366-1368
 * agg_* [N,128] are the mean-aggregated messages of the respective head (msmp_edge_aggregate*_f32); graph_ptr
 * [n_graphs+1]; mode as in msmp_node_update_f32.  The pre-norm tensors stay in registers. */
int msmp_node_tail_f32(const float* h, const float* agg_main, const float* agg_gate, const float* vars,
                       const int32_t* graph_ptr, int64_t n_nodes, int64_t n_graphs, int max_graph_nodes, int nv,
                       const float* packed_main, const float* packed_gate, int mode, float eps, float* out,
                       msmp_stream_t stream);

/* max_graph_nodes = size of the largest graph (0 if unknown): up to 128 nodes the rows of a graph are read once
 * and kept in registers across the three passes; larger graphs use the generic kernels (same results). */
int msmp_instance_norm_f32(const float* x, const int32_t* graph_ptr, int64_t n_graphs,
                           int max_graph_nodes, float eps, float* out, msmp_stream_t stream);

/* L4+L5  gate blend, experiments/models_gnn.py:1204-1207 / 1365-1368 (models_gnn2D.py:267-269, 438-441):
 *   tau = sigmoid(InstanceNorm(gate_pre));  out = (1-tau)*h + tau*Swish(InstanceNorm(main_pre)). */
int msmp_gate_blend_f32(const float* h, const float* gate_pre, const float* main_pre,
                        const int32_t* graph_ptr, int64_t n_graphs, int max_graph_nodes, float eps,
                        float* out, msmp_stream_t stream);

/* One whole message-passing layer: GNN_Layer.forward / GNN_LayerLin.forward
 * (experiments/models_gnn.py:61-67 / 124-130) = L1 -> L2 -> L3 -> L4; when packed_gate != NULL the
 * gated pair of one iteration of the solver loop (L5) is evaluated and blended.  h_out may not alias h.
 * max_in_degree = largest CSR row length (pass -1 if unknown): when <= 256 the fused L1+L2 kernel is
 * used, otherwise the message tensor goes through the workspace; max_graph_nodes as in
 * msmp_instance_norm_f32.  Workspace size from msmp_mp_layer_workspace_bytes (same max_in_degree).
 * tiles (may be NULL): node tiles of the structure (msmp_build_tiles); with them rows L1 + L2 run on the LDS-staged tile
 * kernel (msmp_edge_aggregate_tiled_f32, per-node projections folded in), without them on the gather kernels; feat (may be
 * NULL) as in msmp_edge_aggregate_tiled_f32. */
size_t msmp_mp_layer_workspace_bytes(int64_t n_nodes, int64_t n_edges, int gated, int max_in_degree);
int msmp_mp_layer_f32(const float* h, const float* u, const float* pos, const float* vars,
                      const float* feat, const int32_t* rowptr, const int32_t* col, const int32_t* tgt,
                      const msmp_tiles_t* tiles, const int32_t* graph_ptr, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                      int max_in_degree, int max_graph_nodes, int tw, int nv, const float* packed_main,
                      const float* packed_gate, int mode,
                      float eps, float* h_out, void* workspace, size_t workspace_bytes,
                      msmp_stream_t stream);

/* The LAST layer (pair) of a 1-D solver with the decoder as the node tail's epilogue ("decoder fusion into
 * the last layer's epilogue"; experiments/models_gnn.py:1365-1375: the loop's last iteration, then output_mlp and the Euler update):
 * h_out is still written (the rows are read back from L2 by the workgroup that wrote them, not from HBM), dec->out receives
 * out = u[:, -1] + cumsum(dt) * Conv1d(8,1,k2)(Swish(Conv1d(1,8,k1,stride s1)(h_out))) (dec->u == NULL: the decoder output alone),
 * bit-identical to msmp_mp_layer_f32 followed by msmp_decoder_f32.  Returns MSMP_ERR_UNSUPPORTED where the fused tail does not
 * apply (time_window != 25, graphs of more than 128 nodes, msmp_tune "split" / "tail" off): call the two entry points then. */
typedef struct {
    const float* w1;             /* output_mlp[0].weight [8,1,k1] */
    const float* b1;             /* output_mlp[0].bias   [8]      */
    const float* w2;             /* output_mlp[2].weight [1,8,k2] */
    const float* b2;             /* output_mlp[2].bias   [1]      */
    const float* u;              /* [N, time_window] or NULL      */
    float dt;
    int32_t time_window;
    float* out;                  /* [N, time_window]              */
} msmp_decoder_t;
int msmp_mp_layer_decode_f32(const float* h, const float* u, const float* pos, const float* vars,
                             const float* feat, const int32_t* rowptr, const int32_t* col, const int32_t* tgt,
                             const msmp_tiles_t* tiles, const int32_t* graph_ptr, int64_t n_nodes, int64_t n_edges, int64_t n_graphs,
                             int max_in_degree, int max_graph_nodes, int tw, int nv, const float* packed_main,
                             const float* packed_gate, int mode,
                             float eps, float* h_out, const msmp_decoder_t* dec, void* workspace, size_t workspace_bytes,
                             msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LEM node encoder (replaces the absent `lem_cuda` extension)
 * ------------------------------------------------------------------------------------------- */
/* Pack LEMcuda's parameters (experiments/models_gnn.py:305-330: weights [3*128, 128+ninp], weights_lin_z
 * [128, 128+ninp], bias [3*128], bias_lin_z [128]; column block 0..127 multiplies the state, the rest
 * the step input) and optionally lemoutput_mlp.{0,2} (:1287-1291; all four NULL to skip) for
 * msmp_lem_encoder_f32.  ninp <= 8. */
int64_t msmp_packed_lem_floats(void);
int msmp_pack_lem_f32(const float* weights, const float* weights_lin_z, const float* bias,
                      const float* bias_lin_z, const float* mlp_w0, const float* mlp_b0,
                      const float* mlp_w1, const float* mlp_b1, int ninp, float* packed_out,
                      msmp_stream_t stream);

/* Row stride (floats) of the step-input tensor for a given ninp: ninp rounded up to even. */
int msmp_lem_input_stride(int ninp);

/* LEM.forward (experiments/models_gnn.py:340-342 -> lem_cuda.forward :290) on xin [N, T, stride]
 * (node-major; row t = the step input torch.cat((pos_x, u_t, variables)) of :1360, zero padded from ninp
 * to stride = msmp_lem_input_stride(ninp)), zero initial states,
 * followed when with_mlp != 0 by lemoutput_mlp (:1363).  h_out [N,128] = all_y[-1] (or the MLP of it).
 * PARITY UNPINNED against lem_cuda (source absent); follows the published LEM cell. */
int msmp_lem_encoder_f32(const float* xin, int64_t n_nodes, int t_len, int ninp, float dt,
                         const float* packed, int with_mlp, float* h_out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder
 * ------------------------------------------------------------------------------------------- */
/* 1-D solver decoder, experiments/models_gnn.py:210-224 (output_mlp for time_window 20 / 25 / 50) and
 * :275-279:  out = u[:, -1:] + cumsum(dt) * Conv1d(8,1,k2)(Swish(Conv1d(1,8,k1,stride)(h[:, None, :]))).
 * w1 [8,1,k1], b1 [8], w2 [1,8,k2], b2 [1] in the reference's Conv1d layouts; h [N,128]; u, out [N,tw].
 * u == NULL: out = the decoder output alone (no Euler update), what MSSMP_PDE_Solver_sub returns (:1679-1682). */
int msmp_decoder_f32(const float* h, const float* u, int64_t n_nodes, int tw, const float* w1,
                     const float* b1, const float* w2, const float* b2, float dt, float* out,
                     msmp_stream_t stream);

/* *2D solver decoder, experiments/models_gnn2D.py:79-88 (output_mlp for time_window 25 / 50) and :125-141 after
 * double_mlp:  out = u + cumsum(dt) * Conv1d(8,2,k2)(Swish(Conv1d(2,8,k1,stride)(hd))), hd [N,2,128] = double_mlp(h);
 * w1 [8,2,k1], b1 [8], w2 [2,8,k2], b2 [2]; u, out [N, 2*tw] (component-major). */
int msmp_decoder2d_f32(const float* hd, const float* u, int64_t n_nodes, int tw, const float* w1,
                       const float* b1, const float* w2, const float* b2, float dt, float* out,
                       msmp_stream_t stream);

/* `double_mlp` of the *2D solver classes, experiments/models_gnn2D.py:66-70 (nn.Linear(128, 256) + Swish; the Unflatten is a view):
 * out [rows, n_out] = Swish(x [rows, k] w^T + bias), w [n_out, k] in nn.Linear's layout; n_out a multiple of 128, k a multiple of 4 up to 288.
 * fp32-exact products on the bf16 matrix pipe.  workspace: msmp_linear_swish_workspace_bytes(k, n_out) bytes (0 = unsupported sizes). */
size_t msmp_linear_swish_workspace_bytes(int k, int n_out);
int msmp_linear_swish_f32(const float* x, int64_t rows, int k, const float* w, const float* bias, int n_out, float* out,
                          void* workspace, size_t workspace_bytes, msmp_stream_t stream);

/* msmp_lem_encoder_f32 with the step inputs assembled in the kernel from the node arrays (no [N, T, ninp] tensor in HBM):
 *   two_d = 0: x_t = [pos_x, u_t, variables]                                experiments/models_gnn.py:1357-1360, ninp = 2 + nv
 *   two_d = 1: x_t = [pos_x, u_t, u_{tw+t}, dt_cum_t + pos_t, variables[1:]]  experiments/models_gnn2D.py:429-433, ninp = 3 + nv
 * u [N, tw] ([N, 2 tw] for two_d), pos_x / pos_t [N], vars [N, nv] (column 0 = pos_t), dt_cum [tw] = cumsum(pde.dt);
 * `packed` from msmp_pack_lem_f32 with that ninp.  MSMP_ERR_UNSUPPORTED unless the weight-stationary edition is selected. */
int msmp_lem_encoder_nodes_f32(const float* u, const float* pos_x, const float* pos_t, const float* vars, const float* dt_cum,
                               int64_t n_nodes, int tw, int nv, int two_d, float dt, const float* packed, int with_mlp,
                               float* h_out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LEM encoder, training pair (replaces lem_cuda.forward / lem_cuda.backward as LEMFunction
 * uses them, experiments/models_gnn.py:285-302)
 * ------------------------------------------------------------------------------------------- */
/* LEMFunction.forward (:287-295): the recurrence of msmp_lem_encoder_f32 (no lemoutput_mlp, exact-fp32 MFMA) that also
 * saves what the backward needs, as the reference saves all_X / all_X2 / all_multi_scales / all_lin_new_z_state:
 * saved [6][N][T][128] floats (msmp_lem_saved_floats) = dt*sigmoid(g2), tanh(g3), dt*sigmoid(g1), tanh(lin), y_{t-1} (the state
 * entering step t; y0 at t = 0), z_t.
 * xin / packed as for msmp_lem_encoder_f32; y_out [N,128] = all_y[-1]. */
int64_t msmp_lem_saved_floats(int64_t n_nodes, int t_len);
/* y0 / z0 [N,128]: the initial states (LEMcuda.forward's `states`, :325-332; both NULL = zeros); z_out [N,128] = all_z[-1] (or
 * NULL); saved may be NULL when no backward follows (the state-carrying LEMS of the `Save` variants under no_grad, :345-362). */
int msmp_lem_train_fwd_f32(const float* xin, int64_t n_nodes, int t_len, int ninp, float dt, const float* packed,
                           const float* y0, const float* z0, float* saved, float* y_out, float* z_out, msmp_stream_t stream);
/* The transposed recurrent blocks of weights / weights_lin_z for the backward kernel (16 chunks of [128][32]). */
int64_t msmp_packed_lem_bwd_floats(void);
int msmp_pack_lem_bwd_f32(const float* weights, const float* weights_lin_z, int ninp, float* packed_out,
                          msmp_stream_t stream);
/* LEMFunction.backward (:296-302), the back-propagation through time: grad_y [N,128] = dL/d all_y[-1] ->
 * dg_out [N][T][512] = dL/d(pre-activations) per node and step, columns (g1 | g2 | g3 | lin) in the row order of
 * `weights` then `weights_lin_z`.  The parameter gradients are GEMMs of it over the N*T rows:
 *   d weights = dg[:, :384]^T [y_{t-1} | x_t],  d weights_lin_z = dg[:, 384:]^T [z_t | x_t],  d bias* = column sums
 * (y, z = planes 4, 5 of `saved`; y_{-1} = y0, the forward's initial state, or 0).  Like the reference's use of it, no gradient
 * of the step inputs is produced (:300-302), nor of the initial states (they are carried constants, :350-353). */
int msmp_lem_train_bwd_f32(const float* grad_y, const float* saved, const float* y0, const float* z0, int64_t n_nodes, int t_len,
                           float dt, const float* packed_bwd, float* dg_out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training backward of a message-passing layer: the non-GEMM pieces.  The host layer
 * re-evaluates the layer in materialised form with library GEMMs (the reference does the same through autograd over
 * experiments/models_gnn.py:61-149) and calls these between them.
 * ------------------------------------------------------------------------------------------- */
/* The backward of msmp_mp_layer_f32 in one call: given grad_out = dL/d h_out it recomputes the layer (or the gated pair)
 * from its inputs in materialised form and returns dh_out = dL/dh [N,128] and the gradients of the eight parameters of each
 * layer (what torch.autograd produces over experiments/models_gnn.py:61-149 and the blend :1204-1207).  params_* / grads_*:
 * arrays of 8 device pointers in the order message_net_1.0.weight, .bias, message_net_2.0.weight, .bias, update_net_1.0.weight,
 * .bias, update_net_2.0.weight, .bias, in the reference's [out, in] layouts (NOT the packed blobs); gate arrays NULL for a
 * single layer.  u, pos, vars carry no gradient.  GEMMs with edge- / node-sized outputs run on rocBLAS (looked up in the
 * process at run time: MSMP_ERR_UNSUPPORTED if librocblas cannot be loaded).  src_rowptr [N+1] / src_perm [E] (both or
 * neither): the CSR edge ids regrouped by SOURCE node, ascending inside a source; with them dh's source-side scatter runs in
 * that fixed order (bitwise reproducible gradients), without them it uses float atomics.  tw + 1 + nv <= 64.
 * Workspace: msmp_mp_layer_bwd_workspace_bytes (0 for invalid sizes). */
size_t msmp_mp_layer_bwd_workspace_bytes(int64_t n_nodes, int64_t n_edges, int tw, int nv, int gated);
int msmp_mp_layer_bwd_f32(const float* grad_out, const float* h, const float* u, const float* pos, const float* vars,
                          const int32_t* rowptr, const int32_t* col, const int32_t* tgt, const int32_t* src_rowptr,
                          const int32_t* src_perm, const int32_t* graph_ptr,
                          int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int tw, int nv,
                          const float* const* params_main, const float* const* params_gate, int mode, float eps,
                          float* dh_out, float* const* grads_main, float* const* grads_gate, void* workspace,
                          size_t workspace_bytes, msmp_stream_t stream);

/* Fused AdamW step over all parameter tensors (experiments/train.py:410: optim.AdamW(model.parameters(), lr=args.lr); PyTorch
 * defaults betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2): arrays of n_tensors device pointers (parameter, gradient, first and
 * second moment: fp32, contiguous) and element counts (host array); `step` = 1, 2, ... (bias correction).  One launch per 48
 * tensors.  Same update rule and order of operations as torch.optim.AdamW. */
int msmp_adamw_f32(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, msmp_stream_t stream);
/* The same step for a captured training iteration (a hipGraph replays fixed kernel arguments): the step count lives in device
 * memory (step_dev[0], int64, advanced by this call) and so does the learning rate (lr_dev[0]; a scheduler's new value is a
 * host-to-device copy between replays).  Bias corrections are computed on the device from step_dev. */
int msmp_adamw_capturable_f32(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev, float beta1, float beta2,
                              float eps, float weight_decay, int64_t* step_dev, msmp_stream_t stream);

/* Deterministic reductions for the PyTorch-side pieces of a training iteration (bias gradients of the encoder / decoder, the
 * loss of experiments/train_helper.py:125-141): fixed summation order, two kernel launches, no atomics, no memset -- safe inside a
 * captured (hipGraph) training step.  workspace >= msmp_reduce_workspace_bytes(cols) (cols = 1 for the scalar sum).
 *   msmp_colsum_f32:    out[j] = sum over rows r and columns c in [j group, (j + 1) group) of x[r][c]   (x [rows, cols] row-major)
 *   msmp_sqerr_sum_f32: out[0] = sum_i (a[i] - b[i])^2 */
size_t msmp_reduce_workspace_bytes(int cols);
int msmp_colsum_f32(const float* x, int64_t rows, int cols, int group, float* out, void* workspace, size_t workspace_bytes,
                    msmp_stream_t stream);
int msmp_sqerr_sum_f32(const float* a, const float* b, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                       msmp_stream_t stream);

/* The per-edge input of message_net_1 (models_gnn.py:69-75): out[e] = cat(h[i], h[j], u[i]-u[j], pos[i]-pos[j], vars[i]),
 * i = tgt[e], j = col[e]; out [E, ld] with ld >= 256 + tw + 1 + nv a multiple of 4 (columns past the concat are not written). */
int msmp_edge_concat_f32(const float* h, const float* u, const float* pos, const float* vars, const int32_t* tgt,
                         const int32_t* col, int64_t n_edges, int tw, int nv, int ld, float* out, msmp_stream_t stream);
/* Backward of aggr='mean' (:42,107) fused with the derivative of message_net_2's Swish:
 *   out[e] = dagg[tgt[e]] / max(deg(tgt[e]), 1) * Swish'(a2[e]),  a2 [E,128] the pre-activation of message_net_2. */
int msmp_mean_bwd_dswish_f32(const float* dagg, const int32_t* rowptr, const int32_t* tgt, const float* a2, int64_t n_edges,
                             float* out, msmp_stream_t stream);
/* Backward of msmp_instance_norm_f32 (PyG InstanceNorm, :59,66): x the normalised tensor's input, grad_y = dL/dy. */
int msmp_instance_norm_bwd_f32(const float* x, const float* grad_y, const int32_t* graph_ptr, int64_t n_graphs, float eps,
                               float* dx_out, msmp_stream_t stream);
/* Backward of msmp_gate_blend_f32 (:1204-1207) through both InstanceNorms: grad_out = dL/d out ->
 * d_gate_pre, d_main_pre (gradients of the two pre-norm tensors) and dh_out = the direct (1 - tau) path to h. */
int msmp_gate_blend_bwd_f32(const float* grad_out, const float* h, const float* gate_pre, const float* main_pre,
                            const int32_t* graph_ptr, int64_t n_graphs, float eps, float* d_gate_pre, float* d_main_pre,
                            float* dh_out, msmp_stream_t stream);

/* Weight and bias gradients of up to 8 linear layers in one call (two launches):
 *   out_w[i] [128, k2] = a[i]^T b[i][:, :k2],  out_b[i] [128] = column sums of a[i];  a[i] [rows, 128] (row stride lda >= 128) =
 *   dL/d(pre-activation), b[i] [rows, k2] (row stride ldb >= k2, k2 <= 319) = that layer's input.  Exact-fp32 MFMA partial
 *   products over row splits, summed in a fixed order (deterministic).  workspace: msmp_grad_weights_workspace_floats floats. */
int64_t msmp_grad_weights_workspace_floats(int n_jobs, const int64_t* rows, const int* k2);
int msmp_grad_weights_f32(int n_jobs, const float* const* a, const float* const* b, const int64_t* rows, const int* lda,
                          const int* ldb, const int* k2, float* const* out_w, float* const* out_b, float* workspace,
                          int64_t workspace_floats, msmp_stream_t stream);
/* The same with B given as the virtual column concatenation [b | b2] (columns >= ksplit[i] of job i come from b2[i], row stride ldb2[i]):
 * the LEM weight gradients multiply d gates with [y_{t-1} ; x_t] and [z_t ; x_t], which live in two tensors (lem_cuda.backward,
 * experiments/models_gnn.py:296-302); b2[i] == NULL: the job is as in msmp_grad_weights_f32. */
int msmp_grad_weights_cat_f32(int n_jobs, const float* const* a, const float* const* b, const float* const* b2, const int64_t* rows,
                              const int* lda, const int* ldb, const int* ldb2, const int* ksplit, const int* k2, float* const* out_w,
                              float* const* out_b, float* workspace, int64_t workspace_floats, msmp_stream_t stream);

/* Two-layer node MLP  out = Swish(W2 Swish(W1 x + b1) + b2)  in one launch: the `embedding_mlp` encoder of the LEM-free
 * solver classes (experiments/models_gnn.py:196-201 called at :269-270; models_gnn2D.py:66-71 called at :119-120).
 * w1 [128, k_in] (k_in = in_features of the first Linear <= 128), w2 [128,128], b* [128], reference layout.
 * x: [N, msmp_mlp2_input_stride(k_in)] float32, the concatenated node input [u | pos_x | variables] with rows zero-padded
 * to that stride (a multiple of 32 floats); out [N,128]. */
int64_t msmp_packed_mlp2_floats(int k_in);
int msmp_mlp2_input_stride(int k_in);
int msmp_pack_mlp2_f32(const float* w1, const float* b1, const float* w2, const float* b2, int k_in, float* packed_out,
                       msmp_stream_t stream);
int msmp_mlp2_swish_f32(const float* x, int64_t n_nodes, int k_in, const float* packed, float* out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Width-generic layer pieces: the GLU classes (hidden width 164; experiments/models_gnn.py:1379-1523 MP_PDE_SolverLEMLinGatedGLU,
 * models_gnn2D.py:1198-1366 MP_PDE_Solver2DLEMLinGatedGLU; train.py names 'MSGMP-PDE', 'MSGMP-PDE2D').  The same GNN_LayerLin
 * (message :132-138, mean :107, update :140-149, InstanceNorm :129, gated blend :1486-1489) evaluated from HBM-bound kernels around
 * a general fp32-exact row GEMM; tensors are row-major with row stride ld (a multiple of 4, >= width; padding columns stay 0).
 * ------------------------------------------------------------------------------------------- */
/* out[rows, 0 : 128 ceil(n_out / 128)] = f(x[rows, 0:k] w[n_out, k]^T + bias): mode 0 identity, 1 Swish, 2 out += x w^T (no bias). */
size_t msmp_linear_workspace_bytes(int k, int n_out);
int msmp_linear_f32(const float* x, int ldx, int64_t rows, int k, const float* w, int ldw, const float* bias, int n_out, int mode,
                    float* out, int ld_out, void* workspace, size_t workspace_bytes, msmp_stream_t stream);
/* out[e] = Swish(p[tgt[e]] + q[col[e]])  (message_net_1 factorised per node, then its Swish) */
int msmp_wide_gather_swish_f32(const float* p, const float* q, const int32_t* tgt, const int32_t* col, int64_t n_edges, int width, int ld,
                               float* out, msmp_stream_t stream);
/* PyG aggr='mean' over CSR rows, any width */
int msmp_wide_scatter_mean_f32(const float* msg, const int32_t* rowptr, int64_t n_nodes, int width, int ld, float* agg_out, msmp_stream_t stream);
int msmp_wide_swish_f32(const float* x, int64_t n_floats, float* out, msmp_stream_t stream);
/* The pointwise halves of one LEM time step at any hidden width (the GLU classes' encoder; the cell of experiments/models_gnn.py:285-342,
 * SURVEY 8c), around the two recurrent GEMMs:  g [N, 3 width] = [y, x_t] W^T + b  ->  dtbar_out = dt sigmoid(g1),
 * z <- (1 - dt sigmoid(g2)) z + dt sigmoid(g2) tanh(g3);  then with lin [N, width] = [z, x_t] Wz^T + bz:  y <- (1 - dtbar) y + dtbar tanh(lin). */
int msmp_wide_lem_z_f32(const float* g, int64_t n_nodes, int width, float dt, float* z, float* dtbar_out, msmp_stream_t stream);
int msmp_wide_lem_y_f32(const float* lin, const float* dtbar, int64_t n_floats, float* y, msmp_stream_t stream);
/* gate_pre == NULL: out = InstanceNorm(main_pre); else out = (1 - tau) h + tau Swish(IN(main_pre)), tau = sigmoid(IN(gate_pre)) */
int msmp_wide_norm_blend_f32(const float* h, const float* gate_pre, const float* main_pre, const int32_t* graph_ptr, int64_t n_graphs,
                             int width, int ld, float eps, float* out, msmp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * In-library kernel timing (measurement aid for bench.py; off by default, not part of the data path)
 * When enabled, every launch of the named kernel family is bracketed by hipEvents recorded on the
 * launch stream.  msmp_timing_read synchronises on the recorded events and returns the number of
 * launches and their summed device time since the last reset.
 * ------------------------------------------------------------------------------------------- */
#define MSMP_K_EDGE_MLP     0
#define MSMP_K_SCATTER_MEAN 1
#define MSMP_K_NODE_UPDATE  2
#define MSMP_K_NORM         3   /* instance_norm and gate_blend */
#define MSMP_K_LEM          4
#define MSMP_K_NODE_PROJ    5
#define MSMP_K_DECODER      6
#define MSMP_K_COUNT        7
int msmp_timing_enable(int kernel_mask);   /* bit k enables family k; 0 disables all */
int msmp_timing_reset(void);
int msmp_timing_read(int kernel, int64_t* launches_out, double* total_ms_out);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* MSMP_PDE_H */
