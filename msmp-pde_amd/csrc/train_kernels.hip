// Glue kernels of the training backward of a message-passing layer.  The backward pass
// re-evaluates a layer in materialised form -- every GEMM (data and weight gradients, K = E or N rows) is a plain
// rocBLAS call made by the host layer -- and these kernels are everything between the GEMMs that is not a single
// elementwise library op:
//   msmp_edge_concat_f32          the per-edge input of message_net_1 (models_gnn.py:69-75: cat(x_i, x_j, u_i - u_j, pos_i - pos_j, variables_i))
//   msmp_mean_bwd_dswish_f32      backward of aggr='mean' (:42,107) fused with the Swish' of message_net_2
//   msmp_instance_norm_bwd_f32    backward of PyG InstanceNorm (:59,66; affine=False, biased variance)
//   msmp_gate_blend_bwd_f32       backward of the gated blend (:1204-1207) through both InstanceNorms in one launch
// All fp32, HBM-bound, one pass per tensor where the per-graph statistics allow it.
#include "graph_norm.h"

namespace msmp {

// one wave per edge: lanes 0..31 copy h[tgt] (16 B each), lanes 32..63 copy h[src]; then the tw + 1 + nv tail columns
__global__ __launch_bounds__(256) void edge_concat_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                          const float* __restrict__ pos, const float* __restrict__ vars,
                                                          const int* __restrict__ tgt, const int* __restrict__ col, long n_edges,
                                                          int tw, int nv, int ld, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    for (long e = wave; e < n_edges; e += n_waves) {
        const int i = tgt[e], j = col[e];
        float* row = out + (size_t)e * ld;
        const int node = lane < 32 ? i : j;
        *reinterpret_cast<f32x4*>(row + 4 * lane) = *reinterpret_cast<const f32x4*>(h + (size_t)node * H + 4 * (lane & 31));
        if (lane < tw) row[2 * H + lane] = u[(size_t)i * tw + lane] - u[(size_t)j * tw + lane];
        else if (lane == tw) row[2 * H + tw] = pos[i] - pos[j];
        else if (lane < tw + 1 + nv) row[2 * H + lane] = vars[(size_t)i * nv + (lane - tw - 1)];
    }
}

__device__ __forceinline__ float dswish(float x) {        // d/dx x sigmoid(x)
    const float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}

// da2[e] = dagg[tgt[e]] / max(deg, 1) * Swish'(a2[e]);  thread = (edge, 16-B channel group)
__global__ __launch_bounds__(256) void mean_bwd_dswish_kernel(const float* __restrict__ dagg, const int* __restrict__ rowptr,
                                                              const int* __restrict__ tgt, const float* __restrict__ a2,
                                                              long n_edges, float* __restrict__ out) {
    const long total = n_edges * (H / 4);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long e = p >> 5;
        const int cg = (int)(p & 31), i = tgt[e];
        const float inv = 1.0f / (float)max(rowptr[i + 1] - rowptr[i], 1);
        const f32x4 g = reinterpret_cast<const f32x4*>(dagg)[(size_t)i * (H / 4) + cg];
        const f32x4 a = reinterpret_cast<const f32x4*>(a2)[p];
        f32x4 r;
#pragma unroll
        for (int m = 0; m < 4; ++m) r[m] = g[m] * inv * dswish(a[m]);
        reinterpret_cast<f32x4*>(out)[p] = r;
    }
}

// y = (x - mean) rstd per graph and channel:  dx = rstd (g - mean_graph(g) - y mean_graph(g y))
__device__ __forceinline__ void norm_bwd_sums(const float* __restrict__ x, const float* __restrict__ g, int n0, int n1, int cg, int rs,
                                              f32x4* red, f32x4 mean, f32x4 rstd, f32x4& g_mean, f32x4& gy_mean) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int r = n0 + rs; r < n1; r += 8) {
        const size_t o = (size_t)r * (H / 4) + cg;
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[o];
        s += gv;
        q += gv * ((reinterpret_cast<const f32x4*>(x)[o] - mean) * rstd);
    }
    const float inv = 1.0f / (float)max(n1 - n0, 1);
    g_mean = block_colsum(s, red, cg, rs) * inv;
    gy_mean = block_colsum(q, red, cg, rs) * inv;
}

__global__ __launch_bounds__(256) void instance_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                const int* __restrict__ graph_ptr, float eps, float* __restrict__ dx) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    f32x4 mean, rstd, gm, gym;
    graph_stats(x, n0, n1, cg, rs, red, eps, mean, rstd);
    norm_bwd_sums(x, g, n0, n1, cg, rs, red, mean, rstd, gm, gym);
    for (int r = n0 + rs; r < n1; r += 8) {
        const size_t o = (size_t)r * (H / 4) + cg;
        const f32x4 y = (reinterpret_cast<const f32x4*>(x)[o] - mean) * rstd;
        reinterpret_cast<f32x4*>(dx)[o] = rstd * (reinterpret_cast<const f32x4*>(g)[o] - gm - y * gym);
    }
}

// out = (1 - tau) h + tau s,  tau = sigmoid(IN(gate_pre)),  s = Swish(IN(main_pre)):
//   d IN(gate) = g (s - h) tau (1 - tau);  d IN(main) = g tau Swish'(IN(main));  dh = g (1 - tau);  then both InstanceNorm backwards.
// The two upstream gradients are formed on the fly in each pass (never stored): 3 passes over g, h, gate_pre, main_pre.
__device__ __forceinline__ void blend_grads(f32x4 g, f32x4 hv, f32x4 yg, f32x4 ym, f32x4& dyg, f32x4& dym, f32x4& dh) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float tau = sigmoidf_(yg[m]);
        dyg[m] = g[m] * (swishf(ym[m]) - hv[m]) * tau * (1.0f - tau);
        dym[m] = g[m] * tau * dswish(ym[m]);
        dh[m] = g[m] * (1.0f - tau);
    }
}

__global__ __launch_bounds__(256) void gate_blend_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ h,
                                                             const float* __restrict__ gate, const float* __restrict__ mainp,
                                                             const int* __restrict__ graph_ptr, float eps, float* __restrict__ d_gate,
                                                             float* __restrict__ d_main, float* __restrict__ dh_out) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    f32x4 gm, gr, mm, mr;
    graph_stats(gate, n0, n1, cg, rs, red, eps, gm, gr);
    graph_stats(mainp, n0, n1, cg, rs, red, eps, mm, mr);
    const f32x4* gp = reinterpret_cast<const f32x4*>(gout);
    const f32x4* hp = reinterpret_cast<const f32x4*>(h);
    const f32x4* tp = reinterpret_cast<const f32x4*>(gate);
    const f32x4* mp = reinterpret_cast<const f32x4*>(mainp);
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgy = sg, sm = sg, smy = sg;
    for (int r = n0 + rs; r < n1; r += 8) {
        const size_t o = (size_t)r * (H / 4) + cg;
        const f32x4 yg = (tp[o] - gm) * gr, ym = (mp[o] - mm) * mr;
        f32x4 dyg, dym, dh;
        blend_grads(gp[o], hp[o], yg, ym, dyg, dym, dh);
        sg += dyg; sgy += dyg * yg; sm += dym; smy += dym * ym;
    }
    const float inv = 1.0f / (float)max(n1 - n0, 1);
    sg = block_colsum(sg, red, cg, rs) * inv;
    sgy = block_colsum(sgy, red, cg, rs) * inv;
    sm = block_colsum(sm, red, cg, rs) * inv;
    smy = block_colsum(smy, red, cg, rs) * inv;
    for (int r = n0 + rs; r < n1; r += 8) {
        const size_t o = (size_t)r * (H / 4) + cg;
        const f32x4 yg = (tp[o] - gm) * gr, ym = (mp[o] - mm) * mr;
        f32x4 dyg, dym, dh;
        blend_grads(gp[o], hp[o], yg, ym, dyg, dym, dh);
        reinterpret_cast<f32x4*>(d_gate)[o] = gr * (dyg - sg - yg * sgy);
        reinterpret_cast<f32x4*>(d_main)[o] = mr * (dym - sm - ym * smy);
        reinterpret_cast<f32x4*>(dh_out)[o] = dh;
    }
}

}  // namespace msmp

using namespace msmp;

extern "C" int msmp_edge_concat_f32(const float* h, const float* u, const float* pos, const float* vars, const int32_t* tgt,
                                    const int32_t* col, int64_t n_edges, int tw, int nv, int ld, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && tgt && col && out, MSMP_ERR_ARG, "msmp_edge_concat_f32: null pointer");
    MSMP_REQUIRE(n_edges >= 0 && n_edges < (1L << 31) && tw >= 1 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG, "msmp_edge_concat_f32: bad sizes");
    MSMP_REQUIRE(tw + 1 + nv <= 64, MSMP_ERR_UNSUPPORTED, "msmp_edge_concat_f32: tw + 1 + nv = %d > 64", tw + 1 + nv);
    MSMP_REQUIRE(ld >= 2 * H + tw + 1 + nv && ld % 4 == 0, MSMP_ERR_ARG, "msmp_edge_concat_f32: row stride %d (need a multiple of 4 >= %d)", ld,
                 2 * H + tw + 1 + nv);
    if (n_edges == 0) return MSMP_OK;
    const unsigned grid = (unsigned)((n_edges + 3) / 4 < 16384 ? (n_edges + 3) / 4 : 16384);
    hipLaunchKernelGGL(edge_concat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, h, u, pos, vars, tgt, col, (long)n_edges, tw, nv,
                       ld, out);
    return check_launch("edge_concat_kernel");
}

extern "C" int msmp_mean_bwd_dswish_f32(const float* dagg, const int32_t* rowptr, const int32_t* tgt, const float* a2, int64_t n_edges,
                                        float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(dagg && rowptr && tgt && a2 && out, MSMP_ERR_ARG, "msmp_mean_bwd_dswish_f32: null pointer");
    MSMP_REQUIRE(n_edges >= 0 && n_edges < (1L << 31), MSMP_ERR_ARG, "msmp_mean_bwd_dswish_f32: bad sizes");
    if (n_edges == 0) return MSMP_OK;
    const long blocks = (n_edges * (H / 4) + 255) / 256;
    hipLaunchKernelGGL(mean_bwd_dswish_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, dagg,
                       rowptr, tgt, a2, (long)n_edges, out);
    return check_launch("mean_bwd_dswish_kernel");
}

extern "C" int msmp_instance_norm_bwd_f32(const float* x, const float* grad_y, const int32_t* graph_ptr, int64_t n_graphs, float eps,
                                          float* dx_out, msmp_stream_t stream) {
    MSMP_REQUIRE(x && grad_y && graph_ptr && dx_out, MSMP_ERR_ARG, "msmp_instance_norm_bwd_f32: null pointer");
    MSMP_REQUIRE(n_graphs > 0 && n_graphs < (1L << 31), MSMP_ERR_ARG, "msmp_instance_norm_bwd_f32: bad n_graphs");
    hipLaunchKernelGGL(instance_norm_bwd_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, grad_y, graph_ptr, eps,
                       dx_out);
    return check_launch("instance_norm_bwd_kernel");
}

extern "C" int msmp_gate_blend_bwd_f32(const float* grad_out, const float* h, const float* gate_pre, const float* main_pre,
                                       const int32_t* graph_ptr, int64_t n_graphs, float eps, float* d_gate_pre, float* d_main_pre,
                                       float* dh_out, msmp_stream_t stream) {
    MSMP_REQUIRE(grad_out && h && gate_pre && main_pre && graph_ptr && d_gate_pre && d_main_pre && dh_out, MSMP_ERR_ARG,
                 "msmp_gate_blend_bwd_f32: null pointer");
    MSMP_REQUIRE(n_graphs > 0 && n_graphs < (1L << 31), MSMP_ERR_ARG, "msmp_gate_blend_bwd_f32: bad n_graphs");
    hipLaunchKernelGGL(gate_blend_bwd_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, grad_out, h, gate_pre,
                       main_pre, graph_ptr, eps, d_gate_pre, d_main_pre, dh_out);
    return check_launch("gate_blend_bwd_kernel");
}

// ----------------------------------------------------------------------------------------------
// Weight gradients:  dW[m][n] = sum_r A[r][m] B[r][n]  and  db[m] = sum_r A[r][m]  for up to 8 (A, B) pairs in one call
// (A [R,128] (row stride lda) = gradient of a pre-activation, B [R, >= k2] = the input of that linear layer, R = E or N rows).
// These GEMMs are 128 x k2 outputs with a reduction over R >> 1000 rows: the library runs them on 36 workgroups
// (63 us each at R = 9 408); here the rows are split over workgroups (<= 128 splits per pair; exact-fp32 MFMA partial products, the bias column
// as a virtual all-ones column k2 of B), and a second kernel sums the partials in a fixed order (deterministic).
// ----------------------------------------------------------------------------------------------
namespace msmp {

constexpr int GW_MAX_JOBS = 8;
struct GradWeightJob {
    const float* a;      // [rows, lda], columns 0..127 used
    const float* b;      // [rows, ldb], columns 0..k2-1 used
    float* out;          // [128, k2 + 1]: dW | db
    float* partial;      // [splits][128][32 * nt]
    int rows, lda, ldb, k2, nt, rows_per_split, splits, first_block;
};
struct GradWeightArgs {
    GradWeightJob job[GW_MAX_JOBS];
    int n_jobs;
};

template <int NT>
__device__ __forceinline__ void grad_weight_body(const GradWeightJob& j, int split, int wave, int lane) {
    const int m = lane & 31, kk = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int r0 = split * j.rows_per_split, r1 = min(r0 + j.rows_per_split, j.rows);
    // 4 k-steps (8 rows) per iteration with all 4 (1 + NT) loads issued before the first MFMA: the loop is bound by load
    // latency otherwise.  Every lane runs the same trip count; rows past r1 are clamped and contribute 0.
    constexpr int U = 4;
    for (int rb = r0; rb < r1; rb += 2 * U) {
        float av[U], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + 2 * u + kk;
            const bool in = r < r1;
            const int rc = in ? r : r1 - 1;
            av[u] = j.a[(size_t)rc * j.lda + 32 * wave + m];
            av[u] = in ? av[u] : 0.f;
            const float* brow = j.b + (size_t)rc * j.ldb;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 32 * t + m;
                const int cc = c < j.k2 ? c : 0;
                const float b = brow[cc];
                bv[u][t] = c < j.k2 ? b : (c == j.k2 ? 1.0f : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][t], acc[t], 0, 0, 0);
    }
    float* p = j.partial + (size_t)split * H * (32 * NT);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(size_t)(32 * wave + acc_row(r, kk)) * (32 * NT) + 32 * t + m] = acc[t][r];
}

__global__ __launch_bounds__(256) void grad_weight_kernel(GradWeightArgs a) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < GW_MAX_JOBS; ++i)
        if (i < a.n_jobs && (int)blockIdx.x >= a.job[i].first_block) ji = i;
    const GradWeightJob& j = a.job[ji];
    const int split = blockIdx.x - j.first_block, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    switch (j.nt) {
        case 5: grad_weight_body<5>(j, split, wave, lane); break;
        case 9: grad_weight_body<9>(j, split, wave, lane); break;
        default: grad_weight_body<10>(j, split, wave, lane); break;
    }
}

// out[row][col] = sum over splits, in split order; grid.y = job
__global__ __launch_bounds__(256) void grad_weight_reduce_kernel(GradWeightArgs a) {
    const GradWeightJob& j = a.job[blockIdx.y];
    const int w = j.k2 + 1, ldp = 32 * j.nt, total = H * w;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int row = p / w, c = p - row * w;
        const float* src = j.partial + (size_t)row * ldp + c;
        float s = 0.f;
        for (int k = 0; k < j.splits; ++k) s += src[(size_t)k * H * ldp];
        j.out[p] = s;
    }
}

static int gw_tiles(int k2) { const int t = (k2 + 1 + 31) / 32; return t <= 5 ? 5 : t <= 9 ? 9 : t <= 10 ? 10 : -1; }
// rows per workgroup: 128 (8 pairs of a training batch of 16 graphs already make ~350 workgroups), more once a pair would
// exceed 128 splits -- the partial products (splits x 128 x 32 nt floats) are written and re-read by the reduction
static int gw_rows_per_split(int64_t rows) {
    int64_t rps = 128;
    while ((rows + rps - 1) / rps > 128) rps *= 2;
    return (int)rps;
}

}  // namespace msmp

extern "C" int64_t msmp_grad_weights_workspace_floats(int n_jobs, const int64_t* rows, const int* k2) {
    if (n_jobs < 1 || n_jobs > GW_MAX_JOBS || !rows || !k2) return -1;
    int64_t total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const int nt = gw_tiles(k2[i]);
        if (nt < 0 || rows[i] < 1) return -1;
        const int rps = gw_rows_per_split(rows[i]);
        total += ((rows[i] + rps - 1) / rps) * H * 32 * nt;
    }
    return total;
}

extern "C" int msmp_grad_weights_f32(int n_jobs, const float* const* a, const float* const* b, const int64_t* rows, const int* lda,
                                     const int* ldb, const int* k2, float* const* out, float* workspace, int64_t workspace_floats,
                                     msmp_stream_t stream) {
    MSMP_REQUIRE(n_jobs >= 1 && n_jobs <= GW_MAX_JOBS, MSMP_ERR_ARG, "msmp_grad_weights_f32: n_jobs=%d not in 1..%d", n_jobs, GW_MAX_JOBS);
    MSMP_REQUIRE(a && b && rows && lda && ldb && k2 && out && workspace, MSMP_ERR_ARG, "msmp_grad_weights_f32: null pointer");
    GradWeightArgs args;
    args.n_jobs = n_jobs;
    int64_t used = 0;
    int blocks = 0, max_w = 0;
    for (int i = 0; i < n_jobs; ++i) {
        MSMP_REQUIRE(a[i] && b[i] && out[i], MSMP_ERR_ARG, "msmp_grad_weights_f32: null pointer in job %d", i);
        MSMP_REQUIRE(rows[i] >= 1 && rows[i] < (1L << 31) && k2[i] >= 1 && ldb[i] >= k2[i] && lda[i] >= H, MSMP_ERR_ARG, "msmp_grad_weights_f32: bad sizes in job %d", i);
        const int nt = gw_tiles(k2[i]);
        MSMP_REQUIRE(nt > 0, MSMP_ERR_UNSUPPORTED, "msmp_grad_weights_f32: k2=%d > 319", k2[i]);
        GradWeightJob& j = args.job[i];
        j.a = a[i]; j.b = b[i]; j.out = out[i]; j.partial = workspace + used;
        j.rows = (int)rows[i]; j.lda = lda[i]; j.ldb = ldb[i]; j.k2 = k2[i]; j.nt = nt;
        j.rows_per_split = gw_rows_per_split(rows[i]);
        j.splits = (j.rows + j.rows_per_split - 1) / j.rows_per_split;
        j.first_block = blocks;
        blocks += j.splits;
        used += (int64_t)j.splits * H * 32 * nt;
        max_w = k2[i] + 1 > max_w ? k2[i] + 1 : max_w;
    }
    for (int i = n_jobs; i < GW_MAX_JOBS; ++i) args.job[i] = args.job[0];
    MSMP_REQUIRE(used <= workspace_floats, MSMP_ERR_ARG, "msmp_grad_weights_f32: workspace of %ld floats, need %ld", (long)workspace_floats, (long)used);
    hipLaunchKernelGGL(grad_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, args);
    hipLaunchKernelGGL(grad_weight_reduce_kernel, dim3((unsigned)((H * max_w + 255) / 256), (unsigned)n_jobs), dim3(256), 0,
                       (hipStream_t)stream, args);
    return check_launch("grad_weight_kernel");
}
