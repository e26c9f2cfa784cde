// Graph structure on the device (row G1): radius_graph / knn_graph with the reference's float64
// comparisons (bit-exact edge_index) and CSR-by-target construction.
//
// Compiled with -ffp-contract=off: the float64 squared distance must round exactly like
// ((x_i - x_j)**2).sum(-1) on the host (separate multiply and add), or an edge sitting on the
// radius could flip.  Graph construction runs once per rollout, not per step; these kernels are
// O(N * nodes-per-graph) brute force on purpose (graphs have ~100 nodes).
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "msmp_common.h"

namespace msmp {

constexpr int MAX_DIM = 3;
constexpr int MAX_K = 32;

__device__ __forceinline__ int find_graph(const int* __restrict__ graph_ptr, int n_graphs, int node) {
    int lo = 0, hi = n_graphs;   // graph_ptr[lo] <= node < graph_ptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (graph_ptr[mid] <= node) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double dist2(const double* __restrict__ x, int dim, int i, int j) {
    double s = 0.0;
    for (int d = 0; d < dim; ++d) {
        const double df = x[(size_t)i * dim + d] - x[(size_t)j * dim + d];
        s = s + df * df;
    }
    return s;
}

// FILL = false: deg[i] = min(#{j in graph(i), j != i, d2 < r2}, max_nb).  FILL = true: write the pairs.
template <bool FILL>
__global__ void radius_kernel(const double* __restrict__ x, int dim, const int* __restrict__ graph_ptr, int n_graphs,
                              int n_nodes, double r2, int max_nb, int* __restrict__ deg, const int* __restrict__ rowptr,
                              long n_edges, int64_t* __restrict__ edge_index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int g = find_graph(graph_ptr, n_graphs, i);
    const int j0 = graph_ptr[g], j1 = graph_ptr[g + 1];
    int cnt = 0;
    const long base = FILL ? rowptr[i] : 0;
    for (int j = j0; j < j1 && cnt < max_nb; ++j) {
        if (j == i) continue;
        if (dist2(x, dim, i, j) < r2) {
            if (FILL) {
                edge_index[base + cnt] = j;             // row 0: source
                edge_index[n_edges + base + cnt] = i;   // row 1: target
            }
            ++cnt;
        }
    }
    if (!FILL) deg[i] = cnt;
}

__global__ void knn_kernel(const double* __restrict__ x, int dim, const int* __restrict__ graph_ptr, int n_graphs,
                           int n_nodes, int k, long n_edges, const int* __restrict__ rowptr, int64_t* __restrict__ edge_index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int g = find_graph(graph_ptr, n_graphs, i);
    const int j0 = graph_ptr[g], j1 = graph_ptr[g + 1];
    double bd[MAX_K];
    int bj[MAX_K];
    int cnt = 0;
    for (int j = j0; j < j1; ++j) {
        if (j == i) continue;
        const double d = dist2(x, dim, i, j);
        if (cnt == k && !(d < bd[k - 1])) continue;   // ties keep the earlier (lower) index
        int p = cnt < k ? cnt : k - 1;
        while (p > 0 && d < bd[p - 1]) {              // stable insertion: after every entry with bd <= d
            bd[p] = bd[p - 1];
            bj[p] = bj[p - 1];
            --p;
        }
        bd[p] = d;
        bj[p] = j;
        if (cnt < k) ++cnt;
    }
    const long base = rowptr[i];
    for (int p = 0; p < cnt && base + p < n_edges; ++p) {
        edge_index[base + p] = bj[p];
        edge_index[n_edges + base + p] = i;
    }
}

__global__ void knn_degree_kernel(const int* __restrict__ graph_ptr, int n_graphs, int n_nodes, int k, int* __restrict__ deg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int g = find_graph(graph_ptr, n_graphs, i);
    deg[i] = min(k, graph_ptr[g + 1] - graph_ptr[g] - 1);
}

// rowptr[0..n] = exclusive prefix sum of deg[0..n-1], in place on rowptr (deg stored at rowptr[0..n-1]).
// Single workgroup; N is a few 1e5..1e6 and this runs once per graph build.
__global__ __launch_bounds__(1024) void exclusive_scan_inplace_kernel(int* __restrict__ a, int n) {
    __shared__ int part[1024];
    __shared__ int carry_s;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + t;
        const int v = idx < n ? a[idx] : 0;
        part[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        const int carry = carry_s;
        if (idx < n) a[idx] = carry + part[t] - v;
        __syncthreads();
        if (t == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (t == 0) a[n] = carry_s;
}

__global__ void csr_split_kernel(const int64_t* __restrict__ edge_index, long n_edges, int* __restrict__ keys,
                                 int* __restrict__ ids) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    keys[e] = (int)edge_index[n_edges + e];
    ids[e] = (int)e;
}

__global__ void csr_finish_kernel(const int64_t* __restrict__ edge_index, long n_edges, const int* __restrict__ keys_sorted,
                                  const int* __restrict__ ids_sorted, int* __restrict__ col, int* __restrict__ tgt,
                                  int* __restrict__ rowptr, int n_nodes) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int t = keys_sorted[e];
    tgt[e] = t;
    col[e] = (int)edge_index[ids_sorted[e]];
    // rowptr[v] = first slot whose target >= v: slot e opens rows (prev_target, t]
    const int prev = e == 0 ? -1 : keys_sorted[e - 1];
    for (int v = prev + 1; v <= t; ++v) rowptr[v] = (int)e;
    if (e == n_edges - 1)
        for (int v = t + 1; v <= n_nodes; ++v) rowptr[v] = (int)n_edges;
}

__global__ void fill_int_kernel(int* a, long n, int v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t n_edges) {
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const int*, int*, const int*, int*>(
        nullptr, temp, nullptr, nullptr, nullptr, nullptr, (size_t)n_edges, 0, 32, (hipStream_t)0, false);
    return temp;
}

}  // namespace msmp

using namespace msmp;

extern "C" size_t msmp_build_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes) {
    (void)n_nodes;
    if (n_edges <= 0) return 256;
    return 4 * al256((size_t)n_edges * sizeof(int)) + al256(sort_temp_bytes(n_edges)) + 256;
}

extern "C" int msmp_build_csr(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, int32_t* rowptr_out,
                              int32_t* col_out, int32_t* tgt_out, void* workspace, size_t workspace_bytes,
                              msmp_stream_t stream) {
    MSMP_REQUIRE(rowptr_out && n_nodes > 0 && n_nodes < (1L << 31) && n_edges >= 0 && n_edges < (1L << 31), MSMP_ERR_ARG,
                 "msmp_build_csr: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (n_edges == 0) {
        hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((n_nodes + 1 + 255) / 256)), dim3(256), 0, st, rowptr_out,
                           (long)n_nodes + 1, 0);
        return check_launch("fill_int_kernel");
    }
    MSMP_REQUIRE(edge_index && col_out && tgt_out && workspace, MSMP_ERR_ARG, "msmp_build_csr: null pointer");
    MSMP_REQUIRE(workspace_bytes >= msmp_build_csr_workspace_bytes(n_edges, n_nodes), MSMP_ERR_WORKSPACE,
                 "msmp_build_csr: workspace too small");
    const size_t seg = al256((size_t)n_edges * sizeof(int));
    char* ws = (char*)workspace;
    int* keys = (int*)ws;
    int* ids = (int*)(ws + seg);
    int* keys_s = (int*)(ws + 2 * seg);
    int* ids_s = (int*)(ws + 3 * seg);
    void* temp = ws + 4 * seg;
    size_t temp_bytes = sort_temp_bytes(n_edges);
    const unsigned grid = (unsigned)((n_edges + 255) / 256);
    hipLaunchKernelGGL(csr_split_kernel, dim3(grid), dim3(256), 0, st, edge_index, (long)n_edges, keys, ids);
    // stable LSD radix sort by target keeps the original relative order inside each target
    const hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const int*)keys, keys_s, (const int*)ids, ids_s,
                                                   (size_t)n_edges, 0, 32, st, false);
    if (e != hipSuccess) {
        set_error("msmp_build_csr: radix_sort_pairs: %s", hipGetErrorString(e));
        return MSMP_ERR_HIP;
    }
    hipLaunchKernelGGL(csr_finish_kernel, dim3(grid), dim3(256), 0, st, edge_index, (long)n_edges, (const int*)keys_s,
                       (const int*)ids_s, col_out, tgt_out, rowptr_out, (int)n_nodes);
    return check_launch("csr_finish_kernel");
}

extern "C" int msmp_radius_graph_count_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs,
                                           int64_t n_nodes, double r, int max_neighbors, int32_t* rowptr_out,
                                           msmp_stream_t stream) {
    MSMP_REQUIRE(x && graph_ptr && rowptr_out, MSMP_ERR_ARG, "msmp_radius_graph_count_f64: null pointer");
    MSMP_REQUIRE(dim >= 1 && dim <= MAX_DIM && n_graphs > 0 && n_nodes > 0 && n_nodes < (1L << 31) && max_neighbors > 0,
                 MSMP_ERR_ARG, "msmp_radius_graph_count_f64: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(radius_kernel<false>, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, st, x, dim, graph_ptr,
                       (int)n_graphs, (int)n_nodes, r * r, max_neighbors, rowptr_out, (const int*)nullptr, 0L,
                       (int64_t*)nullptr);
    hipLaunchKernelGGL(exclusive_scan_inplace_kernel, dim3(1), dim3(1024), 0, st, rowptr_out, (int)n_nodes);
    return check_launch("radius_kernel<count>");
}

extern "C" int msmp_radius_graph_fill_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs,
                                          int64_t n_nodes, double r, int max_neighbors, const int32_t* rowptr,
                                          int64_t n_edges, int64_t* edge_index_out, msmp_stream_t stream) {
    MSMP_REQUIRE(x && graph_ptr && rowptr, MSMP_ERR_ARG, "msmp_radius_graph_fill_f64: null pointer");
    MSMP_REQUIRE(dim >= 1 && dim <= MAX_DIM && n_graphs > 0 && n_nodes > 0 && n_nodes < (1L << 31) && max_neighbors > 0 &&
                     n_edges >= 0, MSMP_ERR_ARG, "msmp_radius_graph_fill_f64: bad sizes");
    if (n_edges == 0) return MSMP_OK;
    MSMP_REQUIRE(edge_index_out, MSMP_ERR_ARG, "msmp_radius_graph_fill_f64: null output");
    hipLaunchKernelGGL(radius_kernel<true>, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       dim, graph_ptr, (int)n_graphs, (int)n_nodes, r * r, max_neighbors, (int*)nullptr, rowptr,
                       (long)n_edges, edge_index_out);
    return check_launch("radius_kernel<fill>");
}

extern "C" int msmp_knn_graph_f64(const double* x, int dim, const int32_t* graph_ptr, int64_t n_graphs, int64_t n_nodes,
                                  int k, int64_t n_edges, int32_t* rowptr_out, int64_t* edge_index_out,
                                  msmp_stream_t stream) {
    MSMP_REQUIRE(x && graph_ptr && rowptr_out, MSMP_ERR_ARG, "msmp_knn_graph_f64: null pointer");
    MSMP_REQUIRE(dim >= 1 && dim <= MAX_DIM && n_graphs > 0 && n_nodes > 0 && n_nodes < (1L << 31) && n_edges >= 0,
                 MSMP_ERR_ARG, "msmp_knn_graph_f64: bad sizes");
    MSMP_REQUIRE(k >= 1 && k <= MAX_K, MSMP_ERR_UNSUPPORTED, "msmp_knn_graph_f64: k=%d outside 1..%d", k, MAX_K);
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_nodes + 255) / 256);
    hipLaunchKernelGGL(knn_degree_kernel, dim3(grid), dim3(256), 0, st, graph_ptr, (int)n_graphs, (int)n_nodes, k, rowptr_out);
    hipLaunchKernelGGL(exclusive_scan_inplace_kernel, dim3(1), dim3(1024), 0, st, rowptr_out, (int)n_nodes);
    if (n_edges == 0) return check_launch("knn_degree_kernel");
    MSMP_REQUIRE(edge_index_out, MSMP_ERR_ARG, "msmp_knn_graph_f64: null output");
    hipLaunchKernelGGL(knn_kernel, dim3(grid), dim3(256), 0, st, x, dim, graph_ptr, (int)n_graphs, (int)n_nodes, k,
                       (long)n_edges, (const int*)rowptr_out, edge_index_out);
    return check_launch("knn_kernel");
}
