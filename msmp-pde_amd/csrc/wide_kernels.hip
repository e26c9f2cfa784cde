// Width-generic pieces of the message-passing layer: the GLU classes of the reference run the SAME layers at hidden width 164
// (experiments/models_gnn.py:1379-1523 MP_PDE_SolverLEMLinGatedGLU, models_gnn2D.py:1198-1366), which the fused 128-wide kernels
// (four 32-channel MFMA tiles per wave, packed blobs, LDS layouts) are not built for.  Their layer is evaluated from these
// HBM-bound kernels around msmp_linear_f32 (the fp32-exact bf16x3 row GEMM of train_kernels.hip, any K and any number of output
// channels in 128-column groups):
//     P, Q = msmp_linear_f32 on [h | u | pos | vars]            (message_net_1 factorised per node, :132-138)
//     a1   = msmp_wide_gather_swish_f32(P, Q)                    Swish(P[target] + Q[source]) per edge
//     msg  = msmp_linear_f32(a1, message_net_2, Swish)
//     agg  = msmp_wide_scatter_mean_f32(msg)                     PyG aggr = 'mean' (:107), CSR order, no atomics
//     y    = msmp_linear_f32(Swish(msmp_linear_f32([h | agg | vars], update_net_1)), update_net_2)      (:140-149)
//     h'   = msmp_wide_norm_blend_f32                            InstanceNorm (:129) and the gated blend (:1486-1489)
// All tensors are row-major with a row stride `ld` (a multiple of 4 floats, >= width); columns width .. ld-1 are padding that the
// GEMMs keep at zero.
#include "msmp_common.h"

namespace msmp {

// out[e][c] = Swish(p[tgt[e]][c] + q[col[e]][c]): thread = (edge, 4 channels)
__global__ __launch_bounds__(256) void wide_gather_swish_kernel(const float* __restrict__ p, const float* __restrict__ q, const int* __restrict__ tgt,
                                                                const int* __restrict__ col, long n_edges, int ld4, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_edges * ld4) return;
    const long e = i / ld4;
    const int c = (int)(i - e * ld4);
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[(size_t)tgt[e] * ld4 + c];
    const f32x4 b = reinterpret_cast<const f32x4*>(q)[(size_t)col[e] * ld4 + c];
    f32x4 r;
#pragma unroll
    for (int m = 0; m < 4; ++m) r[m] = swishf(a[m] + b[m]);
    reinterpret_cast<f32x4*>(out)[i] = r;
}

// agg[n][c] = mean over the CSR row of n (fixed order): thread = (node, 4 channels)
__global__ __launch_bounds__(256) void wide_scatter_mean_kernel(const float* __restrict__ msg, const int* __restrict__ rowptr, long n_nodes, int ld4,
                                                                float* __restrict__ agg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes * ld4) return;
    const long n = i / ld4;
    const int c = (int)(i - n * ld4);
    const int r0 = rowptr[n], r1 = rowptr[n + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) s += reinterpret_cast<const f32x4*>(msg)[(size_t)r * ld4 + c];
    reinterpret_cast<f32x4*>(agg)[i] = s * (1.0f / (float)max(r1 - r0, 1));
}

// out[n][c] = Swish(x[n][c]) in place-capable form (the hidden units of update_net_1 when its K = 2 W + nv is split over two GEMMs)
__global__ __launch_bounds__(256) void wide_swish_kernel(const float* __restrict__ x, long n4, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 r;
#pragma unroll
    for (int m = 0; m < 4; ++m) r[m] = swishf(a[m]);
    reinterpret_cast<f32x4*>(out)[i] = r;
}

// The LEM cell's pointwise halves at any hidden width (the GLU classes' encoder, experiments/models_gnn.py:285-342 with the published
// cell: SURVEY 8c): the two recurrent GEMMs of a time step stay library calls, everything between them is ONE launch each instead of
// ~20 PyTorch elementwise kernels per step (each a pass over [N, W] .. [N, 3 W]: 46 -> 24 ms per forward at 2048 graphs, W = 164).
//   z-half:  g = [g1 | g2 | g3] [N, 3 W]:  dtbar = dt sigmoid(g1),  z <- (1 - dt sigmoid(g2)) z + dt sigmoid(g2) tanh(g3)
//   y-half:  y <- (1 - dtbar) y + dtbar tanh(lin)
__device__ __forceinline__ float wide_tanh(float x) {           // 1 - 2 / (1 + e^{2x}), exponent clamped like the fused kernels
    const float e = msmp_exp2(fminf(x * 2.88539008177792681472f, 60.f));
    return 1.0f - 2.0f * msmp_rcp(1.0f + e);
}
__global__ __launch_bounds__(256) void wide_lem_z_kernel(const float* __restrict__ g, long n, int w, float dt, float* __restrict__ z,
                                                         float* __restrict__ dtbar) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * w) return;
    const long r = i / w;
    const int c = (int)(i - r * w);
    const float* gr = g + (size_t)r * 3 * w;
    const float d2 = dt * sigmoidf_(gr[w + c]);
    dtbar[i] = dt * sigmoidf_(gr[c]);
    z[i] = (1.0f - d2) * z[i] + d2 * wide_tanh(gr[2 * w + c]);
}
__global__ __launch_bounds__(256) void wide_lem_y_kernel(const float* __restrict__ lin, const float* __restrict__ dtbar, long n, float* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = dtbar[i];
    y[i] = (1.0f - d) * y[i] + d * wide_tanh(lin[i]);
}

// One workgroup per graph, thread = (4-channel group cg = tid % ld4 ..., row slice): PyG InstanceNorm (biased variance, two passes)
// of `main_pre` (and `gate_pre`), then  out = IN(main)   or   out = (1 - tau) h + tau Swish(IN(main)),  tau = sigmoid(IN(gate)).
__global__ __launch_bounds__(256) void wide_norm_blend_kernel(const float* __restrict__ h, const float* __restrict__ gate_pre,
                                                              const float* __restrict__ main_pre, const int* __restrict__ graph_ptr, int ld4,
                                                              float eps, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    const int cnt = n1 - n0;
    if (cnt <= 0) return;
    const float inv = 1.0f / (float)cnt;
    // channel groups are processed in passes of G = min(ld4, 64) groups by 256 / G row slices
    const int G = ld4 < 64 ? ld4 : 64, S = 256 / G;
    const int g = threadIdx.x % G, rs = threadIdx.x / G;
    for (int c0 = 0; c0 < ld4; c0 += G) {
        const int cg = c0 + g;
        const bool live = cg < ld4 && rs < S;
        auto stats = [&](const float* x, f32x4& mean, f32x4& rstd) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (live)
                for (int r = n0 + rs; r < n1; r += S) s += reinterpret_cast<const f32x4*>(x)[(size_t)r * ld4 + cg];
            red[threadIdx.x] = s;
            __syncthreads();
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < S; ++k) t += red[k * G + g];
            mean = t * inv;
            __syncthreads();
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (live)
                for (int r = n0 + rs; r < n1; r += S) {
                    const f32x4 d = reinterpret_cast<const f32x4*>(x)[(size_t)r * ld4 + cg] - mean;
                    v += d * d;
                }
            red[threadIdx.x] = v;
            __syncthreads();
            t = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < S; ++k) t += red[k * G + g];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 4; ++m) rstd[m] = 1.0f / sqrtf(t[m] * inv + eps);
        };
        f32x4 mm, mr, gm = {0.f, 0.f, 0.f, 0.f}, gr = {0.f, 0.f, 0.f, 0.f};
        stats(main_pre, mm, mr);
        if (gate_pre) stats(gate_pre, gm, gr);
        if (live)
            for (int r = n0 + rs; r < n1; r += S) {
                const size_t o = (size_t)r * ld4 + cg;
                const f32x4 mn = (reinterpret_cast<const f32x4*>(main_pre)[o] - mm) * mr;
                f32x4 res = mn;
                if (gate_pre) {
                    const f32x4 gn = (reinterpret_cast<const f32x4*>(gate_pre)[o] - gm) * gr, hv = reinterpret_cast<const f32x4*>(h)[o];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float tau = sigmoidf_(gn[m]);
                        res[m] = (1.0f - tau) * hv[m] + tau * swishf(mn[m]);
                    }
                }
                reinterpret_cast<f32x4*>(out)[o] = res;
            }
    }
}

}  // namespace msmp

using namespace msmp;

static bool ld_ok(int width, int ld) { return width >= 1 && ld >= width && ld % 4 == 0 && ld <= 4096; }

extern "C" int msmp_wide_gather_swish_f32(const float* p, const float* q, const int32_t* tgt, const int32_t* col, int64_t n_edges, int width, int ld,
                                          float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(p && q && tgt && col && out, MSMP_ERR_ARG, "msmp_wide_gather_swish_f32: null pointer");
    MSMP_REQUIRE(n_edges >= 0 && n_edges < (1L << 31) && ld_ok(width, ld), MSMP_ERR_ARG, "msmp_wide_gather_swish_f32: bad sizes");
    if (n_edges == 0) return MSMP_OK;
    const long total = (long)n_edges * (ld / 4);
    hipLaunchKernelGGL(wide_gather_swish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, q, tgt, col, (long)n_edges,
                       ld / 4, out);
    return check_launch("wide_gather_swish_kernel");
}

extern "C" int msmp_wide_scatter_mean_f32(const float* msg, const int32_t* rowptr, int64_t n_nodes, int width, int ld, float* agg_out,
                                          msmp_stream_t stream) {
    MSMP_REQUIRE(msg && rowptr && agg_out, MSMP_ERR_ARG, "msmp_wide_scatter_mean_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && ld_ok(width, ld), MSMP_ERR_ARG, "msmp_wide_scatter_mean_f32: bad sizes");
    const long total = (long)n_nodes * (ld / 4);
    hipLaunchKernelGGL(wide_scatter_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, msg, rowptr, (long)n_nodes,
                       ld / 4, agg_out);
    return check_launch("wide_scatter_mean_kernel");
}

extern "C" int msmp_wide_lem_z_f32(const float* g, int64_t n_nodes, int width, float dt, float* z, float* dtbar_out, msmp_stream_t stream) {
    MSMP_REQUIRE(g && z && dtbar_out, MSMP_ERR_ARG, "msmp_wide_lem_z_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && width > 0 && n_nodes * (int64_t)width < (1LL << 40), MSMP_ERR_ARG, "msmp_wide_lem_z_f32: bad sizes");
    const long total = (long)n_nodes * width;
    hipLaunchKernelGGL(wide_lem_z_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, (long)n_nodes, width, dt, z, dtbar_out);
    return check_launch("wide_lem_z_kernel");
}

extern "C" int msmp_wide_lem_y_f32(const float* lin, const float* dtbar, int64_t n_floats, float* y, msmp_stream_t stream) {
    MSMP_REQUIRE(lin && dtbar && y, MSMP_ERR_ARG, "msmp_wide_lem_y_f32: null pointer");
    MSMP_REQUIRE(n_floats > 0, MSMP_ERR_ARG, "msmp_wide_lem_y_f32: bad size");
    hipLaunchKernelGGL(wide_lem_y_kernel, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lin, dtbar, (long)n_floats, y);
    return check_launch("wide_lem_y_kernel");
}

extern "C" int msmp_wide_swish_f32(const float* x, int64_t n_floats, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(x && out && n_floats > 0 && n_floats % 4 == 0, MSMP_ERR_ARG, "msmp_wide_swish_f32: bad arguments");
    hipLaunchKernelGGL(wide_swish_kernel, dim3((unsigned)((n_floats / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)(n_floats / 4), out);
    return check_launch("wide_swish_kernel");
}

extern "C" int msmp_wide_norm_blend_f32(const float* h, const float* gate_pre, const float* main_pre, const int32_t* graph_ptr, int64_t n_graphs,
                                        int width, int ld, float eps, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(main_pre && graph_ptr && out && (!gate_pre || h), MSMP_ERR_ARG, "msmp_wide_norm_blend_f32: null pointer");
    MSMP_REQUIRE(n_graphs > 0 && n_graphs < (1L << 31) && ld_ok(width, ld), MSMP_ERR_ARG, "msmp_wide_norm_blend_f32: bad sizes");
    hipLaunchKernelGGL(wide_norm_blend_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, h, gate_pre, main_pre, graph_ptr, ld / 4,
                       eps, out);
    return check_launch("wide_norm_blend_kernel");
}
