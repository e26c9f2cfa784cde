// LEM encoder for TRAINING: the T-step recurrence with its per-step activations saved, and
// its backward pass (back-propagation through time) as one kernel each, replacing `lem_cuda.forward` / `lem_cuda.backward`
// of the reference's absent extension (experiments/models_gnn.py:285-302: LEMFunction.forward saves
// all_X, all_X2, all_multi_scales, all_lin_new_z_state ...; LEMFunction.backward returns the gradients of weights,
// weights_lin_z, bias, bias_lin_z and drops the gradient of `inputs`, :296-302).
//
// Forward (per step, per node; same cell as lem_kernel.hip):
//     g = W [y ; x_t] + b;  a1 = dt s(g1);  a2 = dt s(g2);  c = tanh(g3);  z' = (1-a2) z + a2 c
//     l = Wz [z' ; x_t] + bz;  d = tanh(l);  y' = (1-a1) y + a1 d
// Backward (t = T-1 .. 0, carrying dy = dL/dy_t and dz = dL/dz_t):
//     da1 = dy (d - y);  dd = dy a1;  dy <- dy (1-a1);  dl = dd (1-d^2);  dg1 = da1 a1 (1 - a1/dt)
//     dz += Wz[:, :H]^T dl;  da2 = dz (c - z);  dc = dz a2;  dz <- dz (1-a2);  dg2 = da2 a2 (1 - a2/dt);  dg3 = dc (1-c^2)
//     dy += W[:, :H]^T (dg1, dg2, dg3)
// The kernel writes dG = (dg1 | dg2 | dg3 | dl) per node and step; the parameter gradients are then four plain GEMMs over
// the N*T rows (dW = dG[:, :3H]^T [y_prev ; x], dWz = dl^T [z' ; x]) and two column sums, done by the host layer with
// rocBLAS (K = N*T: library-GEMM shaped).
//
// Both kernels use the channel-major exact-fp32 MFMA scheme of lem_encoder_kernel (one node per lane, the carried
// tensors in accumulator layout for all T steps, weight chunks of [128][32] streamed through the double-buffered LDS
// pipeline), with the four channel tiles of a 32-node block on four waves (see "workgroup shape" below); the backward
// consumes chunks of the TRANSPOSED recurrent blocks (msmp_pack_lem_bwd_f32).  Saved tensors are
// node-major [6][N][T][128] (a2, c, a1, d, y', z'), so each is directly the row matrix of the weight-gradient GEMMs.
// Sized for training batches (tens of graphs: the forward + backward pair replaces ~2 000 PyTorch launches); at
// N = 1 600, T = 25 the six saved tensors are 123 MB.
#include "lem_layout.h"

namespace msmp {

constexpr int LEM_SAVED = 6;      // a2, c, a1, d, y, z
enum { SV_A2 = 0, SV_C = 1, SV_A1 = 2, SV_D = 3, SV_Y = 4, SV_Z = 5 };

struct LemTrainArgs {
    const float* xin;    // [N, T, 2*NS]
    long n_nodes;
    int t_len;
    float dt;
    const float* rec;    // 16 chunks (g2, g3, g1, lin)
    const float* bias;   // [512]
    const float* wx;     // input-column fragments
    float* saved;        // [6][N][T][128], or NULL (no backward will follow: inference with carried states)
    float* out;          // [N,128] = y_T
    const float* y0;     // [N,128] initial states, or NULL = zeros (LEMcuda.forward's `states`, models_gnn.py:325-332)
    const float* z0;
    float* z_out;        // [N,128] = z_T, or NULL
};

// one tile of a node row <-> accumulator registers: register 4q+m of tile T is channel 32T + 8q + 4hh + m
__device__ __forceinline__ void tile_load(const float* row_hh, int T, f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(row_hh + 32 * T + 8 * q);
#pragma unroll
        for (int m = 0; m < 4; ++m) v[4 * q + m] = p[m];
    }
}

__device__ __forceinline__ void tile_store(float* row_hh, int T, const f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 p;
#pragma unroll
        for (int m = 0; m < 4; ++m) p[m] = v[4 * q + m];
        *reinterpret_cast<f32x4*>(row_hh + 32 * T + 8 * q) = p;
    }
}

// ---- workgroup shape -------------------------------------------------------------------------------------------------
// 256 threads = ONE block of 32 nodes; wave ct owns channel tile ct (32 of the 128 channels) of every tensor, so the
// 25-step dependency chain of a wave is 16 MFMAs per weight chunk instead of 64 (the chain, not the throughput, sets the
// time at training batch sizes: 1 600 nodes = 50 workgroups).  A GEMM needs all 128 channels of its B operand: the
// operand tensor is published in LDS as xs[channel][node] (row stride 40 floats: the two half-waves of a B fragment read
// rows 4 apart = 32 banks apart, conflict-free) and every wave reads its fragments from there.
constexpr int XS = 40;

__device__ __forceinline__ void publish_tile(float* xs, int ct, int c, int hh, const f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[(32 * ct + acc_row(r, hh)) * XS + c] = v[r];
}

// acc (tile ct) += W_chunk[32 ct .., k] * X[32 kc + k][node]  for the 32 k of one staged chunk
__device__ __forceinline__ void mma_chunk_tile(const float* wl, const float* xs, int ct, int kc, int c, int hh, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wl + (32 * ct + c) * LDW + 8 * q + 4 * hh);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], xs[(32 * kc + 8 * q + 4 * hh + m) * XS + c], acc, 0, 0, 0);
    }
}

// acc (tile ct) = bias + W[:, H:H+ninp] x_t  (the input columns as NS fp32 MFMA k-steps)
template <int NS>
__device__ __forceinline__ void tile_init(const LemTrainArgs& a, int grp, int ct, int lane, int hh, const float (&x)[2 * NS], f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + H * grp + 32 * ct + 8 * q + 4 * hh);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * q + m] = bv[m];
    }
    const float* wf = a.wx + (size_t)grp * 1024 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[(ct * 4 + s) * 64], hh ? x[2 * s + 1] : x[2 * s], acc, 0, 0, 0);
}

// one GEMM group: 4 chunks (K = 128) against the tensor currently published in xs; the chunk after the last is NEXT
#define LEM_TRAIN_GROUP(SRC, ACC, BASE, NEXT_AFTER_LAST)                                                 \
    _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) {                                                   \
        const float* nxt = kc < 3 ? (SRC) + (size_t)((BASE) + kc + 1) * CHUNK_FLOATS : (NEXT_AFTER_LAST); \
        wstage_load(ws, nxt, tid);                                                                       \
        mma_chunk_tile(lds + (((BASE) + kc) & 1) * H * LDW, xs, ct, kc, c, hh, ACC);                     \
        wstage_store(ws, lds + (((BASE) + kc + 1) & 1) * H * LDW, tid);                                  \
        __syncthreads();                                                                                 \
    }

template <int NS>
__global__ __launch_bounds__(256) void lem_train_fwd_kernel(LemTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    __shared__ float xs[H * XS];
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 32 + c;
    const bool live = n < a.n_nodes;
    const long nc = live ? n : a.n_nodes - 1;
    const float* xrow = a.xin + (size_t)nc * a.t_len * (2 * NS);
    const size_t plane = (size_t)a.n_nodes * a.t_len * H;
    float* srow = a.saved + (size_t)nc * a.t_len * H + 4 * hh;

    f32x16 y, z, g, acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[r] = 0.f; z[r] = 0.f; }
    if (a.y0) tile_load(a.y0 + (size_t)nc * H + 4 * hh, ct, y);
    if (a.z0) tile_load(a.z0 + (size_t)nc * H + 4 * hh, ct, z);
    const bool save = live && a.saved != nullptr;
    publish_tile(xs, ct, c, hh, y);

    WStage ws;
    wstage_load(ws, a.rec, tid);
    wstage_store(ws, lds, tid);
    __syncthreads();

    for (int t = 0; t < a.t_len; ++t) {
        float x[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) x[f] = xrow[t * (2 * NS) + f];
        float* st = srow + (size_t)t * H;

        tile_init<NS>(a, 1, ct, lane, hh, x, g);                      // g2 -> a2            (xs = y)
        LEM_TRAIN_GROUP(a.rec, g, 0, a.rec + 4 * CHUNK_FLOATS)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = a.dt * sigmoidf_(g[r]);
        if (save) tile_store(st + SV_A2 * plane, ct, g);
        tile_init<NS>(a, 2, ct, lane, hh, x, acc);                    // g3 -> c, z'
        LEM_TRAIN_GROUP(a.rec, acc, 4, a.rec + 8 * CHUNK_FLOATS)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = tanhf_(acc[r]);
            z[r] = (1.0f - g[r]) * z[r] + g[r] * acc[r];
        }
        if (save) { tile_store(st + SV_C * plane, ct, acc); tile_store(st + SV_Z * plane, ct, z); }
        tile_init<NS>(a, 0, ct, lane, hh, x, g);                      // g1 -> a1
        LEM_TRAIN_GROUP(a.rec, g, 8, a.rec + 12 * CHUNK_FLOATS)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = a.dt * sigmoidf_(g[r]);
        if (save) tile_store(st + SV_A1 * plane, ct, g);
        publish_tile(xs, ct, c, hh, z);                               // every wave is past the barrier after its last read of y
        __syncthreads();
        tile_init<NS>(a, 3, ct, lane, hh, x, acc);                    // lin -> d, y'        (xs = z')
        LEM_TRAIN_GROUP(a.rec, acc, 12, a.rec)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = tanhf_(acc[r]);
            y[r] = (1.0f - g[r]) * y[r] + g[r] * acc[r];
        }
        if (save) { tile_store(st + SV_D * plane, ct, acc); tile_store(st + SV_Y * plane, ct, y); }
        publish_tile(xs, ct, c, hh, y);
        __syncthreads();
    }
    if (live) tile_store(a.out + (size_t)n * H + 4 * hh, ct, y);
    if (live && a.z_out) tile_store(a.z_out + (size_t)n * H + 4 * hh, ct, z);
}

struct LemBwdArgs {
    const float* gout;   // [N,128] dL/dy_T
    const float* saved;  // [6][N][T][128]
    long n_nodes;
    int t_len;
    float dt;
    const float* rec_t;  // 16 transposed chunks: g1 x4, lin x4, g2 x4, g3 x4 (consumption order)
    float* dg;           // [N][T][512]: dg1 | dg2 | dg3 | dl (the row order of weights, then weights_lin_z)
    const float* y0;     // the forward's initial states (NULL = zeros)
    const float* z0;
};

__global__ __launch_bounds__(256) void lem_bptt_kernel(LemBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    __shared__ float xs[H * XS];
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 32 + c;
    const bool live = n < a.n_nodes;
    const long nc = live ? n : a.n_nodes - 1;
    const size_t plane = (size_t)a.n_nodes * a.t_len * H;
    const float* srow = a.saved + (size_t)nc * a.t_len * H + 4 * hh;
    float* grow = a.dg + (size_t)nc * a.t_len * (4 * H) + 4 * hh;
    const float inv_dt = 1.0f / a.dt;

    f32x16 dy, dz, p, q;
    tile_load(a.gout + (size_t)nc * H + 4 * hh, ct, dy);
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = 0.f;

    WStage ws;
    wstage_load(ws, a.rec_t, tid);
    wstage_store(ws, lds, tid);
    __syncthreads();

    for (int t = a.t_len - 1; t >= 0; --t) {
        const float* st = srow + (size_t)t * H;
        float* gt = grow + (size_t)t * (4 * H);
        {   // y' = (1-a1) y + a1 d:  p = dg1, q = dl
            f32x16 a1, d, yp;
            tile_load(st + SV_A1 * plane, ct, a1);
            tile_load(st + SV_D * plane, ct, d);
            if (t > 0) tile_load(st - H + SV_Y * plane, ct, yp);
            else if (a.y0) tile_load(a.y0 + (size_t)nc * H + 4 * hh, ct, yp);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) yp[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = dy[r];
                q[r] = g * a1[r] * (1.0f - d[r] * d[r]);
                p[r] = g * (d[r] - yp[r]) * a1[r] * (1.0f - a1[r] * inv_dt);
                dy[r] = g * (1.0f - a1[r]);
            }
            if (live) { tile_store(gt, ct, p); tile_store(gt + 3 * H, ct, q); }
        }
        publish_tile(xs, ct, c, hh, p);               // xs is free: the previous group ended with a barrier
        __syncthreads();
        LEM_TRAIN_GROUP(a.rec_t, dy, 0, a.rec_t + 4 * CHUNK_FLOATS)
        publish_tile(xs, ct, c, hh, q);
        __syncthreads();
        LEM_TRAIN_GROUP(a.rec_t, dz, 4, a.rec_t + 8 * CHUNK_FLOATS)
        {   // z' = (1-a2) z + a2 c:  p = dg2, q = dg3
            f32x16 a2, cc, zp;
            tile_load(st + SV_A2 * plane, ct, a2);
            tile_load(st + SV_C * plane, ct, cc);
            if (t > 0) tile_load(st - H + SV_Z * plane, ct, zp);
            else if (a.z0) tile_load(a.z0 + (size_t)nc * H + 4 * hh, ct, zp);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = dz[r];
                q[r] = g * a2[r] * (1.0f - cc[r] * cc[r]);
                p[r] = g * (cc[r] - zp[r]) * a2[r] * (1.0f - a2[r] * inv_dt);
                dz[r] = g * (1.0f - a2[r]);
            }
            if (live) { tile_store(gt + H, ct, p); tile_store(gt + 2 * H, ct, q); }
        }
        publish_tile(xs, ct, c, hh, p);
        __syncthreads();
        LEM_TRAIN_GROUP(a.rec_t, dy, 8, a.rec_t + 12 * CHUNK_FLOATS)
        publish_tile(xs, ct, c, hh, q);
        __syncthreads();
        LEM_TRAIN_GROUP(a.rec_t, dy, 12, a.rec_t)
    }
}
#undef LEM_TRAIN_GROUP

// rec_t chunk ch = 4*grp + kc (grp: 0 g1, 1 lin, 2 g2, 3 g3):  [row k_out][kk] = M[32 kc + kk][k_out],
// M = the state block (columns 0..H-1) of weights rows 0.. / weights_lin_z / weights rows H.. / weights rows 2H..
__global__ void pack_lem_bwd_kernel(const float* w, const float* wz, int ninp, float* out) {
    const int kin = H + ninp;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < 16 * CHUNK_FLOATS; p += gridDim.x * blockDim.x) {
        const int ch = p / CHUNK_FLOATS, row = (p % CHUNK_FLOATS) / KC, kk = p % KC;
        const int grp = ch >> 2, j = (ch & 3) * KC + kk;
        float v;
        if (grp == 0) v = w[(size_t)j * kin + row];
        else if (grp == 1) v = wz[(size_t)j * kin + row];
        else if (grp == 2) v = w[(size_t)(H + j) * kin + row];
        else v = w[(size_t)(2 * H + j) * kin + row];
        out[p] = v;
    }
}

}  // namespace msmp

using namespace msmp;

extern "C" int64_t msmp_packed_lem_bwd_floats(void) { return 16 * CHUNK_FLOATS; }

extern "C" int msmp_pack_lem_bwd_f32(const float* weights, const float* weights_lin_z, int ninp, float* packed_out,
                                     msmp_stream_t stream) {
    MSMP_REQUIRE(weights && weights_lin_z && packed_out, MSMP_ERR_ARG, "msmp_pack_lem_bwd_f32: null pointer");
    MSMP_REQUIRE(ninp >= 1 && ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_pack_lem_bwd_f32: ninp=%d not in 1..%d", ninp, LEM_MAX_INP);
    hipLaunchKernelGGL(pack_lem_bwd_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, weights, weights_lin_z, ninp, packed_out);
    return check_launch("pack_lem_bwd_kernel");
}

extern "C" int64_t msmp_lem_saved_floats(int64_t n_nodes, int t_len) {
    return n_nodes > 0 && t_len > 0 ? (int64_t)LEM_SAVED * n_nodes * t_len * H : -1;
}

extern "C" int msmp_lem_train_fwd_f32(const float* xin, int64_t n_nodes, int t_len, int ninp, float dt, const float* packed,
                                      const float* y0, const float* z0, float* saved, float* y_out, float* z_out,
                                      msmp_stream_t stream) {
    MSMP_REQUIRE(xin && packed && y_out, MSMP_ERR_ARG, "msmp_lem_train_fwd_f32: null pointer");
    MSMP_REQUIRE((y0 != nullptr) == (z0 != nullptr), MSMP_ERR_ARG, "msmp_lem_train_fwd_f32: give both initial states or none");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && t_len >= 1, MSMP_ERR_ARG, "msmp_lem_train_fwd_f32: bad sizes");
    MSMP_REQUIRE(ninp >= 1 && ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_lem_train_fwd_f32: ninp=%d not in 1..%d", ninp, LEM_MAX_INP);
    const LemLayout L = lem_layout();
    LemTrainArgs a{xin, (long)n_nodes, t_len, dt, packed + L.rec, packed + L.bias, packed + L.wx, saved, y_out, y0, z0, z_out};
    const unsigned grid = (unsigned)((n_nodes + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    switch ((ninp + 1) / 2) {
        case 1: hipLaunchKernelGGL(lem_train_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(lem_train_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(lem_train_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(lem_train_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, a); break;
    }
    return check_launch("lem_train_fwd_kernel");
}

extern "C" int msmp_lem_train_bwd_f32(const float* grad_y, const float* saved, const float* y0, const float* z0, int64_t n_nodes,
                                      int t_len, float dt, const float* packed_bwd, float* dg_out, msmp_stream_t stream) {
    MSMP_REQUIRE(grad_y && saved && packed_bwd && dg_out, MSMP_ERR_ARG, "msmp_lem_train_bwd_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && t_len >= 1 && dt != 0.f, MSMP_ERR_ARG, "msmp_lem_train_bwd_f32: bad sizes");
    MSMP_REQUIRE((y0 != nullptr) == (z0 != nullptr), MSMP_ERR_ARG, "msmp_lem_train_bwd_f32: give both initial states or none");
    LemBwdArgs a{grad_y, saved, (long)n_nodes, t_len, dt, packed_bwd, dg_out, y0, z0};
    const unsigned grid = (unsigned)((n_nodes + 31) / 32);
    hipLaunchKernelGGL(lem_bptt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("lem_bptt_kernel");
}
