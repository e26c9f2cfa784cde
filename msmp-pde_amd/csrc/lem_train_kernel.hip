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
// Both kernels keep the carried tensors in accumulator layout for all T steps (one node per lane, channel tile per wave) and
// multiply on the bf16 matrix pipe with fp32-exact products (three-way bf16 split, "workgroup shape" below); the backward
// consumes chunks of the TRANSPOSED recurrent blocks (msmp_pack_lem_bwd_f32).  Saved tensors are
// node-major [6][N][T][128] (a2, c, a1, d, y_{t-1}, z'), so each is directly the row matrix of the weight-gradient GEMMs.
#include "lem_layout.h"

namespace msmp {

constexpr int LEM_SAVED = 6;      // a2, c, a1, d, y, z
enum { SV_A2 = 0, SV_C = 1, SV_A1 = 2, SV_D = 3, SV_Y = 4, SV_Z = 5 };

struct LemTrainArgs {
    const float* xin;    // [N, T, 2*NS]
    long n_nodes;
    int t_len;
    float dt;
    const float* rec_b3; // 16 chunks (g2, g3, g1, lin) as bf16x3 fragments, acc order
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

// ---- workgroup shape (round 2) -----------------------------------------------------------------------------------------
// 256 threads = LEM_NX 32-node tiles; wave ct owns channel tile ct (32 of the 128 channels) of every tensor of every tile.  A GEMM needs all 128 channels of its B operand: the operand tensor is PUBLISHED in LDS as bf16 hi / mid / lo MFMA
// fragments ("acc order": K step s of k-tile kt of a lane is registers 8 s .. 8 s + 7 of wave kt's accumulator tile, so a
// publish is a register split + three 16-byte stores, no transposition), and every wave multiplies its own rows of the weight
// chunk -- bf16x3 fragments straight from the L2-resident packed blob, nothing staged: a wave needs only ITS 32 rows -- with all
// published fragments: six bf16 MFMAs per K = 16 step give fp32-exact products (mfma_tiles.h).  Against the first edition (one
// 32-node tile per workgroup, v_mfma_f32_32x32x2_f32, every 16-KB weight chunk staged through LDS with a barrier: 19 barriers and
// 256 KB of staging per time step) a step is 4 (forward) / 8 (backward) barriers and the matrix time per node drops 2.7x.
constexpr int XF_TILE_U4 = 4 * 2 * 3 * 64;         // published fragments of one node tile: [kt][s][plane][lane], 24 KB
// node tiles per workgroup: two (64 nodes, 228 / 247 registers, two workgroups per CU) halve the weight-fragment traffic but measured
// the same as one (144 / 168 registers, three workgroups per CU): 30.3 vs 30.0 ms per batch-512 training iteration.
constexpr int LEM_NX = 1;

__device__ __forceinline__ void publish_b3(u32x4* xf_tile, int ct, int lane, const f32x16& v) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[8 * s + j];
        const Bf3 f = split_bf16x3(t);
        u32x4* dst = xf_tile + (size_t)((ct * 2 + s) * 3) * 64 + lane;
        dst[0] = __builtin_bit_cast(u32x4, f.hi);
        dst[64] = __builtin_bit_cast(u32x4, f.mid);
        dst[128] = __builtin_bit_cast(u32x4, f.lo);
    }
}

struct WFrag {
    u32x4 f[2][3];       // [s][plane] of this wave's 32 rows of one chunk
};
// Weight fragments come through BUFFER loads: (resource descriptor in scalar registers) + (scalar offset of the fragment) + (one
// 32-bit vector offset, 16 * lane).  With global loads the compiler forms a per-lane 64-bit address for each of the 96 fragments
// of a time step, hoists them out of the t loop as invariants and spills them (49 scratch stores at the head of the first edition).
__device__ __forceinline__ void wfrag_load(WFrag& w, __amdgpu_buffer_rsrc_t chunks, int ch, int ct, unsigned loff) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            w.f[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(chunks, loff, (ch * LEM_B3_CHUNK_U4 + ((s * 4 + ct) * 3 + pl) * 64) * 16, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wfrag_rsrc(const float* b3_chunks) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b3_chunks), 0, 16 * LEM_B3_CHUNK_FLOATS * 4, 0x00027000);
}
// acc[X] (tile ct) += W_chunk[32 ct .., k-tile kc] * published[X][k-tile kc]
__device__ __forceinline__ void mma_chunk_b3(const WFrag& w, const u32x4* xf, int kc, int lane, f32x16 (&acc)[LEM_NX]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, w.f[s][0]), wm = __builtin_bit_cast(bf16x8, w.f[s][1]), wl = __builtin_bit_cast(bf16x8, w.f[s][2]);
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
            const u32x4* b = xf + (size_t)X * XF_TILE_U4 + (size_t)((kc * 2 + s) * 3) * 64 + lane;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[64]), bl = __builtin_bit_cast(bf16x8, b[128]);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc[X], 0, 0, 0);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc[X], 0, 0, 0);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, bm, acc[X], 0, 0, 0);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, bh, acc[X], 0, 0, 0);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bm, acc[X], 0, 0, 0);
            acc[X] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc[X], 0, 0, 0);
        }
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

// one GEMM group: the four k-tiles of chunk group BASE against the tensor currently published; the fragments of the chunk after
// the group's last are requested too (NEXT), so a group starts with its first chunk in registers
#define LEM_B3_GROUP(CHUNKS, ACC, BASE, NEXT)                                            \
    _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) {                                   \
        WFrag& cur_ = (kc & 1) ? wB : wA;                                                \
        WFrag& nxt_ = (kc & 1) ? wA : wB;                                                \
        wfrag_load(nxt_, CHUNKS, kc < 3 ? (BASE) + kc + 1 : (NEXT), cts, loff);           \
        mma_chunk_b3(cur_, xf, kc, lane, ACC);                                           \
        __builtin_amdgcn_sched_barrier(0);   /* one chunk of weight fragments ahead, not sixteen */ \
    }

template <int NS>
__global__ __launch_bounds__(256, 2) void lem_train_fwd_kernel(LemTrainArgs a) {
    __shared__ u32x4 xf[LEM_NX * XF_TILE_U4];                   // 48 KB
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const size_t plane = (size_t)a.n_nodes * a.t_len * H;
    const __amdgpu_buffer_rsrc_t chunks = wfrag_rsrc(a.rec_b3);
    bool live[LEM_NX];
    long nc[LEM_NX];
    const float* xrow[LEM_NX];
    float* srow[LEM_NX];
#pragma unroll
    for (int X = 0; X < LEM_NX; ++X) {
        const long n = (long)blockIdx.x * (32 * LEM_NX) + 32 * X + c;
        live[X] = n < a.n_nodes;
        nc[X] = live[X] ? n : a.n_nodes - 1;
        xrow[X] = a.xin + (size_t)nc[X] * a.t_len * (2 * NS);
        srow[X] = a.saved + (size_t)nc[X] * a.t_len * H + 4 * hh;
    }
    f32x16 y[LEM_NX], z[LEM_NX], g[LEM_NX], acc[LEM_NX];
#pragma unroll
    for (int X = 0; X < LEM_NX; ++X) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { y[X][r] = 0.f; z[X][r] = 0.f; }
        if (a.y0) tile_load(a.y0 + (size_t)nc[X] * H + 4 * hh, ct, y[X]);
        if (a.z0) tile_load(a.z0 + (size_t)nc[X] * H + 4 * hh, ct, z[X]);
        publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, y[X]);
    }
    WFrag wA, wB;
    const int cts = __builtin_amdgcn_readfirstlane(ct);
    unsigned loff = 16u * lane;
    asm volatile("" : "+v"(loff));
    wfrag_load(wA, chunks, 0, cts, loff);
    __syncthreads();

    for (int t = 0; t < a.t_len; ++t) {
        float x[LEM_NX][2 * NS];
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X)
#pragma unroll
            for (int f = 0; f < 2 * NS; ++f) x[X][f] = xrow[X][t * (2 * NS) + f];

#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
            // y_{t-1}, the state ENTERING the step: what the backward and the weight gradients ([y_{t-1} ; x_t] rows) read
            if (live[X] && a.saved) tile_store(srow[X] + (size_t)t * H + SV_Y * plane, ct, y[X]);
            tile_init<NS>(a, 1, ct, lane, hh, x[X], g[X]);                                 // g2 -> a2            (published: y)
        }
        LEM_B3_GROUP(chunks, g, 0, 4)
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[X][r] = a.dt * sigmoidf_(g[X][r]);
            if (live[X] && a.saved) tile_store(srow[X] + (size_t)t * H + SV_A2 * plane, ct, g[X]);
            tile_init<NS>(a, 2, ct, lane, hh, x[X], acc[X]);                               // g3 -> c, z'
        }
        LEM_B3_GROUP(chunks, acc, 4, 8)
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[X][r] = tanhf_(acc[X][r]);
                z[X][r] = (1.0f - g[X][r]) * z[X][r] + g[X][r] * acc[X][r];
            }
            if (live[X] && a.saved) {
                tile_store(srow[X] + (size_t)t * H + SV_C * plane, ct, acc[X]);
                tile_store(srow[X] + (size_t)t * H + SV_Z * plane, ct, z[X]);
            }
            tile_init<NS>(a, 0, ct, lane, hh, x[X], g[X]);                                 // g1 -> a1
        }
        LEM_B3_GROUP(chunks, g, 8, 12)
        __syncthreads();                                   // every wave has read the last fragment of y
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[X][r] = a.dt * sigmoidf_(g[X][r]);
            if (live[X] && a.saved) tile_store(srow[X] + (size_t)t * H + SV_A1 * plane, ct, g[X]);
            publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, z[X]);
            tile_init<NS>(a, 3, ct, lane, hh, x[X], acc[X]);                               // lin -> d, y'        (published: z')
        }
        __syncthreads();
        LEM_B3_GROUP(chunks, acc, 12, 0)
        __syncthreads();                                   // every wave has read the last fragment of z'
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[X][r] = tanhf_(acc[X][r]);
                y[X][r] = (1.0f - g[X][r]) * y[X][r] + g[X][r] * acc[X][r];
            }
            if (live[X] && a.saved) tile_store(srow[X] + (size_t)t * H + SV_D * plane, ct, acc[X]);
            publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, y[X]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int X = 0; X < LEM_NX; ++X) {
        const long n = (long)blockIdx.x * (32 * LEM_NX) + 32 * X + c;
        if (live[X]) tile_store(a.out + (size_t)n * H + 4 * hh, ct, y[X]);
        if (live[X] && a.z_out) tile_store(a.z_out + (size_t)n * H + 4 * hh, ct, z[X]);
    }
}

struct LemBwdArgs {
    const float* gout;   // [N,128] dL/dy_T
    const float* saved;  // [6][N][T][128]
    long n_nodes;
    int t_len;
    float dt;
    const float* rec_t;  // 16 transposed chunks as bf16x3 fragments: g1 x4, lin x4, g2 x4, g3 x4 (consumption order)
    float* dg;           // [N][T][512]: dg1 | dg2 | dg3 | dl (the row order of weights, then weights_lin_z)
    const float* y0;     // the forward's initial states (NULL = zeros)
    const float* z0;
};

__global__ __launch_bounds__(256, 2) void lem_bptt_kernel(LemBwdArgs a) {
    __shared__ u32x4 xf[LEM_NX * XF_TILE_U4];
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const size_t plane = (size_t)a.n_nodes * a.t_len * H;
    const __amdgpu_buffer_rsrc_t chunks = wfrag_rsrc(a.rec_t);
    const float inv_dt = 1.0f / a.dt;
    bool live[LEM_NX];
    long nc[LEM_NX];
    const float* srow[LEM_NX];
    float* grow[LEM_NX];
    f32x16 dy[LEM_NX], dz[LEM_NX], p[LEM_NX], q[LEM_NX];
#pragma unroll
    for (int X = 0; X < LEM_NX; ++X) {
        const long n = (long)blockIdx.x * (32 * LEM_NX) + 32 * X + c;
        live[X] = n < a.n_nodes;
        nc[X] = live[X] ? n : a.n_nodes - 1;
        srow[X] = a.saved + (size_t)nc[X] * a.t_len * H + 4 * hh;
        grow[X] = a.dg + (size_t)nc[X] * a.t_len * (4 * H) + 4 * hh;
        tile_load(a.gout + (size_t)nc[X] * H + 4 * hh, ct, dy[X]);
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[X][r] = 0.f;
    }
    WFrag wA, wB;
    const int cts = __builtin_amdgcn_readfirstlane(ct);
    unsigned loff = 16u * lane;
    asm volatile("" : "+v"(loff));
    wfrag_load(wA, chunks, 0, cts, loff);

    for (int t = a.t_len - 1; t >= 0; --t) {
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {   // y' = (1-a1) y + a1 d:  p = dg1, q = dl
            const float* st = srow[X] + (size_t)t * H;
            f32x16 a1, d, yp;
            tile_load(st + SV_A1 * plane, ct, a1);
            tile_load(st + SV_D * plane, ct, d);
            tile_load(st + SV_Y * plane, ct, yp);              // y_{t-1} (the forward saved the state entering the step)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gr = dy[X][r];
                q[X][r] = gr * a1[r] * (1.0f - d[r] * d[r]);
                p[X][r] = gr * (d[r] - yp[r]) * a1[r] * (1.0f - a1[r] * inv_dt);
                dy[X][r] = gr * (1.0f - a1[r]);
            }
            if (live[X]) {
                float* gt = grow[X] + (size_t)t * (4 * H);
                tile_store(gt, ct, p[X]);
                tile_store(gt + 3 * H, ct, q[X]);
            }
            publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, p[X]);       // the fragments are free: the previous group ended with a barrier
        }
        __syncthreads();
        LEM_B3_GROUP(chunks, dy, 0, 4)
        __syncthreads();
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, q[X]);
        __syncthreads();
        LEM_B3_GROUP(chunks, dz, 4, 8)
        __syncthreads();
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) {   // z' = (1-a2) z + a2 c:  p = dg2, q = dg3
            const float* st = srow[X] + (size_t)t * H;
            f32x16 a2, cc, zp;
            tile_load(st + SV_A2 * plane, ct, a2);
            tile_load(st + SV_C * plane, ct, cc);
            if (t > 0) tile_load(st - H + SV_Z * plane, ct, zp);
            else if (a.z0) tile_load(a.z0 + (size_t)nc[X] * H + 4 * hh, ct, zp);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gr = dz[X][r];
                q[X][r] = gr * a2[r] * (1.0f - cc[r] * cc[r]);
                p[X][r] = gr * (cc[r] - zp[r]) * a2[r] * (1.0f - a2[r] * inv_dt);
                dz[X][r] = gr * (1.0f - a2[r]);
            }
            if (live[X]) {
                float* gt = grow[X] + (size_t)t * (4 * H);
                tile_store(gt + H, ct, p[X]);
                tile_store(gt + 2 * H, ct, q[X]);
            }
            publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, p[X]);
        }
        __syncthreads();
        LEM_B3_GROUP(chunks, dy, 8, 12)
        __syncthreads();
#pragma unroll
        for (int X = 0; X < LEM_NX; ++X) publish_b3(xf + (size_t)X * XF_TILE_U4, ct, lane, q[X]);
        __syncthreads();
        LEM_B3_GROUP(chunks, dy, 12, 0)
        __syncthreads();
    }
}
#undef LEM_B3_GROUP

// rec_t chunk ch = 4*grp + kc (grp: 0 g1, 1 lin, 2 g2, 3 g3):  [row k_out][kk] = M[32 kc + kk][k_out],
// M = the state block (columns 0..H-1) of weights rows 0.. / weights_lin_z / weights rows H.. / weights rows 2H..; written as
// bf16x3 A fragments, acc order: thread = (chunk, s, T, lane)
__global__ __launch_bounds__(256) void pack_lem_bwd_kernel(const float* w, const float* wz, int ninp, float* out) {
    const int kin = H + ninp;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= 16 * 512) return;
    const int lane = id & 63, T = (id >> 6) & 3, s = (id >> 8) & 1, ch = id >> 9;
    const int m = lane & 31, hh = lane >> 5, row = 32 * T + m, grp = ch >> 2;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int j = (ch & 3) * KC + split_k_acc(s, hh, jj);
        if (grp == 0) v[jj] = w[(size_t)j * kin + row];
        else if (grp == 1) v[jj] = wz[(size_t)j * kin + row];
        else if (grp == 2) v[jj] = w[(size_t)(H + j) * kin + row];
        else v[jj] = w[(size_t)(2 * H + j) * kin + row];
    }
    const Bf3 f = split_bf16x3(v);
    u32x4* dst = reinterpret_cast<u32x4*>(out) + (size_t)ch * LEM_B3_CHUNK_U4 + (size_t)((s * 4 + T) * 3) * 64 + lane;
    dst[0] = __builtin_bit_cast(u32x4, f.hi);
    dst[64] = __builtin_bit_cast(u32x4, f.mid);
    dst[128] = __builtin_bit_cast(u32x4, f.lo);
}

}  // namespace msmp

using namespace msmp;

extern "C" int64_t msmp_packed_lem_bwd_floats(void) { return 16 * LEM_B3_CHUNK_FLOATS; }

extern "C" int msmp_pack_lem_bwd_f32(const float* weights, const float* weights_lin_z, int ninp, float* packed_out,
                                     msmp_stream_t stream) {
    MSMP_REQUIRE(weights && weights_lin_z && packed_out, MSMP_ERR_ARG, "msmp_pack_lem_bwd_f32: null pointer");
    MSMP_REQUIRE(ninp >= 1 && ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_pack_lem_bwd_f32: ninp=%d not in 1..%d", ninp, LEM_MAX_INP);
    hipLaunchKernelGGL(pack_lem_bwd_kernel, dim3(16 * 512 / 256), dim3(256), 0, (hipStream_t)stream, weights, weights_lin_z, ninp, packed_out);
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
    LemTrainArgs a{xin, (long)n_nodes, t_len, dt, packed + L.rec_b3, packed + L.bias, packed + L.wx, saved, y_out, y0, z0, z_out};
    const unsigned grid = (unsigned)((n_nodes + 32 * LEM_NX - 1) / (32 * LEM_NX));
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
    const unsigned grid = (unsigned)((n_nodes + 32 * LEM_NX - 1) / (32 * LEM_NX));
    hipLaunchKernelGGL(lem_bptt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("lem_bptt_kernel");
}
