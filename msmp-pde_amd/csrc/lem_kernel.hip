// LEM ("Long Expressive Memory") node encoder as ONE kernel: the T-step recurrence plus the two-layer
// lemoutput_mlp, replacing the reference's absent `lem_cuda` extension and the PyTorch restatement of it
// (experiments/models_gnn.py:285-342 call site; input assembly :1357-1360; lemoutput_mlp :1287-1291, 1363).
//
//   per step t, per node:  g = W [y ; x_t] + b,   (g1, g2, g3) = split(g)           W [3H, H+ninp]
//                          dt_bar = dt s(g1);  dt_ = dt s(g2);  z <- (1-dt_) z + dt_ tanh(g3)
//                          y <- (1-dt_bar) y + dt_bar tanh(Wz [z ; x_t] + bz)        Wz [H, H+ninp]
//   after T steps:         h = Swish(Wb Swish(Wa y + ba) + bb)
//
// Same channel-major fp32-MFMA scheme as the MLP kernels (mfma_tiles.h): one node per lane, the states
// y and z live in MFMA accumulator layout ([128 channels][32 nodes] = 4 tiles x 16 registers) for all T
// steps and are fed back as the B operand of the next GEMM straight from registers; nothing but the
// per-step inputs is read from HBM and only h is written.  The recurrent weights (16 chunks of
// [128][32] per step: g2, g3, g1, lin, in the order they are consumed) stream through the double-buffered
// LDS pipeline from L2; the ninp <= 8 input columns are folded into the accumulator initialisation
// (acc = b + W[:, H:] x_t) as NS = ceil(ninp/2) extra MFMA k-steps per tile (the kernel is specialised on NS;
// the step inputs are read from a row padded to 2*NS floats, so the time loop is branch-free).  Algorithmic work: T * 2*(H+ninp)*4H + 2*2*H*H FLOP per node
// = 3.44 MFLOP per node at T = 25, ninp = 4 (705 GFLOP for 2048 E2 graphs): MFMA-bound.
#include "lem_layout.h"
#include <type_traits>

namespace msmp {

struct LemPackArgs {
    const float *w, *wz, *b, *bz, *wa, *ba, *wb, *bb;
    int ninp;
    float* out;
};

__global__ void pack_lem_kernel(LemPackArgs a) {
    const LemLayout L = lem_layout();
    const int kin = H + a.ninp;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < L.scales; p += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (p < L.mlp) {
            const int ch = (int)(p / CHUNK_FLOATS), row = (int)(p % CHUNK_FLOATS) / KC, kk = (int)(p % KC);
            const int grp = ch >> 2, k = (ch & 3) * KC + kk;          // consumption order: g2, g3, g1, lin
            if (grp == 0) v = a.w[(size_t)(H + row) * kin + k];
            else if (grp == 1) v = a.w[(size_t)(2 * H + row) * kin + k];
            else if (grp == 2) v = a.w[(size_t)row * kin + k];
            else v = a.wz[(size_t)row * kin + k];
        } else if (p < L.bias) {
            const int64_t o = p - L.mlp;
            const int ch = (int)(o / CHUNK_FLOATS), row = (int)(o % CHUNK_FLOATS) / KC, kk = (int)(o % KC);
            const float* m = ch < 4 ? a.wa : a.wb;
            v = m ? m[(size_t)row * H + (ch & 3) * KC + kk] : 0.f;
        } else if (p < L.wx) {
            const int r = (int)(p - L.bias);
            v = r < 3 * H ? a.b[r] : a.bz[r - 3 * H];
        } else if (p < L.mlpb) {
            const int64_t o = p - L.wx;
            const int ln = (int)(o & 63), sidx = (int)(o >> 6) & 3, T = (int)(o >> 8) & 3, grp = (int)(o >> 10);
            const int r = 128 * grp + 32 * T + (ln & 31), f = 2 * sidx + (ln >> 5);
            if (f < a.ninp) v = r < 3 * H ? a.w[(size_t)r * kin + H + f] : a.wz[(size_t)(r - 3 * H) * kin + H + f];
        } else {
            const int r = (int)(p - L.mlpb);
            const float* m = r < H ? a.ba : a.bb;
            v = m ? m[r & (H - 1)] : 0.f;
        }
        a.out[p] = v;
    }
}

// scales[i] = 2^s with max|M_i| 2^s in [16, 32) for M = (W, Wz, Wa, Wb); scales[4+i] = 2^-s.  grid = 4.
__global__ __launch_bounds__(256) void pack_lem_scale_kernel(LemPackArgs a) {
    __shared__ float red[256];
    const LemLayout L = lem_layout();
    const int kin = H + a.ninp;
    const float* w = blockIdx.x == 0 ? a.w : blockIdx.x == 1 ? a.wz : blockIdx.x == 2 ? a.wa : a.wb;
    const int n = blockIdx.x == 0 ? 3 * H * kin : blockIdx.x == 1 ? H * kin : H * H;
    float m = 0.f;
    if (w) for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);
        const int sft = mx > 0.f ? 5 - e : 0;
        a.out[L.scales + blockIdx.x] = ldexpf(1.0f, sft);
        a.out[L.scales + 4 + blockIdx.x] = ldexpf(1.0f, -sft);
    }
}

// split copies: rec_s / mlp_s from the fp32 chunks already in the blob (value at [row][k] of chunk ch -> acc order),
// and the scaled bias / input fragments.
__global__ void pack_lem_split_kernel(LemPackArgs a) {
    const LemLayout L = lem_layout();
    const float* sc = a.out + L.scales;
    _Float16* out = reinterpret_cast<_Float16*>(a.out + L.rec_s);
    const int64_t n_half = (int64_t)24 * 8192;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = tid0; p < n_half; p += stride) {
        const int ch = (int)(p >> 13), idx = (int)(p & 8191);
        const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1, T = (idx >> 10) & 3, s = (idx >> 12) & 1;
        const int row = 32 * T + (lane & 31), h = lane >> 5;
        const float* src = ch < 16 ? a.out + L.rec + (size_t)ch * CHUNK_FLOATS : a.out + L.mlp + (size_t)(ch - 16) * CHUNK_FLOATS;
        // matrix of the chunk: rec groups g2, g3, g1 -> W (scale 0), lin -> Wz (1); mlp Wa (2), Wb (3)
        const int mi = ch < 12 ? 0 : ch < 16 ? 1 : ch < 20 ? 2 : 3;
        const float w = src[row * KC + split_k_acc(s, h, j)] * sc[mi];
        const _Float16 hi = (_Float16)w;
        out[p] = plane == 0 ? hi : (_Float16)(w - (float)hi);
    }
    for (int64_t p = tid0; p < 4 * H; p += stride)           // bias rows: g1, g2, g3 (W), bz (Wz)
        a.out[L.bias_s + p] = a.out[L.bias + p] * sc[p < 3 * H ? 0 : 1];
    for (int64_t p = tid0; p < 4 * H * LEM_MAX_INP; p += stride)   // wxf groups g1, g2, g3 (W), lin (Wz)
        a.out[L.wx_s + p] = a.out[L.wx + p] * sc[(p >> 10) < 3 ? 0 : 1];
    for (int64_t p = tid0; p < 2 * H; p += stride)
        a.out[L.mlpb_s + p] = a.out[L.mlpb + p] * sc[p < H ? 2 : 3];
    _Float16* wh = reinterpret_cast<_Float16*>(a.out + L.wx_h);
    const int kin = H + a.ninp;
    for (int64_t p = tid0; p < 2 * LEM_WXH_FLOATS; p += stride) {
        const int j = (int)(p & 7), lane = (int)(p >> 3) & 63, m = (int)(p >> 9) & 1, T = (int)(p >> 10) & 3, grp = (int)(p >> 12);
        const int slot = 16 * m + 8 * (lane >> 5) + j, f = lem_slot_feature(slot, a.ninp), row = 32 * T + (lane & 31);
        _Float16 v = (_Float16)0.f;
        if (slot == 3 * a.ninp || slot == 3 * a.ninp + 1) {       // bias slots (paired with 1.0): g2, g3, g1 of `b`, then bz
            const float bv = (grp == 0 ? a.b[H + row] : grp == 1 ? a.b[2 * H + row] : grp == 2 ? a.b[row] : a.bz[row]) * sc[grp < 3 ? 0 : 1];
            const _Float16 hi = (_Float16)bv;
            v = slot == 3 * a.ninp ? hi : (_Float16)(bv - (float)hi);
        }
        if (f >= 0) {
            // groups in consumption order g2, g3, g1 (rows H.., 2H.., 0.. of W), lin (Wz)
            const float w = (grp == 0 ? a.w[(size_t)(H + row) * kin + H + f] : grp == 1 ? a.w[(size_t)(2 * H + row) * kin + H + f]
                             : grp == 2 ? a.w[(size_t)row * kin + H + f] : a.wz[(size_t)row * kin + H + f]) * sc[grp < 3 ? 0 : 1];
            const _Float16 hi = (_Float16)w;
            v = lem_slot_part(slot, a.ninp) < 2 ? hi : (_Float16)(w - (float)hi);
        }
        wh[p] = v;
    }
}

struct LemArgs {
    const float* xin;    // [N, T, 2*NS] (features past ninp are zero)
    long n_nodes;
    int t_len, with_mlp;
    float dt;
    const float* rec;    // 16 chunks
    const float* mlp;    // 8 chunks
    const float* bias;   // [512]
    const float* wx;     // wxf fragments [4][4][4][64]
    const float* mlpb;   // [256]
    float* out;          // [N, 128]
};

// acc = bias[128*grp ..] + W[128*grp .., H:H+ninp] x   (the input columns: NS extra MFMA k-steps per tile)
template <int NS>
__device__ __forceinline__ void lem_acc_init(const LemArgs& a, int grp, int lane, int hh, const float (&x)[2 * NS],
                                             f32x16 (&acc)[4][1]) {
    acc_init_bias<1>(a.bias + H * grp, hh, acc);
    const float* wf = a.wx + (size_t)grp * 1024 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float b = hh ? x[2 * s + 1] : x[2 * s];
#pragma unroll
        for (int T = 0; T < 4; ++T)
            acc[T][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[(T * 4 + s) * 64], b, acc[T][0], 0, 0, 0);
    }
}

template <int NS>
__global__ __launch_bounds__(256) void lem_encoder_kernel(LemArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * H * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const float* xrow = a.xin + (size_t)nc * a.t_len * (2 * NS);

    f32x16 y[4][1], z[4][1], g[4][1], acc[4][1];
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) { y[T][0][r] = 0.f; z[T][0][r] = 0.f; }

    // chunk stream: step t consumes rec chunks 0..15 (buffer parity = chunk parity), then the mlp chunks
    WStage ws;
    wstage_load(ws, a.rec, tid);
    wstage_store(ws, lds, tid);
    __syncthreads();

// one GEMM group: 4 chunks (K = 128) with B taken from the state X; NEXT = pointer of the chunk after each
#define LEM_GROUP(X, ACC, BASE, NEXT_AFTER_LAST)                                                       \
    _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) {                                                 \
        const float* nxt = kc < 3 ? a.rec + (size_t)((BASE) + kc + 1) * CHUNK_FLOATS : (NEXT_AFTER_LAST); \
        wstage_load(ws, nxt, tid);                                                                     \
        mma_chunk_from_acc<1>(lds + (((BASE) + kc) & 1) * H * LDW, c, hh, X[kc], ACC);                 \
        wstage_store(ws, lds + (((BASE) + kc + 1) & 1) * H * LDW, tid);                                \
        __syncthreads();                                                                               \
    }

    for (int t = 0; t < a.t_len; ++t) {
        float x[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) x[f] = xrow[t * (2 * NS) + f];
        // after the last step the stream continues with the mlp chunks (harmless prefetch if there is no mlp)
        const float* after = t + 1 == a.t_len ? a.mlp : a.rec;

        // g2 -> dt_ = dt * sigmoid
        lem_acc_init<NS>(a, 1, lane, hh, x, g);
        LEM_GROUP(y, g, 0, a.rec + 4 * CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[T][0][r] = a.dt * sigmoidf_(g[T][0][r]);
        // g3 -> z <- (1 - dt_) z + dt_ tanh(g3)
        lem_acc_init<NS>(a, 2, lane, hh, x, acc);
        LEM_GROUP(y, acc, 4, a.rec + 8 * CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                z[T][0][r] = (1.0f - g[T][0][r]) * z[T][0][r] + g[T][0][r] * tanhf_(acc[T][0][r]);
        // g1 -> dt_bar = dt * sigmoid
        lem_acc_init<NS>(a, 0, lane, hh, x, g);
        LEM_GROUP(y, g, 8, a.rec + 12 * CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[T][0][r] = a.dt * sigmoidf_(g[T][0][r]);
        // lin = Wz [z ; x_t] + bz -> y <- (1 - dt_bar) y + dt_bar tanh(lin)
        lem_acc_init<NS>(a, 3, lane, hh, x, acc);
        LEM_GROUP(z, acc, 12, after)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                y[T][0][r] = (1.0f - g[T][0][r]) * y[T][0][r] + g[T][0][r] * tanhf_(acc[T][0][r]);
    }
#undef LEM_GROUP

    if (a.with_mlp) {
        // h = Swish(Wb Swish(Wa y + ba) + bb); mlp chunk j sits in buffer (j & 1) (16 rec chunks per step: even)
        acc_init_bias<1>(a.mlpb, hh, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wstage_load(ws, a.mlp + (size_t)(j + 1) * CHUNK_FLOATS, tid);
            mma_chunk_from_acc<1>(lds + (j & 1) * H * LDW, c, hh, y[j], acc);
            wstage_store(ws, lds + ((j + 1) & 1) * H * LDW, tid);
            __syncthreads();
        }
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[T][0][r] = swishf(acc[T][0][r]);
        acc_init_bias<1>(a.mlpb + H, hh, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) wstage_load(ws, a.mlp + (size_t)(4 + j + 1) * CHUNK_FLOATS, tid);
            mma_chunk_from_acc<1>(lds + (j & 1) * H * LDW, c, hh, acc[j], y);
            if (j < 3) {
                wstage_store(ws, lds + ((j + 1) & 1) * H * LDW, tid);
                __syncthreads();
            }
        }
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[T][0][r] = swishf(y[T][0][r]);
    }

    if (n < a.n_nodes) {
        float* o = a.out + (size_t)n * H + 4 * hh;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = y[T][0][4 * q + m];
                *reinterpret_cast<f32x4*>(o + 32 * T + 8 * q) = v;
            }
    }
}


// ----------------------------------------------------------------------------------------------
// WEIGHT-STATIONARY edition (default).  The recurrent weights (4 gates x [128 x 128], fp16 hi + lo = 256 KB) fit the
// CU's register file: a 512-thread workgroup = 4 channel slices x 2 roles, and wave (ks, role) keeps the hi/lo A
// fragments of TWO gate tiles (rows 32 ks .. 32 ks + 31; 128 registers) for the whole kernel:
//     role A: g2 (dt_) and g3 (z candidate)  ->  z <- z + dt s(g2) (tanh(g3) - z)      (owns the z slice)
//     role B: g1 (dt_bar) and lin            ->  y <- y + dt s(g1) (tanh(lin) - y)     (owns the y slice)
// so each state update is wave-local, no weight ever moves after the prologue, and the only LDS traffic is the
// hi/lo state fragments every wave publishes for its 32 channels and all waves read as B operands (0.3 KB per MFMA
// instead of 1 KB with streamed weights, which had the LDS port as busy as the matrix pipe).  A workgroup carries
// two node tiles (64 nodes) in a two-stage software pipeline, one barrier per stage:
//     stage 2t + X:     role A works on (tile X, step t):   reads y_X(t),             publishes z_X(t+1)
//     stage 2t + X + 1: role B works on (tile X, step t):   reads y_X(t), z_X(t+1),   publishes y_X(t+1)
// (y is double-buffered in LDS because role-B waves still read y_X(t) while others publish y_X(t+1)).  Both waves of a
// SIMD always have two gate GEMMs (48 + input MFMAs) and 16 registers of activations per stage, and one's VALU
// overlaps the other's matrix work.  The input columns W[:, H:] x_t are one K=16 fp16 MFMA per gate for ninp <= 5
// (two for ninp <= 8) through the slot pairing of lem_slot_feature.  s(a) tanh(b) is evaluated with ONE reciprocal:
//     e_a = 2^(-a log2 e), e_b = 2^(-2 b log2 e), r = 1 / ((1 + e_a)(1 + e_b)):   st += dt r ((1 - e_b) - st (1 + e_b))
// (exponents clamped at 60 so the product stays finite).
// ----------------------------------------------------------------------------------------------
struct LemWsArgs {
    const float* xin;       // MODE 0: [N, T, 2*NS] assembled step inputs
    // MODE 1 / 2: the step inputs are assembled in the kernel from the node arrays (experiments/models_gnn.py:1357-1360:
    // x_t = [pos_x, u_t, variables]; models_gnn2D.py:429-433: x_t = [pos_x, u_t, u_{tw+t}, cumsum(dt)_t + pos_t, variables[1:]])
    const float* u;         // [N, tw] (2-D: [N, 2 tw])
    const float* pos_x;     // [N]
    const float* pos_t;     // [N]      (2-D only)
    const float* vars;      // [N, nv]
    const float* dt_cum;    // [tw]     (2-D only)
    int tw, nv;
    long n_nodes;
    int t_len, with_mlp;
    float dt;
    const float* rec_s;     // 16 split chunks (g2, g3, g1, lin)
    const float* mlp_s;     // 8 split chunks (Wa, Wb)
    const float* bias_s;    // [512] g1, g2, g3, bz (scaled)
    const float* wx_h;      // slot fragments
    const float* mlpb_s;    // [256]
    const float* scales;    // [8]
    float* out;
    int full_wgs = 0x7fffffff;   // ws3: workgroups [0, full_wgs) take three 32-node tiles each, the ones behind them one tile each
};

constexpr int LEM_WS_FR = 1024;      // half8 per (tile) fragment area: [kt 4][s 2][plane 2][lane 64]

// Step inputs of one node.  MODE 0: from the assembled tensor.  MODE 1 / 2: the per-node constants (pos_x, variables; pos_t)
// come from a 2-KB LDS table filled in the prologue (kept out of registers: hoisted into the time loop's live range they
// spilled), the time-dependent entries (u_t; u_{tw+t}, dt_cum_t + pos_t) are loaded per step.
template <int P, int MODE>
__device__ __forceinline__ void lem_ws_load_x(const LemWsArgs& a, const float* xc, long node, int t, float (&x)[2 * ((P + 1) / 2)]) {
    constexpr int NS = (P + 1) / 2;
    if (MODE == 0) {
        const f32x2* p = reinterpret_cast<const f32x2*>(a.xin + ((size_t)node * a.t_len + t) * (2 * NS));
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const f32x2 v = p[i];
            x[2 * i] = v[0];
            x[2 * i + 1] = v[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(xc + 2 * i);
            x[2 * i] = v[0];
            x[2 * i + 1] = v[1];
        }
        if (MODE == 1) {
            x[1] = a.u[(size_t)node * a.tw + t];
        } else {
            x[1] = a.u[(size_t)node * 2 * a.tw + t];
            x[2] = a.u[(size_t)node * 2 * a.tw + a.tw + t];
            x[3] = a.dt_cum[t] + x[3];
        }
    }
}

// B fragments of the input MFMAs from one node's step inputs
template <int P>
__device__ __forceinline__ void lem_ws_slots(const float (&x)[2 * ((P + 1) / 2)], int hh, half8 (&bx)[(3 * P + 2 + 15) / 16]) {
    constexpr int M = (3 * P + 2 + 15) / 16;
    _Float16 xh[P], xl[P];
#pragma unroll
    for (int f = 0; f < P; ++f) {
        xh[f] = (_Float16)x[f];
        xl[f] = (_Float16)(x[f] - (float)xh[f]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s0 = 16 * m + j, s1 = 16 * m + 8 + j;
            const _Float16 v0 = s0 < 3 * P ? (s0 / P == 1 ? xl[s0 % P] : xh[s0 % P]) : (_Float16)(s0 < 3 * P + 2 ? 1.f : 0.f);
            const _Float16 v1 = s1 < 3 * P ? (s1 / P == 1 ? xl[s1 % P] : xh[s1 % P]) : (_Float16)(s1 < 3 * P + 2 ? 1.f : 0.f);
            bx[m][j] = hh ? v1 : v0;
        }
}

// The B fragments of a step's input MFMAs from the tile's CONSTANT fragments (every slot of the static features -- pos_x, the variables,
// the bias ones -- filled once in the prologue, the slots of the time-dependent features zero) and the step's dynamic values xd:
// MODE 1: xd[0] = u_t (feature 1); MODE 2: xd = (u_t, u_{tw+t}, dt_cum_t + pos_t) (features 1, 2, 3).  A dynamic feature f sits in the
// slots f (hi), P + f (lo), 2 P + f (hi) of lem_slot_feature; slot s is element s & 7 of fragment s >> 4 on the lanes with hh = (s >> 3) & 1.
// Same halves as lem_ws_slots writes there: the same bits, for ~12 (MODE 1) / ~36 vector instructions instead of ~40 / ~80.
template <int P, int MODE>
__device__ __forceinline__ void lem_ws3_patch(const float (&xd)[MODE == 1 ? 1 : 3], int hh, half8 (&bx)[(3 * P + 2 + 15) / 16]) {
    constexpr int ND = MODE == 1 ? 1 : 3;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int f = 1 + k;
        const _Float16 h = (_Float16)xd[k];
        const _Float16 l = (_Float16)(xd[k] - (float)h);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int s = p * P + f, m = s >> 4, hhs = (s >> 3) & 1, j = s & 7;
            bx[m][j] = hh == hhs ? (p == 1 ? l : h) : bx[m][j];
        }
    }
}

__device__ __forceinline__ void lem_ws_bias(const float* bl, int hh, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 8 * q + 4 * hh);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[4 * q + m] = bv[m];
    }
}

// publish a [32 channels x 32 nodes] state tile as hi/lo B fragments of K tile `ks`
__device__ __forceinline__ void lem_ws_publish(const f32x16& st, half8* area, int ks, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = st[8 * s + j];
        half8 hi, lo;
        split8(v, hi, lo);
        area[((ks * 2 + s) * 2 + 0) * 64 + lane] = hi;
        area[((ks * 2 + s) * 2 + 1) * 64 + lane] = lo;
    }
}

__device__ __forceinline__ float vmin(float a, float b) {            // bare v_min_f32 (fminf adds a canonicalising v_max)
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
struct LemActConst {
    f32x2 c0, c1, idt, one;
};

// st <- st + dt s(a0) (tanh(a1) - st) for one [32 x 32] tile, then publish it as hi/lo B fragments of K tile `ks`.
// One reciprocal per value: e_a = 2^(c0 a0), e_b = 2^(min(c1 a1, 60)), r = dt / ((1 + e_a)(1 + e_b)),
// st += r ((1 - e_b) - st (1 + e_b)); e_a = inf gives r = 0 (st unchanged), the clamp keeps (1 - e_b) r finite.
__device__ __forceinline__ void lem_ws_update_publish(const f32x16& a0, const f32x16& a1, const LemActConst& k, f32x16& st,
                                                      half8* area, int ks, int lane) {
    // Staged per half tile, with scheduling barriers between the stages: the compiler's hazard recogniser does not look
    // inside inline asm, so an accumulator (MFMA result) is first touched by a compiler-emitted multiply, and every
    // transcendental result is consumed a whole stage later.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x2 ta[4], tb[4], ea[4], eb[4], qb[4], q[4], rr[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = 8 * s + 2 * jj;
            ta[jj] = f32x2{a0[r] * k.c0[0], a0[r + 1] * k.c0[0]};
            tb[jj] = f32x2{fminf(a1[r] * k.c1[0], 60.f), fminf(a1[r + 1] * k.c1[0], 60.f)};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            ea[jj] = f32x2{msmp_exp2(ta[jj][0]), msmp_exp2(ta[jj][1])};
            eb[jj] = f32x2{msmp_exp2(tb[jj][0]), msmp_exp2(tb[jj][1])};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            qb[jj] = pk_add(eb[jj], k.one);
            q[jj] = pk_mul(pk_fma(ea[jj], k.idt, k.idt), qb[jj]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) rr[jj] = f32x2{msmp_rcp(q[jj][0]), msmp_rcp(q[jj][1])};
        f32x2 tt[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = 8 * s + 2 * jj;
            tt[jj] = pk_fnma(f32x2{st[r], st[r + 1]}, qb[jj], pk_sub(k.one, eb[jj]));
        }
        __builtin_amdgcn_sched_barrier(0);
        half8 phi, plo;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = 8 * s + 2 * jj;
            const f32x2 sv = pk_fma(rr[jj], tt[jj], f32x2{st[r], st[r + 1]});
            st[r] = sv[0];
            st[r + 1] = sv[1];
            const half2 hp = __builtin_convertvector(sv, half2);
            const half2 lp = split_lo_pair(hp, sv);
            phi[2 * jj] = hp[0];
            phi[2 * jj + 1] = hp[1];
            plo[2 * jj] = lp[0];
            plo[2 * jj + 1] = lp[1];
        }
        area[((ks * 2 + s) * 2 + 0) * 64 + lane] = phi;
        area[((ks * 2 + s) * 2 + 1) * 64 + lane] = plo;
    }
}

// The state update of the anti-phased kernel: plain (unpacked) fp32 instructions.  Its vector halves run BESIDE the partner wave's
// MFMAs, where a packed-fp32 instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) waits for the matrix pipe: measured 2 700
// cycles per vector half with the packed form of lem_ws_update_publish against 1 500 for the matrix half (scripts/prof_lem.py).
// Two value pairs per stage (half the live temporaries of lem_ws_update_publish: this kernel keeps a work item's accumulators
// across a barrier and one more state tile).
__device__ __forceinline__ void lem_ws_update_publish_q(const f32x16& a0, const f32x16& a1, const LemActConst& k, f32x16& st,
                                                        half8* area, int ks, int lane) {
    const float c0 = k.c0[0], c1 = k.c1[0], idt = k.idt[0];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 phi, plo;
        // eight independent value pipelines, no staging barriers: the scheduler interleaves them within the registers it has
        // (a dependent VALU instruction issues 8 cycles after its producer, an independent one after 4.4: scripts/micro/valu_issue.hip)
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 sv;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 8 * s + i + e;
                const float ea = msmp_exp2(a0[r] * c0);
                const float eb = msmp_exp2(vmin(a1[r] * c1, 60.f));
                const float qb = eb + 1.0f;
                const float rr = msmp_rcp(__builtin_fmaf(ea, idt, idt) * qb);
                sv[e] = __builtin_fmaf(rr, __builtin_fmaf(-st[r], qb, 1.0f - eb), st[r]);
                st[r] = sv[e];
            }
            const half2 hp = __builtin_convertvector(sv, half2);
            const half2 lp = split_lo_pair(hp, sv);
            phi[i] = hp[0];
            phi[i + 1] = hp[1];
            plo[i] = lp[0];
            plo[i + 1] = lp[1];
        }
        area[((ks * 2 + s) * 2 + 0) * 64 + lane] = phi;
        area[((ks * 2 + s) * 2 + 1) * 64 + lane] = plo;
    }
}

// sched_barrier mask: everything may cross except MFMAs (and the catch-all ALU class that contains them)
constexpr int LEM_SCHED_NOT_MFMA = 0x7F6;
// acc0 += W0 B0, acc1 += W1 B1 over K = 128 (B0/B1: published fragment areas; they may be the same area)
template <bool SAME>
__device__ __forceinline__ void lem_ws_gemm2(const half8 (&w0)[4][2][2], const half8 (&w1)[4][2][2], const half8* b0, const half8* b1,
                                             int lane, f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const half8 h0 = b0[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = b0[((kt * 2 + s) * 2 + 1) * 64 + lane];
            half8 h1 = h0, l1 = l0;
            if (!SAME) {
                h1 = b1[((kt * 2 + s) * 2 + 0) * 64 + lane];
                l1 = b1[((kt * 2 + s) * 2 + 1) * 64 + lane];
            }
            // three back-to-back MFMAs per accumulator: a dependent MFMA issued right behind its producer accumulates in
            // place; alternating the two accumulators made every MFMA wait for the previous write-back (2x slower)
            MSMP_MFMA_LOLO(2, acc0, w0[kt][s][1], l0);
            MSMP_MFMA_LOLO(2, acc1, w1[kt][s][1], l1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][1], h0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], l0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], h0, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(LEM_SCHED_NOT_MFMA);      // left alone the scheduler alternates acc0 / acc1 when both read the same fragments
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][1], h1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], l1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], h1, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(LEM_SCHED_NOT_MFMA);
        }
}

template <int P, int MODE>
__global__ __launch_bounds__(512) void lem_encoder_ws_kernel(LemWsArgs a) {
    constexpr int NS = (P + 1) / 2, M = (3 * P + 2 + 15) / 16;
    // y fragments [buffer 2][tile 2] | z fragments [tile 2] (16 KB each) | scaled biases [512 + 256]
    __shared__ __attribute__((aligned(16))) float lds[6 * SPLIT_CHUNK_FLOATS + 768];
    __shared__ __attribute__((aligned(16))) float xconst[64 * 8];
    half8* const yfr = reinterpret_cast<half8*>(lds);
    half8* const zfr = yfr + 4 * LEM_WS_FR;
    float* const bias_l = lds + 6 * SPLIT_CHUNK_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave & 3, role = wave >> 2;
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = (long)blockIdx.x * 64;
    const float LOG2E = 1.44269504088896340736f;
    const float inv_w = a.scales[4], inv_z = a.scales[5];

    // prologue: zero y(0) of both tiles (buffer 0), stage the biases, load this wave's stationary weights
    {
        half8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (_Float16)0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) yfr[tid + 512 * i] = zero;
        bias_l[tid] = a.bias_s[tid];
        if (tid < 256) bias_l[512 + tid] = a.mlpb_s[tid];
        if (MODE != 0 && tid < 64) {        // per-node constants of the 64 nodes: [pos_x, (u_t), variables] / [pos_x, (u_t, u_tw+t), pos_t, variables[1:]]
            const long nn = n0 + tid < a.n_nodes ? n0 + tid : a.n_nodes - 1;
            float* row = xconst + 8 * tid;
#pragma unroll
            for (int f = 0; f < 8; ++f) row[f] = 0.f;
            row[0] = a.pos_x[nn];
            if (MODE == 1) {
                for (int f = 0; f < a.nv; ++f) row[2 + f] = a.vars[(size_t)nn * a.nv + f];
            } else {
                row[3] = a.pos_t[nn];
                for (int f = 1; f < a.nv; ++f) row[3 + f] = a.vars[(size_t)nn * a.nv + f];
            }
        }
    }
    half8 w[2][4][2][2];
    half8 wxh[2][M];
    {
        const half8* rs = reinterpret_cast<const half8*>(a.rec_s);
        const half8* wh = reinterpret_cast<const half8*>(a.wx_h);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = 2 * role + gi;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[gi][kt][s][pl] = rs[(size_t)(grp * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
#pragma unroll
            for (int m = 0; m < M; ++m) wxh[gi][m] = wh[((grp * 4 + ks) * 2 + m) * 64 + lane];
        }
    }
    // exponent constants: gate 0 is the sigmoid gate (W scale); gate 1 the tanh candidate (role A: W, role B: Wz)
    const float c0 = -inv_w * LOG2E, c1 = -2.0f * (role ? inv_z : inv_w) * LOG2E, idt = 1.0f / a.dt;
    const LemActConst kc{f32x2{c0, c0}, f32x2{c1, c1}, f32x2{idt, idt}, f32x2{1.0f, 1.0f}};

    f32x16 st[2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[X][r] = 0.f;

    long node[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const long n = n0 + 32 * X + c;
        node[X] = n < a.n_nodes ? n : a.n_nodes - 1;
    }
    float xn[2 * NS];
    __syncthreads();                    // the constants table is read below
    lem_ws_load_x<P, MODE>(a, xconst + 8 * c, node[0], 0, xn);
    __syncthreads();
    if (role) __syncthreads();          // role B runs one stage behind role A

    for (int t = 0; t < a.t_len; ++t) {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            half8 bx[M];
            lem_ws_slots<P>(xn, hh, bx);
            {   // prefetch the inputs of this wave's next work item: (tile 1, t) or (tile 0, t + 1)
                const int tn = X ? (t + 1 < a.t_len ? t + 1 : t) : t;
                lem_ws_load_x<P, MODE>(a, xconst + 8 * (32 * (X ^ 1) + c), node[X ^ 1], tn, xn);
            }
            // bias + input columns from a zero accumulator input (the bias rides in two K slots paired with 1.0)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxh[0][0], bx[0], zero, 0, 0, 0);
            f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxh[1][0], bx[0], zero, 0, 0, 0);
#pragma unroll
            for (int m = 1; m < M; ++m) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxh[0][m], bx[m], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxh[1][m], bx[m], acc1, 0, 0, 0);
            }
            const half8* yb = yfr + ((t & 1) * 2 + X) * LEM_WS_FR;
            if (role) lem_ws_gemm2<false>(w[0], w[1], yb, zfr + X * LEM_WS_FR, lane, acc0, acc1);
            else lem_ws_gemm2<true>(w[0], w[1], yb, yb, lane, acc0, acc1);
            lem_ws_update_publish(acc0, acc1, kc, st[X], role ? yfr + (((t + 1) & 1) * 2 + X) * LEM_WS_FR : zfr + X * LEM_WS_FR, ks, lane);
            __syncthreads();
        }
    }
    if (!role) __syncthreads();         // role A's idle last stage
    // here: y_X(T) of both tiles is published in buffer T & 1; role-B waves hold their y slices in st[]

    const int X = role;                 // lemoutput_mlp: role A takes tile 0, role B tile 1
    f32x16 res;
    if (a.with_mlp) {
        const half8* ms = reinterpret_cast<const half8*>(a.mlp_s);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[gi][kt][s][pl] = ms[(size_t)(gi * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
        const float invA = a.scales[6], invB = a.scales[7];
        const half8* yb = yfr + ((a.t_len & 1) * 2 + X) * LEM_WS_FR;
        half8* hb = zfr + X * LEM_WS_FR;
        f32x16 acc;
        lem_ws_bias(bias_l + 512 + 32 * ks, hh, acc);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 h0 = yb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = yb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                MSMP_MFMA_LOLO(2, acc, w[0][kt][s][1], l0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][1], h0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], l0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], h0, acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = swishf(acc[r] * invA);
        lem_ws_publish(acc, hb, ks, lane);
        __syncthreads();
        lem_ws_bias(bias_l + 512 + H + 32 * ks, hh, res);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 h0 = hb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = hb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                MSMP_MFMA_LOLO(2, res, w[1][kt][s][1], l0);
                res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][1], h0, res, 0, 0, 0);
                res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], l0, res, 0, 0, 0);
                res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], h0, res, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = swishf(res[r] * invB);
    } else {
        // y itself: the role-B wave of slice ks holds both tiles; it hands tile 0 to its role-A partner through LDS
        if (role) *reinterpret_cast<f32x16*>(lds + (size_t)(ks * 64 + lane) * 16) = st[0];
        __syncthreads();
        if (role) res = st[1];
        else res = *reinterpret_cast<const f32x16*>(lds + (size_t)(ks * 64 + lane) * 16);
    }
    const long n = n0 + 32 * X + c;
    if (n < a.n_nodes) {
        float* o = a.out + (size_t)n * H + 32 * ks + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int m = 0; m < 4; ++m) v[m] = res[4 * q + m];
            *reinterpret_cast<f32x4*>(o + 8 * q) = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// WEIGHT-STATIONARY, ANTI-PHASED edition (default since round 3; msmp_tune("lem", 4)).  Same roles, registers and LDS traffic as
// lem_encoder_ws_kernel, but a workgroup carries THREE node tiles (96 nodes) and every work item is cut into its matrix half M
// (input MFMAs + the two gate GEMMs: 50 MFMAs, accumulators kept in registers) and its vector half V (state update + publish),
// one barrier per half.  The two waves of a SIMD are wave (ks, A) and wave (ks, B); in the two-tile pipeline both ran their
// GEMMs right after the stage's barrier and their activations afterwards -- matrix beside matrix, vector beside vector on
// every SIMD: 6.7 k cycles per stage for 3.2 k cycles of matrix work (rocprofv3 SQ counters, profiles/r03c_*).  Here, per time
// step t and slot j = 0..5:
//     role A:  M(0,t)    V(0,t)     M(1,t)     V(1,t)   M(2,t)   V(2,t)
//     role B:  V(1,t-1)  M(2,t-1)   V(2,t-1)   M(0,t)   V(0,t)   M(1,t)
// so one wave of every SIMD is in a matrix half while the other is in a vector half.  Dependencies (A's V(X,t) publishes
// z_X(t+1), read by B's M(X,t); B's V(X,t) publishes y_X(t+1), read by A's M(X,t+1) and B's M(X,t+1)) are each separated by
// at least one barrier, and every buffer's last reader precedes its next writer by a barrier, so y needs ONE buffer per tile.
// ----------------------------------------------------------------------------------------------
#if MSMP_PROF_LEM
__device__ unsigned long long g_prof_lem[16];
#define LPROF_DECL unsigned lp_m = 0, lp_v = 0, lp_b = 0, lp_w = 0, lp_t = (unsigned)__builtin_readcyclecounter();
#define LPROF(var) do { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); var += t_ - lp_t; lp_t = t_; } while (0)
#define LPROF_FLUSH if (lane == 0 && ks == 0 && (blockIdx.x & 31) == 0) { unsigned long long* o = g_prof_lem + 4 * role; atomicAdd(o, (unsigned long long)lp_m); atomicAdd(o + 1, (unsigned long long)lp_v); atomicAdd(o + 2, (unsigned long long)lp_b); atomicAdd(o + 3, 1ull); atomicAdd(g_prof_lem + 8 + role, (unsigned long long)lp_w); }
#else
#define LPROF_DECL unsigned lp_w = 0; (void)lp_w;
#define LPROF(var)
#define LPROF_FLUSH
#endif
#define LEM_SYNC() do { LPROF(lp_w); __syncthreads(); LPROF(lp_b); } while (0)

// lem_ws_gemm2 with every fragment address formed as  byte base (a __shared__ array)  +  ONE opaque per-lane register  +  a
// compile-time offset that fits ds_read's 16-bit field: left to itself the compiler hoists the ~100 distinct fragment addresses of
// the unrolled time step out of the loop as invariants and spills them (scratch reloads with vmcnt(0) waits between the MFMAs).
template <bool SAME, int OFF0, int OFF1>
__device__ __forceinline__ void lem_ws3_gemm2(const half8 (&w0)[4][2][2], const half8 (&w1)[4][2][2], const char* b0, const char* b1,
                                              f32x16& acc0, f32x16& acc1) {
    if (!SAME) {
        // role B reads two different state tiles (y for acc0, z for acc1): sixteen chains of three MFMAs, each behind its own fragment
        // pair.  The hi fragment of chain n + 1 is requested before chain n's MFMAs, the lo fragment as soon as chain n's second MFMA has
        // consumed its own (4 registers more than the plain form, no chain waits for LDS).  Same products in the same order per accumulator.
        constexpr int FB = 64 * 16;
        auto frag = [&](int n, int plane) -> half8 {        // chain n = 2 * (kt * 2 + s) + (0: acc0 / y, 1: acc1 / z)
            const int f = (n >> 1) * 2 + plane;
            return (n & 1) ? *reinterpret_cast<const half8*>(b1 + OFF1 + f * FB) : *reinterpret_cast<const half8*>(b0 + OFF0 + f * FB);
        };
        half8 h = frag(0, 0), l = frag(0, 1);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int kt = n >> 2, s = (n >> 1) & 1;
            half8 hn = h, ln = l;
            if (n < 15) hn = frag(n + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (n & 1) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][1], h, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], l, acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][1], h, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], l, acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (n < 15) ln = frag(n + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (n & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], h, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], h, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            h = hn;
            l = ln;
        }
        return;
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            constexpr int FB = 64 * 16;       // bytes of one fragment plane
            const int f = (kt * 2 + s) * 2;
            const half8 h0 = *reinterpret_cast<const half8*>(b0 + OFF0 + f * FB), l0 = *reinterpret_cast<const half8*>(b0 + OFF0 + (f + 1) * FB);
            half8 h1 = h0, l1 = l0;
            if (!SAME) {
                h1 = *reinterpret_cast<const half8*>(b1 + OFF1 + f * FB);
                l1 = *reinterpret_cast<const half8*>(b1 + OFF1 + (f + 1) * FB);
            }
            MSMP_MFMA_LOLO(2, acc0, w0[kt][s][1], l0);
            MSMP_MFMA_LOLO(2, acc1, w1[kt][s][1], l1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][1], h0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], l0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kt][s][0], h0, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(LEM_SCHED_NOT_MFMA);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][1], h1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], l1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kt][s][0], h1, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(LEM_SCHED_NOT_MFMA);
        }
}

template <int P, int MODE>
__global__ __launch_bounds__(512) void lem_encoder_ws3_kernel(LemWsArgs a) {
    constexpr int NS = (P + 1) / 2, M = (3 * P + 2 + 15) / 16;
    // y fragments [tile 3] | z fragments [tile 3] (16 KB each) | scaled biases [512 + 256]
    __shared__ __attribute__((aligned(16))) float lds[6 * SPLIT_CHUNK_FLOATS + 768];
    __shared__ __attribute__((aligned(16))) float xconst[96 * 8];
    __shared__ half8 bxc[3 * M * 64];            // constant input fragments of the three tiles: [tile][m][lane]  (lem_ws3_patch; MODE 1 / 2)
    // the input-column fragments (W[:, H:] and the bias slots) of every wave: kept out of the register file, which holds the
    // recurrent weights (128), three state tiles (48) and a work item's accumulators (32)
    __shared__ half8 wxl[8 * 2 * M * 64];
    half8* const yfr = reinterpret_cast<half8*>(lds);
    half8* const zfr = yfr + 3 * LEM_WS_FR;
    float* const bias_l = lds + 6 * SPLIT_CHUNK_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: the role branches become scalar branches
    const int ks = wave & 3, role = wave >> 2;
    const int c = lane & 31, hh = lane >> 5;
    // A launch is cut into rounds of three-tile workgroups (one per CU) and, when that leaves at most a round of single tiles, a
    // last round of ONE-tile workgroups: the recurrence of a tile is the dependent chain M_A -> V_A -> M_B -> V_B per step, three
    // tiles fill it to 6 half-slots per step; a lone tile runs 4 (and with each wave alone on its SIMD): ~0.5 of the time, instead
    // of a last round of three-tile workgroups that keeps a third of the CUs busy for a whole round (msmp_lem_encoder*: lem_partition).
    const bool single = (int)blockIdx.x >= a.full_wgs;
    const long n0 = single ? 96L * a.full_wgs + 32L * ((int)blockIdx.x - a.full_wgs) : (long)blockIdx.x * 96;
    const long n_lim = single ? (n0 + 32 < a.n_nodes ? n0 + 32 : a.n_nodes) : a.n_nodes;      // nodes this workgroup writes
    const int nt = single ? 1 : 3;
    const float LOG2E = 1.44269504088896340736f;
    const float inv_w = a.scales[4], inv_z = a.scales[5];
    const int T = a.t_len;

    {   // prologue: y(0) = 0 for the three tiles, biases, per-node constants
        half8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (_Float16)0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) yfr[tid + 512 * i] = zero;
        bias_l[tid] = a.bias_s[tid];
        if (tid < 256) bias_l[512 + tid] = a.mlpb_s[tid];
        if (MODE != 0 && tid < 96) {
            const long nn = n0 + tid < a.n_nodes ? n0 + tid : a.n_nodes - 1;
            float* row = xconst + 8 * tid;
#pragma unroll
            for (int f = 0; f < 8; ++f) row[f] = 0.f;
            row[0] = a.pos_x[nn];
            if (MODE == 1) {
                for (int f = 0; f < a.nv; ++f) row[2 + f] = a.vars[(size_t)nn * a.nv + f];
            } else {
                row[3] = a.pos_t[nn];
                for (int f = 1; f < a.nv; ++f) row[3 + f] = a.vars[(size_t)nn * a.nv + f];
            }
        }
    }
    half8 w[2][4][2][2];
    half8* const wxw = wxl + (size_t)wave * (2 * M * 64) + lane;       // this wave's: [gate 2][m M][lane 64]
    {
        const half8* rs = reinterpret_cast<const half8*>(a.rec_s);
        const half8* wh = reinterpret_cast<const half8*>(a.wx_h);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = 2 * role + gi;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[gi][kt][s][pl] = rs[(size_t)(grp * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
#pragma unroll
            for (int m = 0; m < M; ++m) wxw[(gi * M + m) * 64] = wh[((grp * 4 + ks) * 2 + m) * 64 + lane];
        }
    }
    const float c0 = -inv_w * LOG2E, c1 = -2.0f * (role ? inv_z : inv_w) * LOG2E, idt = 1.0f / a.dt;
    const LemActConst kc{f32x2{c0, c0}, f32x2{c1, c1}, f32x2{idt, idt}, f32x2{1.0f, 1.0f}};

    f32x16 st[3];
#pragma unroll
    for (int X = 0; X < 3; ++X)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[X][r] = 0.f;
    // LDS fragment addressing: two opaque per-lane byte addresses (y area; z area, beyond ds_read's 64-KB offset field) + constants
    constexpr int ZBASE = 3 * LEM_WS_FR * 16;
    const char* lb_y;
    const char* lb_z;
    {
        unsigned lo = lane * 16;
        asm volatile("" : "+v"(lo));
        lb_y = reinterpret_cast<const char*>(lds) + lo;
        unsigned lo2 = lane * 16 + ZBASE;
        asm volatile("" : "+v"(lo2));
        lb_z = reinterpret_cast<const char*>(lds) + lo2;
    }
    float xn[2 * NS];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    __syncthreads();                    // the constants table and y(0) are read below

    // the two halves of a work item; X is a compile-time tile index
    float xd[MODE == 1 ? 1 : 3];
    if (MODE != 0 && tid < 192) {       // constant fragments of the tiles: the prologue's static features through lem_ws_slots, once
        const int Xt = tid >> 6, l64 = tid & 63;
        float xs[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) xs[f] = xconst[8 * (32 * Xt + (l64 & 31)) + f];
        if (MODE == 2) xs[3] = 0.f;     // (pos_t stays in the table: it is added to dt_cum_t per step)
        half8 b0[M];
        lem_ws_slots<P>(xs, l64 >> 5, b0);
#pragma unroll
        for (int m = 0; m < M; ++m) bxc[(Xt * M + m) * 64 + l64] = b0[m];
    }
    __syncthreads();
    auto fetch_x = [&](auto Xc, int t) {
        constexpr int X = decltype(Xc)::value;
        const long n = n0 + 32 * X + c;
        if (MODE != 0) {
            const long node = n < a.n_nodes ? n : a.n_nodes - 1;
            if (MODE == 1) {
                xd[0] = a.u[(size_t)node * a.tw + t];
            } else {
                xd[0] = a.u[(size_t)node * 2 * a.tw + t];
                xd[MODE == 1 ? 0 : 1] = a.u[(size_t)node * 2 * a.tw + a.tw + t];
                xd[MODE == 1 ? 0 : 2] = a.dt_cum[t] + xconst[8 * (32 * X + c) + 3];
            }
            return;
        }
        lem_ws_load_x<P, MODE>(a, xconst + 8 * (32 * X + c), n < a.n_nodes ? n : a.n_nodes - 1, t, xn);
    };
    auto half_m = [&](auto Xc) {
        constexpr int X = decltype(Xc)::value;
        half8 bx[M];
        // the step input fetched at the head of the PREVIOUS half-slot is first touched here: without the fence the compiler hoists its
        // fp16 split into that vector half (a few dozen instructions behind the load) and the half stalls on the load's latency
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 0) {        // the tile's constant fragment + the step's one or three dynamic features (lem_ws3_patch)
#pragma unroll
            for (int m = 0; m < M; ++m) bx[m] = bxc[(X * M + m) * 64 + lane];
            lem_ws3_patch<P, MODE>(xd, hh, bx);
        } else {
            lem_ws_slots<P>(xn, hh, bx);
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxw[0], bx[0], zero, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxw[M * 64], bx[0], zero, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < M; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxw[m * 64], bx[m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wxw[(M + m) * 64], bx[m], acc1, 0, 0, 0);
        }
        constexpr int YO = X * LEM_WS_FR * 16, ZO = (3 + X) * LEM_WS_FR * 16 - ZBASE;
        if (role) lem_ws3_gemm2<false, YO, ZO>(w[0], w[1], lb_y, lb_z, acc0, acc1);
        else lem_ws3_gemm2<true, YO, YO>(w[0], w[1], lb_y, lb_y, acc0, acc1);
    };
    auto half_v = [&](auto Xc) {
        constexpr int X = decltype(Xc)::value;
        lem_ws_update_publish_q(acc0, acc1, kc, st[X], (role ? yfr : zfr) + X * LEM_WS_FR, ks, lane);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    LPROF_DECL
    if (single) {
        // one tile: four half-slots per step, A's matrix / vector halves, then B's (4 T barriers for either role)
        if (!role) {
            fetch_x(I0{}, 0);
            for (int t = 0; t < T; ++t) {
                { half_m(I0{}); LPROF(lp_m); }                                                   LEM_SYNC();      // M_A
                fetch_x(I0{}, t + 1 < T ? t + 1 : t); { half_v(I0{}); LPROF(lp_v); }             LEM_SYNC();      // V_A: z(t)
                LEM_SYNC();
                LEM_SYNC();
            }
        } else {
            for (int t = 0; t < T; ++t) {
                LEM_SYNC();
                fetch_x(I0{}, t);                                                               LEM_SYNC();
                { half_m(I0{}); LPROF(lp_m); }                                                   LEM_SYNC();      // M_B (reads z(t))
                { half_v(I0{}); LPROF(lp_v); }                                                   LEM_SYNC();      // V_B: y(t)
            }
        }
    } else
    // Two separate instruction streams (the role is wave-uniform): each executes 6 T + 3 barriers.
    if (!role) {
        fetch_x(I0{}, 0);
        for (int t = 0; t < T; ++t) {
            { half_m(I0{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 0
            fetch_x(I1{}, t); { half_v(I0{}); LPROF(lp_v); }                         LEM_SYNC();      // slot 1
            { half_m(I1{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 2
            fetch_x(I2{}, t); { half_v(I1{}); LPROF(lp_v); }                         LEM_SYNC();      // slot 3
            { half_m(I2{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 4
            fetch_x(I0{}, t + 1 < T ? t + 1 : t); { half_v(I2{}); LPROF(lp_v); }     LEM_SYNC();      // slot 5
        }
        LEM_SYNC();
        LEM_SYNC();
        LEM_SYNC();
    } else {
        // t = 0: slots 0-2 have no work yet (the items of step -1)
        LEM_SYNC();
        LEM_SYNC();
        fetch_x(I0{}, 0);                                           LEM_SYNC();
        { half_m(I0{}); LPROF(lp_m); }                                               LEM_SYNC();      // slot 3
        fetch_x(I1{}, 0); { half_v(I0{}); LPROF(lp_v); }                             LEM_SYNC();      // slot 4
        { half_m(I1{}); LPROF(lp_m); }                                               LEM_SYNC();      // slot 5
        for (int t = 1; t < T; ++t) {
            fetch_x(I2{}, t - 1); { half_v(I1{}); LPROF(lp_v); }                     LEM_SYNC();      // slot 0: V(1, t-1)
            { half_m(I2{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 1: M(2, t-1)
            fetch_x(I0{}, t); { half_v(I2{}); LPROF(lp_v); }                         LEM_SYNC();      // slot 2: V(2, t-1)
            { half_m(I0{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 3: M(0, t)
            fetch_x(I1{}, t); { half_v(I0{}); LPROF(lp_v); }                         LEM_SYNC();      // slot 4: V(0, t)
            { half_m(I1{}); LPROF(lp_m); }                                           LEM_SYNC();      // slot 5: M(1, t)
        }
        fetch_x(I2{}, T - 1); { half_v(I1{}); LPROF(lp_v); }                         LEM_SYNC();      // V(1, T-1)
        { half_m(I2{}); LPROF(lp_m); }                                               LEM_SYNC();      // M(2, T-1)
        { half_v(I2{}); LPROF(lp_v); }                                               LEM_SYNC();      // V(2, T-1)
    }
    LPROF_FLUSH
    // here: y_X(T) of the three tiles is published in yfr; role-B waves hold their y slices in st[]

    if (a.with_mlp) {
        const half8* ms = reinterpret_cast<const half8*>(a.mlp_s);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[gi][kt][s][pl] = ms[(size_t)(gi * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
        const float invA = a.scales[6], invB = a.scales[7];
        // lemoutput_mlp: pass 0: role A takes tile 0, role B tile 1; pass 1: role A takes tile 2
        for (int pass = 0; pass < 2; ++pass) {
            const int X = pass ? 2 : role;
            const bool work = (pass == 0 || role == 0) && X < nt;
            const half8* yb = yfr + X * LEM_WS_FR;
            half8* hb = zfr + X * LEM_WS_FR;
            if (work) {
                f32x16 acc;
                lem_ws_bias(bias_l + 512 + 32 * ks, hh, acc);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const half8 h0 = yb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = yb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                        MSMP_MFMA_LOLO(2, acc, w[0][kt][s][1], l0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][1], h0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], l0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], h0, acc, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = swishf(acc[r] * invA);
                lem_ws_publish(acc, hb, ks, lane);
            }
            __syncthreads();
            if (work) {
                f32x16 res;
                lem_ws_bias(bias_l + 512 + H + 32 * ks, hh, res);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const half8 h0 = hb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = hb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                        MSMP_MFMA_LOLO(2, res, w[1][kt][s][1], l0);
                        res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][1], h0, res, 0, 0, 0);
                        res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], l0, res, 0, 0, 0);
                        res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], h0, res, 0, 0, 0);
                    }
                const long n = n0 + 32 * X + c;
                if (n < n_lim) {
                    float* o = a.out + (size_t)n * H + 32 * ks + 4 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = swishf(res[4 * q + m] * invB);
                        *reinterpret_cast<f32x4*>(o + 8 * q) = v;
                    }
                }
            }
        }
    } else if (role) {
        // y itself: the role-B wave of slice ks holds its 32 channels of all three tiles
#pragma unroll
        for (int X = 0; X < 3; ++X) {
            const long n = n0 + 32 * X + c;
            if (n < n_lim) {
                float* o = a.out + (size_t)n * H + 32 * ks + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = st[X][4 * q + m];
                    *reinterpret_cast<f32x4*>(o + 8 * q) = v;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// WEIGHT-STATIONARY, ONE WAVE PER SIMD edition (round 4; msmp_tune("lem", 5)).  The issue-port measurements of round 3
// (scripts/micro/mfma_valu_overlap.hip) say what overlaps on a SIMD: a wave's OWN vector instructions issued between its MFMAs are
// free, vector and matrix work of DIFFERENT waves add up.  The anti-phased kernel above pairs a matrix half of one wave with a
// vector half of the other wave of the SIMD -- the sum, 2 150-2 400 clocks per slot for 1 600 clocks of MFMAs.  Here a workgroup
// is FOUR waves, one per SIMD, with the whole register file each (512 registers: all four gate blocks of the wave's 32 channels,
// 256 registers of stationary weights, live in it), every wave plays both roles for its channel slice, and a wave's matrix work of
// one node tile is interleaved IN PROGRAM ORDER with its vector work of the other tile.  Two tiles (64 nodes) per workgroup, four
// fused phases per time step, one barrier per phase:
//     phase 0:  M_A(0,t)  ||  V_B(1,t-1)        M_A(X,t): g2, g3 of tile X from y_X(t)            (50 MFMAs)
//     phase 1:  M_A(1,t)  ||  V_A(0,t)          V_A(X,t): z_X <- z + dt s(g2)(tanh(g3) - z), published as hi/lo fragments
//     phase 2:  M_B(0,t)  ||  V_A(1,t)          M_B(X,t): g1 from y_X(t), lin from z_X(t+1)       (50 MFMAs)
//     phase 3:  M_B(1,t)  ||  V_B(0,t)          V_B(X,t): y_X <- y + dt s(g1)(tanh(lin) - y), published
// Every fragment area's last reader precedes its next writer by a barrier (y_0: read in phases 0 and 2, written in 3; y_1: read
// in 1 and 3, written in 0; z_0: read in 2, written in 1; z_1: read in 3, written in 2), so one buffer per tile and state.
// The chain M_A -> V_A -> M_B -> V_B of a tile is four phases, two tiles offset by one phase keep the matrix pipe fed in every
// phase: 4 x 50 MFMAs per step and wave = the MFMA-bound 6.4 k clocks if the vector work hides (264 issue slots of the ~300 behind
// 50 MFMAs).  Same per-value arithmetic as the other weight-stationary editions: a tile's result does not depend on the edition.
// ----------------------------------------------------------------------------------------------
// acc += W B with the stationary weight fragment read STRAIGHT from the accumulation registers (AGPRs): the 256 registers of weights
// live there for the whole kernel, the 256 architectural registers hold states, accumulators and the vector work.  (Left to the
// register allocator the weights end up in AGPRs as SPILL slots: four v_accvgpr_read per fragment in front of every MFMA.)
// Inline asm is invisible to the hazard recogniser: B comes from LDS reads (waited for by the compiler), acc is read by vector
// instructions only a whole phase (and a barrier) later, and a B-fragment register is next written by an LDS read issued at least one
// K group (six MFMAs) after the MFMA that read it.  (A queued dependent MFMA reads its A / B operands when it STARTS, tens of clocks
// after it issued: round 4 measured run-to-run differences of 1e-1 in the node tail when an inline-asm split overwrote B registers two
// instructions behind the MFMAs that read them.  This edition is opt-in -- msmp_tune("lem", 5) -- and checked bit for bit against the
// default edition in tests/test_gpu_kernels.py; the default edition's MFMAs are compiler builtins.)
__device__ __forceinline__ void mfma_aw(f32x16& acc, const half8& w_agpr, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_agpr), "v"(b));
}
// One value pair (r, r + 1) of a state update st <- st + dt s(a0) (tanh(a1) - st), cut into SIX pieces of 3-6 instructions, one
// behind each MFMA of a K group (a dependent vector instruction issues ~8 clocks after its producer, a transcendental later still:
// every piece only consumes what the previous piece -- a whole MFMA earlier -- produced).  Same operations in the same order as
// lem_ws_update_publish_q: the same bits.
struct LemPairTmp {
    float ta[2], tb[2], ea[2], eb[2], qb[2], m[2], rr[2], d[2];
    f32x2 sv;
};
template <int PIECE>
__device__ __forceinline__ void lem_ws1_piece(const f32x16& a0, const f32x16& a1, int r, float c0, float c1, float idt, f32x16& st,
                                              LemPairTmp& p, half8& phi, half8& plo, int j) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (PIECE == 0) {
            p.ta[e] = a0[r + e] * c0;
            p.tb[e] = vmin(a1[r + e] * c1, 60.f);
        } else if (PIECE == 1) {
            p.ea[e] = msmp_exp2(p.ta[e]);
            p.eb[e] = msmp_exp2(p.tb[e]);
        } else if (PIECE == 2) {
            p.qb[e] = p.eb[e] + 1.0f;
            p.m[e] = __builtin_fmaf(p.ea[e], idt, idt) * p.qb[e];
        } else if (PIECE == 3) {
            p.rr[e] = msmp_rcp(p.m[e]);
            p.d[e] = 1.0f - p.eb[e];
        } else if (PIECE == 4) {
            p.sv[e] = __builtin_fmaf(p.rr[e], __builtin_fmaf(-st[r + e], p.qb[e], p.d[e]), st[r + e]);
            st[r + e] = p.sv[e];
        }
    }
    if (PIECE == 5) {
        const half2 hp = __builtin_convertvector(p.sv, half2);
        const half2 lp = split_lo_pair(hp, p.sv);
        phi[j] = hp[0];
        phi[j + 1] = hp[1];
        plo[j] = lp[0];
        plo[j + 1] = lp[1];
    }
}

// One fused phase: (o0, o1) = input MFMAs + the two gate GEMMs over K = 128 of one tile;  beside them, when DO_V, the state update of
// ANOTHER tile from (i0, i1) -- accumulators a previous phase produced -- and its publication.  Eight K groups of 6 MFMAs; behind
// EVERY MFMA one piece of the group's value pair, pinned there with scheduling barriers (left alone the scheduler gathers the
// vector instructions in front of three back-to-back MFMAs: measured 2.9 k clocks per phase instead of the MFMAs' 1.6 k).
template <bool SAME, bool DO_V, int M>
__device__ __forceinline__ void lem_ws1_phase(const half8 (&w0)[4][2][2], const half8 (&w1)[4][2][2], const half8* wx0, const half8* wx1,
                                              const half8 (&bx)[M], const char* fb0, const char* fb1, f32x16& o0, f32x16& o1,
                                              const f32x16& i0, const f32x16& i1, float c0, float c1, float idt, f32x16& st, char* pub) {
    constexpr int FB = 64 * 16;       // bytes of one fragment plane
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 h0 = *reinterpret_cast<const half8*>(fb0), l0 = *reinterpret_cast<const half8*>(fb0 + FB);
    half8 h1 = h0, l1 = l0;
    if (!SAME) {
        h1 = *reinterpret_cast<const half8*>(fb1);
        l1 = *reinterpret_cast<const half8*>(fb1 + FB);
    }
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx0[0], bx[0], zero, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx1[0], bx[0], zero, 0, 0, 0);
#pragma unroll
    for (int m = 1; m < M; ++m) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx0[m * 64], bx[m], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx1[m * 64], bx[m], o1, 0, 0, 0);
    }
    half8 phi, plo;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int kt = g >> 1, s = g & 1;
        const int r = 2 * g, j = 2 * (g & 3);      // the value pair (r, r + 1) of the update: half g >> 2, elements j, j + 1
        LemPairTmp tmp;
        half8 nh0 = h0, nl0 = l0, nh1 = h1, nl1 = l1;
        __builtin_amdgcn_sched_barrier(0);
        // The two accumulators ALTERNATE: a dependent MFMA that is not issued right behind its producer waits for the producer's
        // write-back (~64 clocks after issue: the anti-phased kernel measured 2x for alternating accumulators with nothing between
        // them; with one vector piece between the MFMAs the dependent one is two MFMAs and two pieces behind -- nothing waits, and
        // each chain still accumulates in its own order (w_lo h, w_hi l, w_hi h per K group): the same bits).
        if (MSMP_LOLO >= 2) { mfma_aw(o0, w0[kt][s][1], l0); mfma_aw(o1, w1[kt][s][1], l1); }
        mfma_aw(o0, w0[kt][s][1], h0);
        if (g < 7) {                                // next group's fragments: requested behind this group's first MFMA
            const int f = (g + 1) * 2;
            nh0 = *reinterpret_cast<const half8*>(fb0 + f * FB);
            nl0 = *reinterpret_cast<const half8*>(fb0 + (f + 1) * FB);
            if (!SAME) {
                nh1 = *reinterpret_cast<const half8*>(fb1 + f * FB);
                nl1 = *reinterpret_cast<const half8*>(fb1 + (f + 1) * FB);
            }
        }
        if (DO_V) lem_ws1_piece<0>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_aw(o1, w1[kt][s][1], h1);
        if (DO_V) lem_ws1_piece<1>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_aw(o0, w0[kt][s][0], l0);
        if (DO_V) lem_ws1_piece<2>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_aw(o1, w1[kt][s][0], l1);
        if (DO_V) lem_ws1_piece<3>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_aw(o0, w0[kt][s][0], h0);
        if (DO_V) lem_ws1_piece<4>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma_aw(o1, w1[kt][s][0], h1);
        if (DO_V) {
            lem_ws1_piece<5>(i0, i1, r, c0, c1, idt, st, tmp, phi, plo, j);
            if ((g & 3) == 3) {                     // a half tile (8 values) is complete: publish its hi / lo fragments
                const int sv = g >> 2;
                *reinterpret_cast<half8*>(pub + (sv * 2 + 0) * FB) = phi;
                *reinterpret_cast<half8*>(pub + (sv * 2 + 1) * FB) = plo;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        h0 = nh0; l0 = nl0;
        if (SAME) { h1 = nh0; l1 = nl0; } else { h1 = nh1; l1 = nl1; }
    }
}

template <int P, int MODE>
__global__ __launch_bounds__(256, 1) void lem_encoder_ws1_kernel(LemWsArgs a) {
    constexpr int NS = (P + 1) / 2, M = (3 * P + 2 + 15) / 16;
    // y fragments [tile 2] | z fragments [tile 2] (16 KB each) | scaled biases [512 + 256]
    __shared__ __attribute__((aligned(16))) float lds[4 * SPLIT_CHUNK_FLOATS + 768];
    __shared__ __attribute__((aligned(16))) float xconst[64 * 8];
    __shared__ half8 wxl[4 * 4 * M * 64];               // input-column fragments: [wave 4][gate 4][m M][lane 64]
    half8* const yfr = reinterpret_cast<half8*>(lds);
    half8* const zfr = yfr + 2 * LEM_WS_FR;
    float* const bias_l = lds + 4 * SPLIT_CHUNK_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = (long)blockIdx.x * 64;
    const bool two = n0 + 32 < a.n_nodes;              // (uniform) the second tile holds nodes
    const float LOG2E = 1.44269504088896340736f;
    const float inv_w = a.scales[4], inv_z = a.scales[5];
    const int T = a.t_len;

    {   // prologue: y(0) = 0 for both tiles, biases, per-node constants
        half8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (_Float16)0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) yfr[tid + 256 * i] = zero;
        bias_l[tid] = a.bias_s[tid];
        bias_l[256 + tid] = a.bias_s[256 + tid];
        bias_l[512 + tid] = a.mlpb_s[tid];
        if (MODE != 0 && tid < 64) {
            const long nn = n0 + tid < a.n_nodes ? n0 + tid : a.n_nodes - 1;
            float* row = xconst + 8 * tid;
#pragma unroll
            for (int f = 0; f < 8; ++f) row[f] = 0.f;
            row[0] = a.pos_x[nn];
            if (MODE == 1) {
                for (int f = 0; f < a.nv; ++f) row[2 + f] = a.vars[(size_t)nn * a.nv + f];
            } else {
                row[3] = a.pos_t[nn];
                for (int f = 1; f < a.nv; ++f) row[3 + f] = a.vars[(size_t)nn * a.nv + f];
            }
        }
    }
    // stationary weights: the four gate blocks (rec_s order: g2, g3 | g1, lin) of this wave's 32 rows
    half8 w[4][4][2][2];
    half8* const wxw = wxl + (size_t)ks * (4 * M * 64) + lane;        // this wave's: [gate 4][m M][lane 64]
    {
        const half8* rs = reinterpret_cast<const half8*>(a.rec_s);
        const half8* wh = reinterpret_cast<const half8*>(a.wx_h);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[grp][kt][s][pl] = rs[(size_t)(grp * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
#pragma unroll
            for (int m = 0; m < M; ++m) wxw[(grp * M + m) * 64] = wh[((grp * 4 + ks) * 2 + m) * 64 + lane];
        }
    }
    const float c0 = -inv_w * LOG2E, c1a = -2.0f * inv_w * LOG2E, c1b = -2.0f * inv_z * LOG2E, idt = 1.0f / a.dt;

    f32x16 sz[2], sy[2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int r = 0; r < 16; ++r) { sz[X][r] = 0.f; sy[X][r] = 0.f; }
    // per-lane byte addresses: fragment areas as B operands (read) and this wave's K tile of them (published)
    constexpr int AREA = LEM_WS_FR * 16;
    const char* const fy0 = reinterpret_cast<const char*>(yfr) + lane * 16;
    const char* const fy1 = fy0 + AREA;
    const char* const fz0 = reinterpret_cast<const char*>(zfr) + lane * 16;
    const char* const fz1 = fz0 + AREA;
    char* const py0 = reinterpret_cast<char*>(yfr) + lane * 16 + ks * (4 * 64 * 16);
    char* const py1 = py0 + AREA;
    char* const pz0 = reinterpret_cast<char*>(zfr) + lane * 16 + ks * (4 * 64 * 16);
    char* const pz1 = pz0 + AREA;

    // Step inputs: x_X(t) is requested a phase before its B fragments are formed (the loads fly behind that phase's MFMAs).
    float xn0[2 * NS], xn1[2 * NS];
    half8 bx0[M], bx1[M];
    auto fetch0 = [&](int t) {
        const long n = n0 + c;
        lem_ws_load_x<P, MODE>(a, xconst + 8 * c, n < a.n_nodes ? n : a.n_nodes - 1, t, xn0);
    };
    auto fetch1 = [&](int t) {
        const long n = n0 + 32 + c;
        lem_ws_load_x<P, MODE>(a, xconst + 8 * (32 + c), n < a.n_nodes ? n : a.n_nodes - 1, t, xn1);
    };
    f32x16 a0, a1, b0, b1, c0a, c1acc, d0, d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = a1[r] = b0[r] = b1[r] = c0a[r] = c1acc[r] = d0[r] = d1[r] = 0.f; }
    __syncthreads();                    // the constants table and y(0) are read below
    fetch0(0);
    fetch1(0);
    lem_ws_slots<P>(xn0, hh, bx0);
    lem_ws_slots<P>(xn1, hh, bx1);

    const int role = 0;
    (void)role;
    LPROF_DECL
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
        // phase 0: M_A(0,t) || V_B(1,t-1)  (nothing to update before the first step)
        if (t == 0) lem_ws1_phase<true, false, M>(w[0], w[1], wxw, wxw + M * 64, bx0, fy0, fy0, a0, a1, d0, d1, c0, c1b, idt, sy[1], py1);
        else lem_ws1_phase<true, true, M>(w[0], w[1], wxw, wxw + M * 64, bx0, fy0, fy0, a0, a1, d0, d1, c0, c1b, idt, sy[1], py1);
        LPROF(lp_m);
        LEM_SYNC();
        // phase 1: M_A(1,t) || V_A(0,t)
        lem_ws1_phase<true, true, M>(w[0], w[1], wxw, wxw + M * 64, bx1, fy1, fy1, b0, b1, a0, a1, c0, c1a, idt, sz[0], pz0);
        LPROF(lp_m);
        LEM_SYNC();
        // phase 2: M_B(0,t) || V_A(1,t); x_0(t+1) requested, its fragments formed behind the phase (the last use of x_0(t) is this phase's)
        fetch0(tn);
        lem_ws1_phase<false, true, M>(w[2], w[3], wxw + 2 * M * 64, wxw + 3 * M * 64, bx0, fy0, fz0, c0a, c1acc, b0, b1, c0, c1a, idt, sz[1], pz1);
        LPROF(lp_m);
        lem_ws_slots<P>(xn0, hh, bx0);
        LPROF(lp_v);
        LEM_SYNC();
        // phase 3: M_B(1,t) || V_B(0,t); x_1(t+1) likewise
        fetch1(tn);
        lem_ws1_phase<false, true, M>(w[2], w[3], wxw + 2 * M * 64, wxw + 3 * M * 64, bx1, fy1, fz1, d0, d1, c0a, c1acc, c0, c1b, idt, sy[0], py0);
        LPROF(lp_m);
        lem_ws_slots<P>(xn1, hh, bx1);
        LPROF(lp_v);
        LEM_SYNC();
    }
    LPROF_FLUSH
    {   // V_B(1, T-1): the last update of tile 1
        half8 phi, plo;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            LemPairTmp tmp;
            lem_ws1_piece<0>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            lem_ws1_piece<1>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            lem_ws1_piece<2>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            lem_ws1_piece<3>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            lem_ws1_piece<4>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            lem_ws1_piece<5>(d0, d1, 2 * g, c0, c1b, idt, sy[1], tmp, phi, plo, 2 * (g & 3));
            if ((g & 3) == 3) {
                *reinterpret_cast<half8*>(py1 + ((g >> 2) * 2 + 0) * 1024) = phi;
                *reinterpret_cast<half8*>(py1 + ((g >> 2) * 2 + 1) * 1024) = plo;
            }
        }
    }
    __syncthreads();
    // here: y_X(T) of both tiles is published in yfr; every wave holds its y slices in sy[]

    if (a.with_mlp) {
        const half8* ms = reinterpret_cast<const half8*>(a.mlp_s);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        w[gi][kt][s][pl] = ms[(size_t)(gi * 4 + kt) * 1024 + ((s * 4 + ks) * 2 + pl) * 64 + lane];
        const float invA = a.scales[6], invB = a.scales[7];
        // lemoutput_mlp on both tiles: Swish(Wa y + ba) published into the (dead) z areas, then Swish(Wb . + bb)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const half8* yb = yfr + X * LEM_WS_FR;
            f32x16 acc;
            lem_ws_bias(bias_l + 512 + 32 * ks, hh, acc);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 h0 = yb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = yb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                    MSMP_MFMA_LOLO(2, acc, w[0][kt][s][1], l0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][1], h0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], l0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][kt][s][0], h0, acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = swishf(acc[r] * invA);
            lem_ws_publish(acc, zfr + X * LEM_WS_FR, ks, lane);
        }
        __syncthreads();
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const half8* hb = zfr + X * LEM_WS_FR;
            f32x16 res;
            lem_ws_bias(bias_l + 512 + H + 32 * ks, hh, res);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 h0 = hb[((kt * 2 + s) * 2 + 0) * 64 + lane], l0 = hb[((kt * 2 + s) * 2 + 1) * 64 + lane];
                    MSMP_MFMA_LOLO(2, res, w[1][kt][s][1], l0);
                    res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][1], h0, res, 0, 0, 0);
                    res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], l0, res, 0, 0, 0);
                    res = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][kt][s][0], h0, res, 0, 0, 0);
                }
            const long n = n0 + 32 * X + c;
            if (n < a.n_nodes) {
                float* o = a.out + (size_t)n * H + 32 * ks + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = swishf(res[4 * q + m] * invB);
                    *reinterpret_cast<f32x4*>(o + 8 * q) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const long n = n0 + 32 * X + c;
            if (n < a.n_nodes) {
                float* o = a.out + (size_t)n * H + 32 * ks + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v[m] = sy[X][4 * q + m];
                    *reinterpret_cast<f32x4*>(o + 8 * q) = v;
                }
            }
        }
    }
    (void)two;
}

// fp32 chunks [128 out][32 k] (row-major, as `rec`) -> bf16x3 A fragments, acc order: thread = (chunk, s, T, lane)
__global__ __launch_bounds__(256) void pack_lem_b3_kernel(const float* __restrict__ chunks, int n_chunks, float* __restrict__ out) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n_chunks * 512) return;
    const int lane = id & 63, T = (id >> 6) & 3, s = (id >> 8) & 1, ch = id >> 9;
    const int m = lane & 31, hh = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = chunks[(size_t)ch * CHUNK_FLOATS + (32 * T + m) * KC + split_k_acc(s, hh, j)];
    const Bf3 f = split_bf16x3(v);
    u32x4* dst = reinterpret_cast<u32x4*>(out) + (size_t)ch * LEM_B3_CHUNK_U4 + (size_t)((s * 4 + T) * 3) * 64 + lane;
    dst[0] = __builtin_bit_cast(u32x4, f.hi);
    dst[64] = __builtin_bit_cast(u32x4, f.mid);
    dst[128] = __builtin_bit_cast(u32x4, f.lo);
}

}  // namespace msmp

using namespace msmp;

#if MSMP_PROF_LEM
extern "C" __attribute__((visibility("default"))) int msmp_debug_prof_lem(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_lem), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof_lem), 16 * sizeof(unsigned long long));
}
#endif
int g_lem_share = 1;     // msmp_tune("lem_share", k): k launches of this kind share the GPU (sub-batches on k streams): a launch plans for CUs / k
int g_lem_tail = 1;      // msmp_tune("lem_tail", 0): every workgroup of the ws3 kernel takes three tiles (no round of one-tile workgroups)
// Partition of n_nodes into three-tile workgroups [0, full) and one-tile workgroups behind them (lem_encoder_ws3_kernel).  Cost
// model in rounds of one workgroup per CU: a three-tile workgroup 1, a one-tile workgroup 0.5 (measured at 2048 graphs: 203 vs 99 us per
// round at T = 25); the one-tile round is taken only when it saves at least 0.3 of a round (small launches measured slower with it).
static unsigned lem_partition_for(int64_t n_nodes, int cus, int tail, int* full_wgs) {
    const int64_t tiles = (n_nodes + 31) / 32, groups = (n_nodes + 95) / 96;
    *full_wgs = 0x7fffffff;
    if (!tail || cus < 1) return (unsigned)groups;
    const int64_t full = (groups / cus) * cus;               // whole rounds of three-tile workgroups
    const int64_t rem = tiles - 3 * full;                    // tiles left for the last round(s)
    if (rem <= 0) return (unsigned)groups;
    const double all_three = (double)((groups + cus - 1) / cus), mixed = (double)(full / cus) + 0.5 * (double)((rem + cus - 1) / cus);
    if (mixed > all_three - 0.3) return (unsigned)groups;
    *full_wgs = (int)full;
    return (unsigned)(full + rem);
}
static unsigned lem_partition(int64_t n_nodes, int* full_wgs) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cus = cus / (g_lem_share > 0 ? g_lem_share : 1);
    if (cus < 1) cus = 1;
    return lem_partition_for(n_nodes, cus, g_lem_tail, full_wgs);
}
// (tests, no GPU needed) the partition of n_nodes for a device of `cus` CUs: grid size and the number of three-tile workgroups
extern "C" __attribute__((visibility("default"))) int msmp_debug_lem_partition(int64_t n_nodes, int cus, int64_t* grid_out, int64_t* full_out) {
    int full = 0;
    const unsigned g = lem_partition_for(n_nodes, cus, 1, &full);
    if (grid_out) *grid_out = g;
    if (full_out) *full_out = full == 0x7fffffff ? (int64_t)g : full;
    return 0;
}
int g_lem_nodes = 1;     // msmp_tune("lem_nodes", 0): msmp_lem_encoder_nodes_f32 declines, callers assemble the [N,T,ninp] tensor (A/B)
int g_lem_split = 4;     // 4: weight-stationary anti-phased kernel (three node tiles), 3: weight-stationary two-tile kernel (round 2),
                         // 0: fp32 MFMA (msmp_tune "lem"; "split" 1/0 selects 4/0)
extern "C" int64_t msmp_packed_lem_floats(void) { return lem_layout().total; }

extern "C" int msmp_pack_lem_f32(const float* weights, const float* weights_lin_z, const float* bias, const float* bias_lin_z,
                                 const float* mlp_w0, const float* mlp_b0, const float* mlp_w1, const float* mlp_b1,
                                 int ninp, float* packed_out, msmp_stream_t stream) {
    MSMP_REQUIRE(weights && weights_lin_z && bias && bias_lin_z && packed_out, MSMP_ERR_ARG, "msmp_pack_lem_f32: null pointer");
    MSMP_REQUIRE(ninp >= 1 && ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_pack_lem_f32: ninp=%d outside 1..%d", ninp, LEM_MAX_INP);
    const bool mlp = mlp_w0 && mlp_b0 && mlp_w1 && mlp_b1;
    MSMP_REQUIRE(mlp || !(mlp_w0 || mlp_b0 || mlp_w1 || mlp_b1), MSMP_ERR_ARG, "msmp_pack_lem_f32: give all four mlp tensors or none");
    LemPackArgs a{weights, weights_lin_z, bias, bias_lin_z, mlp_w0, mlp_b0, mlp_w1, mlp_b1, ninp, packed_out};
    hipLaunchKernelGGL(pack_lem_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pack_lem_scale_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pack_lem_split_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pack_lem_b3_kernel, dim3(16 * 512 / 256), dim3(256), 0, (hipStream_t)stream, packed_out + lem_layout().rec, 16,
                       packed_out + lem_layout().rec_b3);
    return check_launch("pack_lem_kernel");
}

extern "C" int msmp_lem_input_stride(int ninp) { return ninp >= 1 && ninp <= LEM_MAX_INP ? 2 * ((ninp + 1) / 2) : -1; }

extern "C" int msmp_lem_encoder_f32(const float* xin, int64_t n_nodes, int t_len, int ninp, float dt, const float* packed,
                                    int with_mlp, float* h_out, msmp_stream_t stream) {
    MSMP_REQUIRE(xin && packed && h_out, MSMP_ERR_ARG, "msmp_lem_encoder_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && t_len >= 1, MSMP_ERR_ARG, "msmp_lem_encoder_f32: bad sizes");
    MSMP_REQUIRE(ninp >= 1 && ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_lem_encoder_f32: ninp=%d outside 1..%d", ninp, LEM_MAX_INP);
    const LemLayout L = lem_layout();
    LemArgs a{xin, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec, packed + L.mlp, packed + L.bias, packed + L.wx,
              packed + L.mlpb, h_out};
    const unsigned grid = (unsigned)((n_nodes + 127) / 128);
    timing_begin(MSMP_K_LEM, (hipStream_t)stream);
    if (g_lem_split == 5) {
        LemWsArgs wa{xin, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec_s,
                     packed + L.mlp_s, packed + L.bias_s, packed + L.wx_h, packed + L.mlpb_s, packed + L.scales, h_out};
        const unsigned g1 = (unsigned)((n_nodes + 63) / 64);
        hipStream_t st = (hipStream_t)stream;
        switch (ninp) {
            case 1: hipLaunchKernelGGL((lem_encoder_ws1_kernel<1, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 2: hipLaunchKernelGGL((lem_encoder_ws1_kernel<2, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 3: hipLaunchKernelGGL((lem_encoder_ws1_kernel<3, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 4: hipLaunchKernelGGL((lem_encoder_ws1_kernel<4, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws1_kernel<5, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws1_kernel<6, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws1_kernel<7, 0>), dim3(g1), dim3(256), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws1_kernel<8, 0>), dim3(g1), dim3(256), 0, st, wa); break;
        }
    } else if (g_lem_split == 4) {
        LemWsArgs wa{xin, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec_s,
                     packed + L.mlp_s, packed + L.bias_s, packed + L.wx_h, packed + L.mlpb_s, packed + L.scales, h_out};
        const unsigned g96 = lem_partition(n_nodes, &wa.full_wgs);
        hipStream_t st = (hipStream_t)stream;
        switch (ninp) {
            case 1: hipLaunchKernelGGL((lem_encoder_ws3_kernel<1, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 2: hipLaunchKernelGGL((lem_encoder_ws3_kernel<2, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 3: hipLaunchKernelGGL((lem_encoder_ws3_kernel<3, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 4: hipLaunchKernelGGL((lem_encoder_ws3_kernel<4, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws3_kernel<5, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws3_kernel<6, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws3_kernel<7, 0>), dim3(g96), dim3(512), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws3_kernel<8, 0>), dim3(g96), dim3(512), 0, st, wa); break;
        }
    } else if (g_lem_split == 3) {
        LemWsArgs wa{xin, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec_s,
                     packed + L.mlp_s, packed + L.bias_s, packed + L.wx_h, packed + L.mlpb_s, packed + L.scales, h_out};
        const unsigned g64 = (unsigned)((n_nodes + 63) / 64);
        hipStream_t st = (hipStream_t)stream;
        switch (ninp) {
            case 1: hipLaunchKernelGGL((lem_encoder_ws_kernel<1, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 2: hipLaunchKernelGGL((lem_encoder_ws_kernel<2, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 3: hipLaunchKernelGGL((lem_encoder_ws_kernel<3, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 4: hipLaunchKernelGGL((lem_encoder_ws_kernel<4, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws_kernel<5, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws_kernel<6, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws_kernel<7, 0>), dim3(g64), dim3(512), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws_kernel<8, 0>), dim3(g64), dim3(512), 0, st, wa); break;
        }
    } else
    switch ((ninp + 1) / 2) {
        case 1: hipLaunchKernelGGL(lem_encoder_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(lem_encoder_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(lem_encoder_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(lem_encoder_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
    }
    timing_end(MSMP_K_LEM, (hipStream_t)stream);
    return check_launch("lem_encoder_kernel");
}

// The same weight-stationary kernel with the step inputs assembled in the kernel from the node arrays (no [N, T, ninp]
// tensor in HBM): two_d = 0: x_t = [pos_x, u_t, variables] (experiments/models_gnn.py:1357-1360, ninp = 2 + nv);
// two_d = 1: x_t = [pos_x, u_t, u_{tw+t}, dt_cum_t + pos_t, variables[1:]] (models_gnn2D.py:429-433, ninp = 3 + nv).
extern "C" int msmp_lem_encoder_nodes_f32(const float* u, const float* pos_x, const float* pos_t, const float* vars, const float* dt_cum,
                                          int64_t n_nodes, int tw, int nv, int two_d, float dt, const float* packed, int with_mlp,
                                          float* h_out, msmp_stream_t stream) {
    MSMP_REQUIRE(u && pos_x && vars && packed && h_out, MSMP_ERR_ARG, "msmp_lem_encoder_nodes_f32: null pointer");
    MSMP_REQUIRE(!two_d || (pos_t && dt_cum), MSMP_ERR_ARG, "msmp_lem_encoder_nodes_f32: the 2-D input needs pos_t and dt_cum");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31) && tw >= 1 && nv >= 1, MSMP_ERR_ARG, "msmp_lem_encoder_nodes_f32: bad sizes");
    const int ninp = (two_d ? 3 : 2) + nv;
    MSMP_REQUIRE(ninp <= LEM_MAX_INP, MSMP_ERR_UNSUPPORTED, "msmp_lem_encoder_nodes_f32: ninp=%d > %d", ninp, LEM_MAX_INP);
    MSMP_REQUIRE((g_lem_split == 3 || g_lem_split == 4 || g_lem_split == 5) && g_lem_nodes, MSMP_ERR_UNSUPPORTED,
                 "msmp_lem_encoder_nodes_f32: only the weight-stationary editions (msmp_tune lem 3 / 4)");
    const LemLayout L = lem_layout();
    LemWsArgs wa{nullptr, u, pos_x, pos_t, vars, dt_cum, tw, nv, (long)n_nodes, tw, with_mlp, dt, packed + L.rec_s, packed + L.mlp_s,
                 packed + L.bias_s, packed + L.wx_h, packed + L.mlpb_s, packed + L.scales, h_out};
    const unsigned g64 = (unsigned)((n_nodes + 63) / 64), g96 = g_lem_split == 4 ? lem_partition(n_nodes, &wa.full_wgs) : 0u;
    hipStream_t st = (hipStream_t)stream;
    timing_begin(MSMP_K_LEM, st);
    if (g_lem_split == 5) {
        const unsigned g1 = (unsigned)((n_nodes + 63) / 64);
        if (!two_d) switch (ninp) {
            case 3: hipLaunchKernelGGL((lem_encoder_ws1_kernel<3, 1>), dim3(g1), dim3(256), 0, st, wa); break;
            case 4: hipLaunchKernelGGL((lem_encoder_ws1_kernel<4, 1>), dim3(g1), dim3(256), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws1_kernel<5, 1>), dim3(g1), dim3(256), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws1_kernel<6, 1>), dim3(g1), dim3(256), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws1_kernel<7, 1>), dim3(g1), dim3(256), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws1_kernel<8, 1>), dim3(g1), dim3(256), 0, st, wa); break;
        } else switch (ninp) {
            case 4: hipLaunchKernelGGL((lem_encoder_ws1_kernel<4, 2>), dim3(g1), dim3(256), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws1_kernel<5, 2>), dim3(g1), dim3(256), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws1_kernel<6, 2>), dim3(g1), dim3(256), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws1_kernel<7, 2>), dim3(g1), dim3(256), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws1_kernel<8, 2>), dim3(g1), dim3(256), 0, st, wa); break;
        }
    } else
    if (g_lem_split == 4) {
        if (!two_d) switch (ninp) {
            case 3: hipLaunchKernelGGL((lem_encoder_ws3_kernel<3, 1>), dim3(g96), dim3(512), 0, st, wa); break;
            case 4: hipLaunchKernelGGL((lem_encoder_ws3_kernel<4, 1>), dim3(g96), dim3(512), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws3_kernel<5, 1>), dim3(g96), dim3(512), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws3_kernel<6, 1>), dim3(g96), dim3(512), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws3_kernel<7, 1>), dim3(g96), dim3(512), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws3_kernel<8, 1>), dim3(g96), dim3(512), 0, st, wa); break;
        } else switch (ninp) {
            case 4: hipLaunchKernelGGL((lem_encoder_ws3_kernel<4, 2>), dim3(g96), dim3(512), 0, st, wa); break;
            case 5: hipLaunchKernelGGL((lem_encoder_ws3_kernel<5, 2>), dim3(g96), dim3(512), 0, st, wa); break;
            case 6: hipLaunchKernelGGL((lem_encoder_ws3_kernel<6, 2>), dim3(g96), dim3(512), 0, st, wa); break;
            case 7: hipLaunchKernelGGL((lem_encoder_ws3_kernel<7, 2>), dim3(g96), dim3(512), 0, st, wa); break;
            default: hipLaunchKernelGGL((lem_encoder_ws3_kernel<8, 2>), dim3(g96), dim3(512), 0, st, wa); break;
        }
    } else
    if (!two_d) switch (ninp) {
        case 3: hipLaunchKernelGGL((lem_encoder_ws_kernel<3, 1>), dim3(g64), dim3(512), 0, st, wa); break;
        case 4: hipLaunchKernelGGL((lem_encoder_ws_kernel<4, 1>), dim3(g64), dim3(512), 0, st, wa); break;
        case 5: hipLaunchKernelGGL((lem_encoder_ws_kernel<5, 1>), dim3(g64), dim3(512), 0, st, wa); break;
        case 6: hipLaunchKernelGGL((lem_encoder_ws_kernel<6, 1>), dim3(g64), dim3(512), 0, st, wa); break;
        case 7: hipLaunchKernelGGL((lem_encoder_ws_kernel<7, 1>), dim3(g64), dim3(512), 0, st, wa); break;
        default: hipLaunchKernelGGL((lem_encoder_ws_kernel<8, 1>), dim3(g64), dim3(512), 0, st, wa); break;
    } else switch (ninp) {
        case 4: hipLaunchKernelGGL((lem_encoder_ws_kernel<4, 2>), dim3(g64), dim3(512), 0, st, wa); break;
        case 5: hipLaunchKernelGGL((lem_encoder_ws_kernel<5, 2>), dim3(g64), dim3(512), 0, st, wa); break;
        case 6: hipLaunchKernelGGL((lem_encoder_ws_kernel<6, 2>), dim3(g64), dim3(512), 0, st, wa); break;
        case 7: hipLaunchKernelGGL((lem_encoder_ws_kernel<7, 2>), dim3(g64), dim3(512), 0, st, wa); break;
        default: hipLaunchKernelGGL((lem_encoder_ws_kernel<8, 2>), dim3(g64), dim3(512), 0, st, wa); break;
    }
    timing_end(MSMP_K_LEM, st);
    return check_launch("lem_encoder_ws_kernel");
}
