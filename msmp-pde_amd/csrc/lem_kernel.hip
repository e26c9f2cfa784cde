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
#include "mfma_tiles.h"

namespace msmp {

constexpr int LEM_MAX_INP = 8;

// packed LEM blob (floats):  rec (16 chunks: g2 x4, g3 x4, g1 x4, lin x4) | mlp (8 chunks: Wa x4, Wb x4) |
//   bias [512] (g1, g2, g3, bz in the reference's row order) |
//   wxf [4 groups: g1,g2,g3,lin][4 T][4 s][64 lanes]: the input columns as MFMA A fragments, value
//        = W[128*group + 32T + (lane & 31)][H + 2s + (lane >> 5)] (0 past ninp) |
//   mlp bias [256] (ba, bb)
//   fp16-split copies for the split kernel (mfma_tiles.h): scales [8] (2^s of W, Wz, Wa, Wb, then 2^-s) |
//   rec_s (16 split chunks, acc order, same consumption order) | mlp_s (8 split chunks) |
//   bias_s [512], wxf_s [4096], mlpb_s [256]: the fp32 bias / input-column fragments pre-multiplied by 2^s of their matrix
struct LemLayout {
    int64_t rec, mlp, bias, wx, mlpb, scales, rec_s, mlp_s, bias_s, wx_s, mlpb_s, total;
};

__host__ __device__ inline LemLayout lem_layout() {
    LemLayout L;
    int64_t o = 0;
    L.rec = o; o += 16 * CHUNK_FLOATS;
    L.mlp = o; o += 8 * CHUNK_FLOATS;
    L.bias = o; o += 4 * H;
    L.wx = o; o += 4 * H * LEM_MAX_INP;
    L.mlpb = o; o += 2 * H;
    L.scales = o; o += 8;
    L.rec_s = o; o += 16 * CHUNK_FLOATS;
    L.mlp_s = o; o += 8 * CHUNK_FLOATS;
    L.bias_s = o; o += 4 * H;
    L.wx_s = o; o += 4 * H * LEM_MAX_INP;
    L.mlpb_s = o; o += 2 * H;
    L.total = o;
    return L;
}

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
}

__device__ __forceinline__ float tanhf_(float x) {
    // (1 - e^{-2|x|}) / (1 + e^{-2|x|}) with the sign restored; absolute error ~1e-7
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.88539008177792681472f);
    const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return copysignf(r, x);
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
// fp16-split edition (mfma_tiles.h): same recurrence, the four K = 128 GEMMs per step on the fp16 matrix pipe.
// The states stay fp32 in accumulator layout; y is split into hi/lo fragments once per step (shared by the three
// gate GEMMs), z once (for the lin GEMM, in the same registers).  Input columns remain NS fp32 MFMA k-steps, with
// bias and fragments pre-multiplied by the matrix's 2^s so that they accumulate into the same scaled accumulator.
// ----------------------------------------------------------------------------------------------
struct LemSplitArgs {
    LemArgs b;            // rec / mlp point at the SPLIT chunks, bias / wx / mlpb at the pre-scaled copies
    const float* scales;  // [8]
};

template <int NS>
__device__ __forceinline__ void lem_acc_init_s(const LemArgs& a, int grp, int lane, int hh, const float (&x)[2 * NS],
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
__global__ __launch_bounds__(256) void lem_encoder_split_kernel(LemSplitArgs sa) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS];
    const LemArgs& a = sa.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const float* xrow = a.xin + (size_t)nc * a.t_len * (2 * NS);
    const float invW = sa.scales[4], invZ = sa.scales[5];

    f32x16 y[4][1], z[4][1], g[4][1], acc[4][1];
    half8 sh[4][1][2], sl[4][1][2];            // hi / lo fragments of the state currently used as B operand
    acc_zero<1>(y);
    acc_zero<1>(z);
#pragma unroll
    for (int T = 0; T < 4; ++T) split_acc_tile<1>(y[T], sh[T], sl[T]);

    WStage ws;
    wstage_load(ws, a.rec, tid);
    wstage_store_linear(ws, lds, tid);
    __syncthreads();

#define LEM_GROUP_S(ACC, BASE, NEXT_AFTER_LAST)                                                        \
    _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) {                                                 \
        const float* nxt = kc < 3 ? a.rec + (size_t)((BASE) + kc + 1) * SPLIT_CHUNK_FLOATS : (NEXT_AFTER_LAST); \
        wstage_load(ws, nxt, tid);                                                                     \
        mma_chunk_split<1>(lds + (((BASE) + kc) & 1) * SPLIT_CHUNK_FLOATS, lane, sh[kc], sl[kc], ACC);  \
        wstage_store_linear(ws, lds + (((BASE) + kc + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);              \
        __syncthreads();                                                                               \
    }

    for (int t = 0; t < a.t_len; ++t) {
        float x[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) x[f] = xrow[t * (2 * NS) + f];
        const float* after = t + 1 == a.t_len ? a.mlp : a.rec;

        lem_acc_init_s<NS>(a, 1, lane, hh, x, g);                    // g2 -> dt_
        LEM_GROUP_S(g, 0, a.rec + 4 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[T][0][r] = a.dt * sigmoidf_(g[T][0][r] * invW);
        lem_acc_init_s<NS>(a, 2, lane, hh, x, acc);                  // g3 -> z update
        LEM_GROUP_S(acc, 4, a.rec + 8 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                z[T][0][r] = (1.0f - g[T][0][r]) * z[T][0][r] + g[T][0][r] * tanhf_(acc[T][0][r] * invW);
        lem_acc_init_s<NS>(a, 0, lane, hh, x, g);                    // g1 -> dt_bar (still from the y fragments)
        LEM_GROUP_S(g, 8, a.rec + 12 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[T][0][r] = a.dt * sigmoidf_(g[T][0][r] * invW);
#pragma unroll
        for (int T = 0; T < 4; ++T) split_acc_tile<1>(z[T], sh[T], sl[T]);      // fragments now hold the new z
        lem_acc_init_s<NS>(a, 3, lane, hh, x, acc);                  // lin -> y update
        LEM_GROUP_S(acc, 12, after)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                y[T][0][r] = (1.0f - g[T][0][r]) * y[T][0][r] + g[T][0][r] * tanhf_(acc[T][0][r] * invZ);
#pragma unroll
        for (int T = 0; T < 4; ++T) split_acc_tile<1>(y[T], sh[T], sl[T]);      // fragments of the new y
    }
#undef LEM_GROUP_S

    if (a.with_mlp) {
        const float invA = sa.scales[6], invB = sa.scales[7];
        acc_init_bias<1>(a.mlpb, hh, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wstage_load(ws, a.mlp + (size_t)(j + 1) * SPLIT_CHUNK_FLOATS, tid);
            mma_chunk_split<1>(lds + (j & 1) * SPLIT_CHUNK_FLOATS, lane, sh[j], sl[j], acc);
            wstage_store_linear(ws, lds + ((j + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
            __syncthreads();
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[T][0][r] = swishf(acc[T][0][r] * invA);
            split_acc_tile<1>(acc[T], sh[T], sl[T]);
        }
        acc_init_bias<1>(a.mlpb + H, hh, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) wstage_load(ws, a.mlp + (size_t)(4 + j + 1) * SPLIT_CHUNK_FLOATS, tid);
            mma_chunk_split<1>(lds + (j & 1) * SPLIT_CHUNK_FLOATS, lane, sh[j], sl[j], y);
            if (j < 3) {
                wstage_store_linear(ws, lds + ((j + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
                __syncthreads();
            }
        }
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[T][0][r] = swishf(y[T][0][r] * invB);
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
// Two-waves-per-SIMD edition of the split kernel.  A 512-thread workgroup = 4 node blocks x 2 CHANNEL HALVES:
// wave (nbk = wave & 3, chh = wave >> 2) owns the output tiles {2 chh, 2 chh + 1} (64 of the 128 channels) of
// node block nbk for every GEMM, so it keeps only half of each state / accumulator (y, z, g, acc: 4 x 32
// registers) and fits 256 registers: two waves share a SIMD and one's activation VALU overlaps the other's
// matrix work.  A GEMM needs all 128 k (channels) of the state as B operand: after every state update each wave
// splits its two tiles into hi/lo fragments and publishes them in LDS ([node block][tile 4][step 2][plane 2][lane 64]
// half8 = 16 KB per node block, one area for y and one for z); the GEMMs read their B fragments from there.
// ----------------------------------------------------------------------------------------------
struct WStage2 {
    f32x4 r[2];
};
__device__ __forceinline__ void wstage2_load(WStage2& s, const float* __restrict__ chunk, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) s.r[i] = *reinterpret_cast<const f32x4*>(chunk + 4 * (tid + 512 * i));
}
__device__ __forceinline__ void wstage2_store(const WStage2& s, float* buf, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(buf + 4 * (tid + 512 * i)) = s.r[i];
}

// acc[i] += W_chunk[rows of tile tile0 + i] * B for the 32 k of one split chunk (two of the four row tiles)
__device__ __forceinline__ void mma_chunk_split_half(const float* wl, int lane, int tile0, const half8 (&bhi)[2],
                                                     const half8 (&blo)[2], f32x16 (&acc)[2]) {
    const half8* w = reinterpret_cast<const half8*>(wl) + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const half8 ahi = w[((s * 4 + tile0 + i) * 2 + 0) * 64];
            const half8 alo = w[((s * 4 + tile0 + i) * 2 + 1) * 64];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[s], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[s], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[s], acc[i], 0, 0, 0);
        }
}

// acc[i] = bias[row0 + 32 i + ..] (+ input columns as NS fp32 MFMA k-steps from the pre-scaled fragments)
template <int NS>
__device__ __forceinline__ void lem_acc_init_half(const float* bias_rows, const float* wf_tile0, int hh, const float (&x)[2 * NS],
                                                  bool with_inputs, f32x16 (&acc)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_rows + 32 * i + 8 * q + 4 * hh);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[i][4 * q + m] = bv[m];
        }
    if (with_inputs) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float b = hh ? x[2 * s + 1] : x[2 * s];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf_tile0[(i * 4 + s) * 64], b, acc[i], 0, 0, 0);
        }
    }
}

// split this wave's two state tiles into hi/lo fragments and publish them in the node block's exchange area
__device__ __forceinline__ void lem_publish(const f32x16 (&st)[2], half8* xb, int tile0, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = st[i][8 * s + j];
            half8 hi, lo;
            split8(v, hi, lo);
            xb[(((tile0 + i) * 2 + s) * 2 + 0) * 64 + lane] = hi;
            xb[(((tile0 + i) * 2 + s) * 2 + 1) * 64 + lane] = lo;
        }
    __syncthreads();
}

// B fragments of K tile kc (all 128 channels are published by the two waves of the node block)
__device__ __forceinline__ void lem_frags(const half8* xb, int kc, int lane, half8 (&bhi)[2], half8 (&blo)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bhi[s] = xb[((kc * 2 + s) * 2 + 0) * 64 + lane];
        blo[s] = xb[((kc * 2 + s) * 2 + 1) * 64 + lane];
    }
}

template <int NS>
__global__ __launch_bounds__(512, 2) void lem_encoder_split2_kernel(LemSplitArgs sa) {
    // weights 2 x 16 KB | y fragments 4 node blocks x 16 KB | z fragments 4 x 16 KB = 160 KB: one workgroup per CU
    __shared__ __attribute__((aligned(16))) float lds[10 * SPLIT_CHUNK_FLOATS];
    const LemArgs& a = sa.b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbk = wave & 3, chh = wave >> 2, tile0 = 2 * chh;
    const int c = lane & 31, hh = lane >> 5;
    const long n = (long)blockIdx.x * 128 + nbk * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const float* xrow = a.xin + (size_t)nc * a.t_len * (2 * NS);
    half8* xy = reinterpret_cast<half8*>(lds + 2 * SPLIT_CHUNK_FLOATS) + nbk * 1024;
    half8* xz = reinterpret_cast<half8*>(lds + 6 * SPLIT_CHUNK_FLOATS) + nbk * 1024;
    const float LOG2E = 1.44269504088896340736f;
    const float cW = -sa.scales[4] * LOG2E, cW2 = -2.0f * sa.scales[4] * LOG2E, cZ2 = -2.0f * sa.scales[5] * LOG2E;

    f32x16 y[2], z[2], g[2], acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { y[i][r] = 0.f; z[i][r] = 0.f; }

    WStage2 ws;
    wstage2_load(ws, a.rec, tid);
    wstage2_store(ws, lds, tid);
    lem_publish(y, xy, tile0, lane);           // y = 0 fragments (+ the barrier that also covers the weight chunk)

#define LEM_GROUP_H(ACC, BASE, XB, NEXT_AFTER_LAST)                                                    \
    _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) {                                                 \
        const float* nxt = kc < 3 ? a.rec + (size_t)((BASE) + kc + 1) * SPLIT_CHUNK_FLOATS : (NEXT_AFTER_LAST); \
        wstage2_load(ws, nxt, tid);                                                                    \
        half8 bhi[2], blo[2];                                                                          \
        lem_frags(XB, kc, lane, bhi, blo);                                                             \
        mma_chunk_split_half(lds + (((BASE) + kc) & 1) * SPLIT_CHUNK_FLOATS, lane, tile0, bhi, blo, ACC); \
        wstage2_store(ws, lds + (((BASE) + kc + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);                    \
        __syncthreads();                                                                               \
    }
    // sigmoid(x 2^-s) = 1 / (1 + 2^(c x)),  tanh(x 2^-s) = sign(x) (1 - e) / (1 + e), e = 2^(c2 |x|)
    auto sig = [](float x, float cc) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * cc)); };
    auto tnh = [](float x, float c2) {
        const float e = __builtin_amdgcn_exp2f(fabsf(x) * c2);
        return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
    };
    const float* wfb = a.wx + lane + (size_t)tile0 * 4 * 64;

    for (int t = 0; t < a.t_len; ++t) {
        float x[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) x[f] = xrow[t * (2 * NS) + f];
        const float* after = t + 1 == a.t_len ? a.mlp : a.rec;

        lem_acc_init_half<NS>(a.bias + H * 1 + 32 * tile0, wfb + 1 * 1024, hh, x, true, g);      // g2 -> dt_
        LEM_GROUP_H(g, 0, xy, a.rec + 4 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[i][r] = a.dt * sig(g[i][r], cW);
        lem_acc_init_half<NS>(a.bias + H * 2 + 32 * tile0, wfb + 2 * 1024, hh, x, true, acc);    // g3 -> z
        LEM_GROUP_H(acc, 4, xy, a.rec + 8 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) z[i][r] = fmaf(g[i][r], tnh(acc[i][r], cW2) - z[i][r], z[i][r]);
        lem_publish(z, xz, tile0, lane);                                                         // z fragments (lin reads them)
        lem_acc_init_half<NS>(a.bias + H * 0 + 32 * tile0, wfb + 0 * 1024, hh, x, true, g);      // g1 -> dt_bar
        LEM_GROUP_H(g, 8, xy, a.rec + 12 * SPLIT_CHUNK_FLOATS)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[i][r] = a.dt * sig(g[i][r], cW);
        lem_acc_init_half<NS>(a.bias + H * 3 + 32 * tile0, wfb + 3 * 1024, hh, x, true, acc);    // lin -> y
        LEM_GROUP_H(acc, 12, xz, after)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[i][r] = fmaf(g[i][r], tnh(acc[i][r], cZ2) - y[i][r], y[i][r]);
        lem_publish(y, xy, tile0, lane);                                                         // y fragments for the next step
    }

    if (a.with_mlp) {
        const float invA = sa.scales[6], invB = sa.scales[7];
        float dummy[2 * NS];
#pragma unroll
        for (int f = 0; f < 2 * NS; ++f) dummy[f] = 0.f;
        lem_acc_init_half<NS>(a.mlpb + 32 * tile0, nullptr, hh, dummy, false, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wstage2_load(ws, a.mlp + (size_t)(j + 1) * SPLIT_CHUNK_FLOATS, tid);
            half8 bhi[2], blo[2];
            lem_frags(xy, j, lane, bhi, blo);
            mma_chunk_split_half(lds + (j & 1) * SPLIT_CHUNK_FLOATS, lane, tile0, bhi, blo, acc);
            wstage2_store(ws, lds + ((j + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = swishf(acc[i][r] * invA);
        lem_publish(acc, xz, tile0, lane);
        lem_acc_init_half<NS>(a.mlpb + H + 32 * tile0, nullptr, hh, dummy, false, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) wstage2_load(ws, a.mlp + (size_t)(4 + j + 1) * SPLIT_CHUNK_FLOATS, tid);
            half8 bhi[2], blo[2];
            lem_frags(xz, j, lane, bhi, blo);
            mma_chunk_split_half(lds + (j & 1) * SPLIT_CHUNK_FLOATS, lane, tile0, bhi, blo, y);
            if (j < 3) {
                wstage2_store(ws, lds + ((j + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[i][r] = swishf(y[i][r] * invB);
    }
#undef LEM_GROUP_H

    if (n < a.n_nodes) {
        float* o = a.out + (size_t)n * H + 32 * tile0 + 4 * hh;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = y[i][4 * q + m];
                *reinterpret_cast<f32x4*>(o + 32 * i + 8 * q) = v;
            }
    }
}

}  // namespace msmp

using namespace msmp;

int g_lem_split = 1;     // 1: two-waves-per-SIMD split kernel, 2: one-wave split kernel, 0: fp32 MFMA (msmp_tune "split" / "lem")
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
    if (g_lem_split == 1) {
        LemSplitArgs sa{LemArgs{xin, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec_s, packed + L.mlp_s, packed + L.bias_s,
                                packed + L.wx_s, packed + L.mlpb_s, h_out},
                        packed + L.scales};
        switch ((ninp + 1) / 2) {
            case 1: hipLaunchKernelGGL(lem_encoder_split2_kernel<1>, dim3(grid), dim3(512), 0, (hipStream_t)stream, sa); break;
            case 2: hipLaunchKernelGGL(lem_encoder_split2_kernel<2>, dim3(grid), dim3(512), 0, (hipStream_t)stream, sa); break;
            case 3: hipLaunchKernelGGL(lem_encoder_split2_kernel<3>, dim3(grid), dim3(512), 0, (hipStream_t)stream, sa); break;
            default: hipLaunchKernelGGL(lem_encoder_split2_kernel<4>, dim3(grid), dim3(512), 0, (hipStream_t)stream, sa); break;
        }
    } else if (g_lem_split) {
        LemSplitArgs sa{LemArgs{xin, (long)n_nodes, t_len, with_mlp, dt, packed + L.rec_s, packed + L.mlp_s, packed + L.bias_s,
                                packed + L.wx_s, packed + L.mlpb_s, h_out},
                        packed + L.scales};
        switch ((ninp + 1) / 2) {
            case 1: hipLaunchKernelGGL(lem_encoder_split_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa); break;
            case 2: hipLaunchKernelGGL(lem_encoder_split_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa); break;
            case 3: hipLaunchKernelGGL(lem_encoder_split_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa); break;
            default: hipLaunchKernelGGL(lem_encoder_split_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sa); break;
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
