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
struct LemLayout {
    int64_t rec, mlp, bias, wx, mlpb, total;
};

__host__ __device__ inline LemLayout lem_layout() {
    LemLayout L;
    int64_t o = 0;
    L.rec = o; o += 16 * CHUNK_FLOATS;
    L.mlp = o; o += 8 * CHUNK_FLOATS;
    L.bias = o; o += 4 * H;
    L.wx = o; o += 4 * H * LEM_MAX_INP;
    L.mlpb = o; o += 2 * H;
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
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < L.total; p += (int64_t)gridDim.x * blockDim.x) {
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

}  // namespace msmp

using namespace msmp;

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
    switch ((ninp + 1) / 2) {
        case 1: hipLaunchKernelGGL(lem_encoder_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(lem_encoder_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(lem_encoder_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(lem_encoder_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
    }
    timing_end(MSMP_K_LEM, (hipStream_t)stream);
    return check_launch("lem_encoder_kernel");
}
