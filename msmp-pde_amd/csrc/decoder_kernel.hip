// Decoder of the 1-D solver classes fused into one pass:
//   diff = Conv1d(8 -> 1, k2)( Swish( Conv1d(1 -> 8, k1, stride s1)( h[:, None, :] ) ) )      experiments/models_gnn.py:210-224, 278
//   out  = u[:, -1:] + cumsum(dt)[None, :] * diff                                              :275-279
// One node per lane, the 128-channel row h[n] in registers; the eight intermediate channels are produced
// one at a time (L1 = (128-k1)/s1+1 values), passed through Swish and folded into the tw outputs at once,
// so nothing but h (read) and out (write) touches memory.  Weights are indexed with compile-time
// constants: the compiler keeps them in SGPRs (scalar loads).  ~7.7 kFMA per node: VALU work, ~3 GFLOP per
// E2-2048 batch; HBM traffic N*(512 + 4 + 100) bytes = 126 MB.
#include "msmp_common.h"
#include "decoder_body.h"

namespace msmp {

struct DecArgs {
    const float* h;      // [N,128]
    const float* u;      // [N,tw]
    long n_nodes;
    const float* w1;     // [8][k1]
    const float* b1;     // [8]
    const float* w2;     // [8][k2]
    const float* b2;     // [1]
    float dt;
    float* out;          // [N,tw]
};

template <int TW, int K1, int S1, int K2>
__global__ __launch_bounds__(256) void decoder_kernel(DecArgs a) {
    constexpr int L1 = (H - K1) / S1 + 1;
    static_assert(L1 - K2 + 1 == TW, "decoder geometry");
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= a.n_nodes) return;
    float x[H];
    const f32x4* hp = reinterpret_cast<const f32x4*>(a.h + (size_t)n * H);
#pragma unroll
    for (int i = 0; i < H / 4; ++i) {
        const f32x4 v = hp[i];
        x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
    }
    float o[TW];
    const float bias2 = a.b2[0];
#pragma unroll
    for (int t = 0; t < TW; ++t) o[t] = bias2;
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        float w1c[K1], w2c[K2];
#pragma unroll
        for (int j = 0; j < K1; ++j) w1c[j] = a.w1[c * K1 + j];
#pragma unroll
        for (int j = 0; j < K2; ++j) w2c[j] = a.w2[c * K2 + j];
        const float bc = a.b1[c];
        float mid[L1];
#pragma unroll
        for (int p = 0; p < L1; ++p) {
            float s = bc;
#pragma unroll
            for (int j = 0; j < K1; ++j) s = fmaf(w1c[j], x[p * S1 + j], s);
            mid[p] = swishf(s);
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            float s = o[t];
#pragma unroll
            for (int j = 0; j < K2; ++j) s = fmaf(w2c[j], mid[t + j], s);
            o[t] = s;
        }
    }
    float* op = a.out + (size_t)n * TW;
    if (a.u == nullptr) {                   // the decoder output alone (MSSMP_PDE_Solver_sub, models_gnn.py:1679-1682)
#pragma unroll
        for (int t = 0; t < TW; ++t) op[t] = o[t];
        return;
    }
    const float ul = a.u[(size_t)n * TW + TW - 1];
    float tcum = 0.f;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        tcum += a.dt;                       // cumsum of a constant, float32 like torch.cumsum on the device
        op[t] = ul + tcum * o[t];
    }
}

// ----------------------------------------------------------------------------------------------
// The same decoder with EIGHT lanes per node, split by position (default since round 3).  decoder_kernel above keeps a node's
// whole row and both intermediate arrays in one lane's registers: 256 VGPRs + 250 AGPRs, ONE wave per SIMD, which issues at most one
// vector instruction per 4.4 clocks (scripts/micro/valu_issue.hip) on a SIMD that takes two -- 112 us per launch at 2048 graphs for
// 8.9 k instructions per wave, in 3.125 -> 4 rounds.  Here lane q of a node computes the intermediate positions
// [q PP, (q + 1) PP) of all eight channels from a 26-28-value window of the row, the node's 8 x L1 intermediates meet in LDS,
// and lane q then forms OPL consecutive outputs from a (OPL + K2 - 1)-value window per channel.  < 100 registers, three
// workgroups per CU.  The taps of every sum are added in the order of decoder_kernel: the same bits.
// ----------------------------------------------------------------------------------------------
template <int TW, int K1, int S1, int K2>
__global__ __launch_bounds__(256) void decoder_split_kernel(DecArgs a) {
    using G = DecSplit<TW, K1, S1, K2>;
    constexpr int LP = G::LP, NODES = G::NODES;
    __shared__ __attribute__((aligned(16))) float mid[NODES * 8 * LP];
    const int q = threadIdx.x & 7, nl = threadIdx.x >> 3;
    const long n = (long)blockIdx.x * NODES + nl;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const DecW w{a.w1, a.b1, a.w2, a.b2, a.u, a.dt, a.out};
    decoder_split_node<TW, K1, S1, K2, false>(a.h + (size_t)nc * H, mid + (size_t)nl * 8 * LP, q, n < a.n_nodes, n, w, [] { __syncthreads(); });
}

// ----------------------------------------------------------------------------------------------
// Decoder of the *2D solver classes (two solution components), experiments/models_gnn2D.py:79-88, 125-141:
//   diff = Conv1d(8 -> 2, k2)( Swish( Conv1d(2 -> 8, k1, stride s1)( hd ) ) ),   hd = double_mlp(h)  [N, 2, 128]
//   out  = unflatten(u) + cumsum(dt) * diff, flattened back to [N, 2*tw]
// Eight lanes per node, lane c = intermediate channel c: it builds mid[c][:] from both input rows (the eight
// lanes of a node read the same two 512-B rows: one L1 line fetch serves them), applies Swish, forms its
// contribution to the 2*tw outputs, and the eight contributions are summed with three xor-shuffles.
// ----------------------------------------------------------------------------------------------
struct Dec2Args {
    const float* hd;     // [N, 2, 128]
    const float* u;      // [N, 2*tw]
    long n_nodes;
    const float* w1;     // [8][2][k1]
    const float* b1;     // [8]
    const float* w2;     // [2][8][k2]
    const float* b2;     // [2]
    float dt;
    float* out;          // [N, 2*tw]
};

template <int TW, int K1, int S1, int K2>
__global__ __launch_bounds__(256) void decoder2d_kernel(Dec2Args a) {
    constexpr int L1 = (H - K1) / S1 + 1;
    static_assert(L1 - K2 + 1 == TW, "decoder geometry");
    const int c = threadIdx.x & 7;
    const long n = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    float mid[L1];
    const float bc = a.b1[c];
#pragma unroll
    for (int p = 0; p < L1; ++p) mid[p] = bc;
#pragma unroll 1
    for (int ci = 0; ci < 2; ++ci) {
        float x[H];
        const f32x4* hp = reinterpret_cast<const f32x4*>(a.hd + ((size_t)nc * 2 + ci) * H);
#pragma unroll
        for (int i = 0; i < H / 4; ++i) {
            const f32x4 v = hp[i];
            x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
        }
        float w[K1];
#pragma unroll
        for (int j = 0; j < K1; ++j) w[j] = a.w1[(c * 2 + ci) * K1 + j];
#pragma unroll
        for (int p = 0; p < L1; ++p) {
            float s = mid[p];
#pragma unroll
            for (int j = 0; j < K1; ++j) s = fmaf(w[j], x[p * S1 + j], s);
            mid[p] = s;
        }
    }
#pragma unroll
    for (int p = 0; p < L1; ++p) mid[p] = swishf(mid[p]);
    float o[2 * TW];
#pragma unroll
    for (int co = 0; co < 2; ++co) {
        float w[K2];
#pragma unroll
        for (int j = 0; j < K2; ++j) w[j] = a.w2[(co * 8 + c) * K2 + j];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < K2; ++j) s = fmaf(w[j], mid[t + j], s);
            o[co * TW + t] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * TW; ++i) {
        float v = o[i];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        o[i] = v;
    }
    if (n >= a.n_nodes) return;
    const float b20 = a.b2[0], b21 = a.b2[1];
    float tcum = 0.f;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        tcum += a.dt;
        // lane c writes the outputs whose flat index i = co*TW + t satisfies i % 8 == c
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            const int i = co * TW + t;
            if ((i & 7) == c) a.out[(size_t)n * 2 * TW + i] = a.u[(size_t)n * 2 * TW + i] + tcum * (o[i] + (co ? b21 : b20));
        }
    }
}

// ----------------------------------------------------------------------------------------------
// The *2D decoder split by position like decoder_split_kernel (eight lanes per node; decoder2d_kernel keeps a 128-value input row,
// 38-59 intermediates and 2 tw partial outputs per lane).  Lane q builds the intermediate positions [q PP, (q + 1) PP) of all eight
// channels from windows of BOTH input rows, the node's 8 x L1 intermediates meet in LDS, lane q forms OPL consecutive outputs of both
// components.  Sums are formed exactly as decoder2d_kernel forms them -- per channel over the taps, the eight channel sums in the
// order of its xor-shuffle tree, then the bias -- the same bits.
// ----------------------------------------------------------------------------------------------
template <int TW, int K1, int S1, int K2>
__global__ __launch_bounds__(256) void decoder2d_split_kernel(Dec2Args a) {
    using G = DecSplit<TW, K1, S1, K2>;
    constexpr int L1 = G::L1, PP = G::PP, XW = G::XW, OPL = G::OPL, MW4 = G::MW4, LP = G::LP, NODES = G::NODES;
    __shared__ __attribute__((aligned(16))) float mid[NODES * 8 * LP];
    const int q = threadIdx.x & 7, nl = threadIdx.x >> 3;
    const long n = (long)blockIdx.x * NODES + nl;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const int p0 = q * PP;
    float x[2][XW];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const float* row = a.hd + ((size_t)nc * 2 + ci) * H;
        const int x0 = p0 * S1;
#pragma unroll
        for (int i = 0; i < XW; ++i) x[ci][i] = row[x0 + i < H ? x0 + i : H - 1];
    }
    float* mrow = mid + (size_t)nl * 8 * LP;
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        const float bc = a.b1[c];
        float s[PP];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) s[pp] = bc;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            float w[K1];
#pragma unroll
            for (int j = 0; j < K1; ++j) w[j] = a.w1[(c * 2 + ci) * K1 + j];
#pragma unroll
            for (int j = 0; j < K1; ++j)
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) s[pp] = fmaf(w[j], x[ci][pp * S1 + j], s[pp]);
        }
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
            if (p0 + pp < L1) mrow[c * LP + p0 + pp] = swishf(s[pp]);
    }
    __syncthreads();
    const int t0 = q * OPL;
    float sc[2][8][OPL];          // per component and channel: the tap sums of this lane's outputs
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float m[4 * MW4];
#pragma unroll
        for (int i = 0; i < MW4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(mrow + c * LP + t0 + 4 * i);
            m[4 * i] = v[0]; m[4 * i + 1] = v[1]; m[4 * i + 2] = v[2]; m[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            float w[K2];
#pragma unroll
            for (int j = 0; j < K2; ++j) w[j] = a.w2[(co * 8 + c) * K2 + j];
#pragma unroll
            for (int i = 0; i < OPL; ++i) sc[co][c][i] = 0.f;
#pragma unroll
            for (int j = 0; j < K2; ++j)
#pragma unroll
                for (int i = 0; i < OPL; ++i) sc[co][c][i] = fmaf(w[j], m[i + j], sc[co][c][i]);
        }
    }
    if (n >= a.n_nodes || t0 >= TW) return;
    const float b2v[2] = {a.b2[0], a.b2[1]};
    float tcum = 0.f;
    for (int t = 0; t < t0; ++t) tcum += a.dt;
#pragma unroll
    for (int i = 0; i < OPL; ++i) {
        tcum += a.dt;
        if (t0 + i < TW) {
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                // the xor-shuffle tree of decoder2d_kernel: pairs, pairs of pairs, halves
                const float s01 = sc[co][0][i] + sc[co][1][i], s23 = sc[co][2][i] + sc[co][3][i];
                const float s45 = sc[co][4][i] + sc[co][5][i], s67 = sc[co][6][i] + sc[co][7][i];
                const float v = (s01 + s23) + (s45 + s67);
                const size_t o = (size_t)n * 2 * TW + co * TW + t0 + i;
                a.out[o] = a.u[o] + tcum * (v + b2v[co]);
            }
        }
    }
}

}  // namespace msmp

using namespace msmp;

extern "C" int msmp_decoder2d_f32(const float* hd, const float* u, int64_t n_nodes, int tw, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float dt, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(hd && u && w1 && b1 && w2 && b2 && out, MSMP_ERR_ARG, "msmp_decoder2d_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31), MSMP_ERR_ARG, "msmp_decoder2d_f32: bad n_nodes");
    Dec2Args a{hd, u, (long)n_nodes, w1, b1, w2, b2, dt, out};
    const unsigned grid = (unsigned)((n_nodes + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    timing_begin(MSMP_K_DECODER, st);
#define MSMP_DEC2_SPLIT(TW_, K1_, S1_, K2_) do { using G_ = DecSplit<TW_, K1_, S1_, K2_>; \
        hipLaunchKernelGGL((decoder2d_split_kernel<TW_, K1_, S1_, K2_>), dim3((unsigned)((n_nodes + G_::NODES - 1) / G_::NODES)), dim3(G_::NODES * 8), 0, st, a); } while (0)
    if (msmp_tune_get("decoder") && (tw == 25 || tw == 50)) {
        if (tw == 25) MSMP_DEC2_SPLIT(25, 16, 3, 14); else MSMP_DEC2_SPLIT(50, 12, 2, 10);
    } else
#undef MSMP_DEC2_SPLIT_GUARD
    switch (tw) {   // experiments/models_gnn2D.py:79-88
        case 25: hipLaunchKernelGGL((decoder2d_kernel<25, 16, 3, 14>), dim3(grid), dim3(256), 0, st, a); break;
        case 50: hipLaunchKernelGGL((decoder2d_kernel<50, 12, 2, 10>), dim3(grid), dim3(256), 0, st, a); break;
        default:
            set_error("msmp_decoder2d_f32: time_window %d (the reference defines 25, 50)", tw);
            return MSMP_ERR_UNSUPPORTED;
    }
    timing_end(MSMP_K_DECODER, st);
    return check_launch("decoder2d_kernel");
}


extern "C" int msmp_decoder_f32(const float* h, const float* u, int64_t n_nodes, int tw, const float* w1, const float* b1,
                                const float* w2, const float* b2, float dt, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(h && w1 && b1 && w2 && b2 && out, MSMP_ERR_ARG, "msmp_decoder_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31), MSMP_ERR_ARG, "msmp_decoder_f32: bad n_nodes");
    DecArgs a{h, u, (long)n_nodes, w1, b1, w2, b2, dt, out};
    const unsigned grid = (unsigned)((n_nodes + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    timing_begin(MSMP_K_DECODER, st);
#define MSMP_DEC_SPLIT(TW_, K1_, S1_, K2_) do { using G_ = DecSplit<TW_, K1_, S1_, K2_>; \
        hipLaunchKernelGGL((decoder_split_kernel<TW_, K1_, S1_, K2_>), dim3((unsigned)((n_nodes + G_::NODES - 1) / G_::NODES)), dim3(G_::NODES * 8), 0, st, a); } while (0)
    if (msmp_tune_get("decoder")) switch (tw) {   // experiments/models_gnn.py:210-224
        case 20: MSMP_DEC_SPLIT(20, 15, 4, 10); break;
        case 25: MSMP_DEC_SPLIT(25, 16, 3, 14); break;
        case 50: MSMP_DEC_SPLIT(50, 12, 2, 10); break;
        default:
            set_error("msmp_decoder_f32: time_window %d (the reference defines 20, 25, 50)", tw);
            return MSMP_ERR_UNSUPPORTED;
    } else
    switch (tw) {   // the one-lane-per-node edition (msmp_tune("decoder", 0))
        case 20: hipLaunchKernelGGL((decoder_kernel<20, 15, 4, 10>), dim3(grid), dim3(256), 0, st, a); break;
        case 25: hipLaunchKernelGGL((decoder_kernel<25, 16, 3, 14>), dim3(grid), dim3(256), 0, st, a); break;
        case 50: hipLaunchKernelGGL((decoder_kernel<50, 12, 2, 10>), dim3(grid), dim3(256), 0, st, a); break;
        default:
            set_error("msmp_decoder_f32: time_window %d (the reference defines 20, 25, 50)", tw);
            return MSMP_ERR_UNSUPPORTED;
    }
#undef MSMP_DEC_SPLIT
    timing_end(MSMP_K_DECODER, st);
    return check_launch("decoder_kernel");
}
