// Two-layer node MLP  out = Swish(W2 Swish(W1 x + b1) + b2)  in one launch: the `embedding_mlp` encoder of the LEM-free
// solver classes (experiments/models_gnn.py:196-201 with its call at :269-270; models_gnn2D.py:66-71, 119-120), which was two
// rocBLAS GEMMs and two bias / Swish passes over [N,128] in PyTorch.  x is the concatenated node input
// [u | pos_x | variables] with rows zero-padded to a multiple of 32 floats (K <= 128).
//
// fp16-split matrix path (mfma_tiles.h).  Layer 1 is channel-major like the other node kernels (a node per lane, the lane's
// slices of its input row as B fragments); layer 2 is computed transposed from the Swish output in registers with a
// row-dealt copy of W2, so a lane ends up with four consecutive channels of 16 node rows and `out` is written as 16-byte
// pieces of whole 512-byte rows.
#include "mfma_tiles.h"

namespace msmp {

constexpr int MLP2_MAX_CHUNKS = 4;      // K <= 128

// packed blob (floats): scales [4] (2^s1, 2^s2, 2^-s1, 2^-s2) | pad [28] | b1 [128] | b2 [128] |
//   w1s (nc split chunks, natural k order, zero padded past K) | w2t (4 split chunks, acc order, rows dealt round-robin)
struct Mlp2Layout {
    int nc;
    int64_t scales, b1, b2, w1s, w2t, total;
};
__host__ __device__ inline Mlp2Layout mlp2_layout(int k_in) {
    Mlp2Layout L;
    L.nc = (k_in + KC - 1) / KC;
    int64_t o = 0;
    L.scales = o; o += 32;
    L.b1 = o; o += H;
    L.b2 = o; o += H;
    L.w1s = o; o += (int64_t)L.nc * CHUNK_FLOATS;
    L.w2t = o; o += 4 * CHUNK_FLOATS;
    L.total = o;
    return L;
}

struct Mlp2PackArgs {
    const float *w1, *b1, *w2, *b2;
    int k_in;
    float* out;
};

// grid = 2: block 0 scales W1, block 1 W2 (max|w| 2^s in [16, 32), exact powers of two); block 0 also copies the biases
__global__ __launch_bounds__(256) void mlp2_scale_kernel(Mlp2PackArgs a) {
    __shared__ float red[256];
    const Mlp2Layout L = mlp2_layout(a.k_in);
    const float* w = blockIdx.x == 0 ? a.w1 : a.w2;
    const int n = H * (blockIdx.x == 0 ? a.k_in : H);
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
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
        a.out[L.scales + 2 + blockIdx.x] = ldexpf(1.0f, -sft);
    }
    if (blockIdx.x == 0 && threadIdx.x < H) {
        a.out[L.b1 + threadIdx.x] = a.b1[threadIdx.x];
        a.out[L.b2 + threadIdx.x] = a.b2[threadIdx.x];
    }
}

__global__ void mlp2_split_pack_kernel(Mlp2PackArgs a) {
    const Mlp2Layout L = mlp2_layout(a.k_in);
    const float* sc = a.out + L.scales;
    _Float16* o1 = reinterpret_cast<_Float16*>(a.out + L.w1s);
    _Float16* o2 = reinterpret_cast<_Float16*>(a.out + L.w2t);
    const int64_t n_half = (int64_t)(L.nc + 4) * 8192;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_half; p += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(p >> 13), idx = (int)(p & 8191);
        const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1, T = (idx >> 10) & 3, s = (idx >> 12) & 1;
        const int h = lane >> 5;
        float w;
        if (ch < L.nc) {
            const int k = 32 * ch + split_k_natural(s, h, j);
            w = k < a.k_in ? a.w1[(size_t)(32 * T + (lane & 31)) * a.k_in + k] * sc[0] : 0.f;
        } else {
            w = a.w2[(size_t)(4 * (lane & 31) + T) * H + 32 * (ch - L.nc) + split_k_acc(s, h, j)] * sc[1];
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 v = plane == 0 ? hi : (_Float16)(w - (float)hi);
        if (ch < L.nc) o1[p] = v;
        else o2[p - (int64_t)L.nc * 8192] = v;
    }
}

struct Mlp2Args {
    const float* x;       // [N, 32 nc] zero padded rows
    long n_nodes;
    int nc;
    const float* b1;
    const float* b2;
    const float* w1s;     // nc chunks, followed by w2t (4 chunks)
    const float* scales;
    float* out;           // [N,128]
};

template <int NC>
__global__ __launch_bounds__(256, 2) void mlp2_split_kernel(Mlp2Args a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * SPLIT_CHUNK_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const long n0 = (long)blockIdx.x * 128;
    const long n = n0 + wave * 32 + c;
    const long nc = n < a.n_nodes ? n : a.n_nodes - 1;
    const float sc1 = a.scales[0], sc2 = a.scales[1], inv1 = a.scales[2], inv2 = a.scales[3];

    // the lane's slices of its input row for every k-chunk: floats 32 ch + 16 s + 8 hh .. + 7 (all in flight at once)
    f32x4 xr[NC][4];
    WStage ws;
    wstage_load(ws, a.w1s, tid);
    {
        const float* row = a.x + (size_t)nc * (32 * NC) + 8 * hh;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                xr[ch][2 * s] = *reinterpret_cast<const f32x4*>(row + 32 * ch + 16 * s);
                xr[ch][2 * s + 1] = *reinterpret_cast<const f32x4*>(row + 32 * ch + 16 * s + 4);
            }
    }
    f32x16 z[4][1];
    acc_init_bias_scaled<1>(a.b1, sc1, hh, z);
    wstage_store_linear(ws, lds, tid);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
        wstage_load(ws, a.w1s + (size_t)(ch + 1) * SPLIT_CHUNK_FLOATS, tid);      // chunk NC = first w2t chunk
        half8 bhi[1][2], blo[1][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 v0 = xr[ch][2 * s], v1 = xr[ch][2 * s + 1];
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split8(v, bhi[0][s], blo[0][s]);
        }
        mma_chunk_split<1>(lds + (ch & 1) * SPLIT_CHUNK_FLOATS, lane, bhi, blo, z);
        wstage_store_linear(ws, lds + ((ch + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
        __syncthreads();
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[T][0][r] = swishf(z[T][0][r] * inv1);

    // layer 2 transposed: yT[T][r] = 2^s2 (W2 z + b2)[channel 4 c + T] of node acc_row(r, hh)
    f32x16 yT[4];
    {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2 + 4 * c) * sc2;
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) yT[T][r] = bv[T];
    }
    const float* w2t = a.w1s + (size_t)NC * SPLIT_CHUNK_FLOATS;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) wstage_load(ws, w2t + (size_t)(t + 1) * SPLIT_CHUNK_FLOATS, tid);
        half8 zhi[1][2], zlo[1][2];
        split_acc_tile<1>(z[t], zhi, zlo);
        const half8* w = reinterpret_cast<const half8*>(lds + ((NC + t) & 1) * SPLIT_CHUNK_FLOATS) + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const half8 whi = w[((s * 4 + T) * 2 + 0) * 64], wlo = w[((s * 4 + T) * 2 + 1) * 64];
                MSMP_MFMA_LOLO(2, yT[T], zlo[0][s], wlo);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi[0][s], wlo, yT[T], 0, 0, 0);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zlo[0][s], whi, yT[T], 0, 0, 0);
                yT[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zhi[0][s], whi, yT[T], 0, 0, 0);
            }
        if (t < 3) {
            wstage_store_linear(ws, lds + ((NC + t + 1) & 1) * SPLIT_CHUNK_FLOATS, tid);
            __syncthreads();
        }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) yT[T][r] = swishf(yT[T][r] * inv2);

    const size_t base = ((size_t)n0 + wave * 32 + 4 * hh) * H + 4 * c;
    const long lim = a.n_nodes - n0 - wave * 32 - 4 * hh;
    if (n0 + wave * 32 + 32 <= a.n_nodes) {         // wave-uniform common path: no per-row predication
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *reinterpret_cast<f32x4*>(a.out + base + (size_t)((r & 3) + 8 * (r >> 2)) * H) = f32x4{yT[0][r], yT[1][r], yT[2][r], yT[3][r]};
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);
            if (row < lim) *reinterpret_cast<f32x4*>(a.out + base + (size_t)row * H) = f32x4{yT[0][r], yT[1][r], yT[2][r], yT[3][r]};
        }
    }
}

}  // namespace msmp

using namespace msmp;

extern "C" int64_t msmp_packed_mlp2_floats(int k_in) {
    return k_in >= 1 && k_in <= 32 * MLP2_MAX_CHUNKS ? mlp2_layout(k_in).total : -1;
}

extern "C" int msmp_mlp2_input_stride(int k_in) { return k_in >= 1 && k_in <= 32 * MLP2_MAX_CHUNKS ? 32 * mlp2_layout(k_in).nc : -1; }

extern "C" int msmp_pack_mlp2_f32(const float* w1, const float* b1, const float* w2, const float* b2, int k_in, float* packed_out,
                                  msmp_stream_t stream) {
    MSMP_REQUIRE(w1 && b1 && w2 && b2 && packed_out, MSMP_ERR_ARG, "msmp_pack_mlp2_f32: null pointer");
    MSMP_REQUIRE(k_in >= 1 && k_in <= 32 * MLP2_MAX_CHUNKS, MSMP_ERR_UNSUPPORTED, "msmp_pack_mlp2_f32: in_features %d outside 1..%d", k_in,
                 32 * MLP2_MAX_CHUNKS);
    Mlp2PackArgs a{w1, b1, w2, b2, k_in, packed_out};
    hipLaunchKernelGGL(mlp2_scale_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(mlp2_split_pack_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mlp2_split_pack_kernel");
}

extern "C" int msmp_mlp2_swish_f32(const float* x, int64_t n_nodes, int k_in, const float* packed, float* out, msmp_stream_t stream) {
    MSMP_REQUIRE(x && packed && out, MSMP_ERR_ARG, "msmp_mlp2_swish_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31), MSMP_ERR_ARG, "msmp_mlp2_swish_f32: bad n_nodes");
    MSMP_REQUIRE(k_in >= 1 && k_in <= 32 * MLP2_MAX_CHUNKS, MSMP_ERR_UNSUPPORTED, "msmp_mlp2_swish_f32: in_features %d outside 1..%d", k_in,
                 32 * MLP2_MAX_CHUNKS);
    const Mlp2Layout L = mlp2_layout(k_in);
    Mlp2Args a{x, (long)n_nodes, L.nc, packed + L.b1, packed + L.b2, packed + L.w1s, packed + L.scales, out};
    const unsigned grid = (unsigned)((n_nodes + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    switch (L.nc) {
        case 1: hipLaunchKernelGGL(mlp2_split_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(mlp2_split_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(mlp2_split_kernel<3>, dim3(grid), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(mlp2_split_kernel<4>, dim3(grid), dim3(256), 0, st, a); break;
    }
    return check_launch("mlp2_split_kernel");
}
