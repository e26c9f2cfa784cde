// HBM-bound pieces of the message-passing layer: weight packing, mean aggregation (row L2),
// InstanceNorm (row L4), gate blend (row L5), and the whole-layer entry point that chains the pieces.
#include <stdarg.h>
#include "graph_norm.h"
#include <vector>
#include "mfma_tiles.h"

namespace msmp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MSMP_ERR_HIP;
    }
    return MSMP_OK;
}

// ----------------------------------------------------------------------------------------------
// timing aid (bench.py roofline leg)
// ----------------------------------------------------------------------------------------------
struct EventPair {
    hipEvent_t a, b;
};
static int g_timing_mask = 0;
static std::vector<EventPair> g_events[MSMP_K_COUNT];
static size_t g_used[MSMP_K_COUNT] = {};

void timing_begin(int kernel, hipStream_t st) {
    if (!(g_timing_mask >> kernel & 1)) return;
    if (g_used[kernel] == g_events[kernel].size()) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        g_events[kernel].push_back(p);
    }
    (void)hipEventRecord(g_events[kernel][g_used[kernel]].a, st);
}

void timing_end(int kernel, hipStream_t st) {
    if (!(g_timing_mask >> kernel & 1)) return;
    if (g_used[kernel] >= g_events[kernel].size()) return;
    (void)hipEventRecord(g_events[kernel][g_used[kernel]].b, st);
    ++g_used[kernel];
}

// ----------------------------------------------------------------------------------------------
// pack
// ----------------------------------------------------------------------------------------------
struct PackArgs {
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    int tw, nv;
    float* out;
};

__global__ void pack_layer_kernel(PackArgs a) {
    const PackedLayout L = packed_layout(a.tw, a.nv);
    const int k1 = 2 * H + a.tw + 1 + a.nv;   // in-features of message_net_1
    const int k3 = 2 * H + a.nv;              // in-features of update_net_1
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < L.w1s; p += (int64_t)gridDim.x * blockDim.x) {
        if (p >= L.w3s && p < L.w1) continue;  // split copies and scales: written by the two kernels below
        float v = 0.f;
        if (p < L.w4) {                       // w3 chunks
            const int64_t o = p - L.w3;
            const int ch = (int)(o / CHUNK_FLOATS), row = (int)(o % CHUNK_FLOATS) / KC, kk = (int)(o % KC);
            v = a.w3[(size_t)row * k3 + ch * KC + kk];
        } else if (p < L.b1) {                // w4 chunks
            const int64_t o = p - L.w4;
            const int ch = (int)(o / CHUNK_FLOATS), row = (int)(o % CHUNK_FLOATS) / KC, kk = (int)(o % KC);
            v = a.w4[(size_t)row * H + ch * KC + kk];
        } else if (p < L.b2) v = a.b1[p - L.b1];
        else if (p < L.b3) v = a.b2[p - L.b2];
        else if (p < L.b4) v = a.b3[p - L.b3];
        else if (p < L.w3v) v = a.b4[p - L.b4];
        else if (p < L.w3s) {                 // variables columns of update_net_1
            const int64_t o = p - L.w3v;
            const int row = (int)(o / MSMP_MAX_VARS), vv = (int)(o % MSMP_MAX_VARS);
            v = vv < a.nv ? a.w3[(size_t)row * k3 + 2 * H + vv] : 0.f;
        } else if (p < L.w2) {                // w1 chunks, zero padded past k1
            const int64_t o = p - L.w1;
            const int ch = (int)(o / CHUNK_FLOATS), row = (int)(o % CHUNK_FLOATS) / KC, kk = (int)(o % KC);
            const int k = ch * KC + kk;
            v = k < k1 ? a.w1[(size_t)row * k1 + k] : 0.f;
        } else {                              // w2 chunks
            const int64_t o = p - L.w2;
            const int ch = (int)(o / CHUNK_FLOATS), row = (int)(o % CHUNK_FLOATS) / KC, kk = (int)(o % KC);
            v = a.w2[(size_t)row * H + ch * KC + kk];
        }
        a.out[p] = v;
    }
}

// Power-of-two scale of each weight matrix for the fp16-split copies: max|w| * 2^s in [16, 32).
// grid = 4 (one block per matrix); writes scales[i] = 2^s, scales[4 + i] = 2^-s.
__global__ __launch_bounds__(256) void pack_layer_scale_kernel(PackArgs a) {
    __shared__ float red[256];
    const PackedLayout L = packed_layout(a.tw, a.nv);
    const int k1 = 2 * H + a.tw + 1 + a.nv, k3 = 2 * H + a.nv;
    const float* w = blockIdx.x == 0 ? a.w1 : blockIdx.x == 1 ? a.w2 : blockIdx.x == 2 ? a.w3 : a.w4;
    const int n = H * (blockIdx.x == 0 ? k1 : blockIdx.x == 2 ? k3 : H);
    // n is a multiple of 128; four independent 16-byte loads in flight per thread (one 4-byte load per step left this kernel at
    // 22 us: 12 of them per training iteration)
    f32x4 mv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const int n4 = n / 4;
    for (int i = threadIdx.x; i < n4; i += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = i + 256 * u;
            const f32x4 v = w4[k < n4 ? k : i];
#pragma unroll
            for (int x = 0; x < 4; ++x) mv[u][x] = fmaxf(mv[u][x], fabsf(v[x]));
        }
    }
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int x = 0; x < 4; ++x) m = fmaxf(m, mv[u][x]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
        const int sft = mx > 0.f ? 5 - e : 0;                        // mx * 2^sft in [16, 32)
        a.out[L.scales + blockIdx.x] = ldexpf(1.0f, sft);
        a.out[L.scales + 4 + blockIdx.x] = ldexpf(1.0f, -sft);
    }
}

// fp16-split copies of the four weight matrices (mfma_tiles.h): element index within a 8192-half chunk is
// (((s*4 + T)*2 + plane)*64 + lane)*8 + j; values are W * 2^s split into hi + lo.
__global__ void pack_layer_split_kernel(PackArgs a) {
    const PackedLayout L = packed_layout(a.tw, a.nv);
    const int k1 = 2 * H + a.tw + 1 + a.nv, k3 = 2 * H + a.nv;
    _Float16* out_a = reinterpret_cast<_Float16*>(a.out + L.w3s);     // w3s, w4s
    _Float16* out_b = reinterpret_cast<_Float16*>(a.out + L.w1s);     // w1s, w2s
    const float* sc = a.out + L.scales;
    const int64_t n_half = (int64_t)(12 + L.nc1 + 4) * 8192;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_half; p += (int64_t)gridDim.x * blockDim.x) {
        int ch = (int)(p >> 13);
        const int idx = (int)(p & 8191);
        const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1, T = (idx >> 10) & 3, s = (idx >> 12) & 1;
        const int row = 32 * T + (lane & 31), h = lane >> 5;
        float w;
        if (ch < 8) {                                   // w3s: natural order over [h | agg] columns
            w = a.w3[(size_t)row * k3 + 32 * ch + split_k_natural(s, h, j)] * sc[2];
        } else if (ch < 12) {                           // w4s: acc order
            w = a.w4[(size_t)row * H + 32 * (ch - 8) + split_k_acc(s, h, j)] * sc[3];
        } else if (ch < 12 + L.nc1) {                   // w1s: natural order, zero padded past k1
            const int k = 32 * (ch - 12) + split_k_natural(s, h, j);
            w = k < k1 ? a.w1[(size_t)row * k1 + k] * sc[0] : 0.f;
        } else {                                        // w2s: acc order
            w = a.w2[(size_t)row * H + 32 * (ch - 12 - L.nc1) + split_k_acc(s, h, j)] * sc[1];
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 val = plane == 0 ? hi : (_Float16)(w - (float)hi);
        if (ch < 12) out_a[p] = val;
        else out_b[p - 12 * 8192] = val;
    }
    // update_net_2 once more with the rows dealt round-robin over the four tiles (row of (T, lane) = 4 (lane & 31) + T)
    _Float16* out_t = reinterpret_cast<_Float16*>(a.out + L.w4t);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < (int64_t)4 * 8192; p += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(p >> 13), idx = (int)(p & 8191);
        const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1, T = (idx >> 10) & 3, s = (idx >> 12) & 1;
        const float w = a.w4[(size_t)(4 * (lane & 31) + T) * H + 32 * ch + split_k_acc(s, lane >> 5, j)] * sc[3];
        const _Float16 hi = (_Float16)w;
        out_t[p] = plane == 0 ? hi : (_Float16)(w - (float)hi);
        // message_net_2 likewise, natural k order (its A operand is gathered from memory)
        const float w2 = a.w2[(size_t)(4 * (lane & 31) + T) * H + 32 * ch + split_k_natural(s, lane >> 5, j)] * sc[1];
        const _Float16 hi2 = (_Float16)w2;
        reinterpret_cast<_Float16*>(a.out + L.w2t)[p] = plane == 0 ? hi2 : (_Float16)(w2 - (float)hi2);
    }
    // message_net_1 with dealt rows (natural k order, zero padded past k1): node_proj's transposed B operand
    _Float16* out_w1t = reinterpret_cast<_Float16*>(a.out + L.w1t);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < (int64_t)L.nc1 * 8192; p += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(p >> 13), idx = (int)(p & 8191);
        const int j = idx & 7, lane = (idx >> 3) & 63, plane = (idx >> 9) & 1, T = (idx >> 10) & 3, s = (idx >> 12) & 1;
        const int k = 32 * ch + split_k_natural(s, lane >> 5, j);
        const float w = k < k1 ? a.w1[(size_t)(4 * (lane & 31) + T) * k1 + k] * sc[0] : 0.f;
        const _Float16 hi = (_Float16)w;
        out_w1t[p] = plane == 0 ? hi : (_Float16)(w - (float)hi);
    }
    // variables columns of update_net_1 as slot fragments (A operand: lane = row, slots 16 m + 8 h + j)
    _Float16* out_v = reinterpret_cast<_Float16*>(a.out + L.w3vh);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < 2 * VAR_SLOT_FLOATS; p += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(p & 7), lane = (int)(p >> 3) & 63, T = (int)(p >> 9) & 3, m = (int)(p >> 11);
        const int slot = 16 * m + 8 * (lane >> 5) + j, f = slot & 7, part = var_slot_part(slot), row = 32 * T + (lane & 31);
        _Float16 v = (_Float16)0.f;
        if (part < 3 && f < a.nv) {
            const float w = a.w3[(size_t)row * k3 + 2 * H + f] * sc[2];
            const _Float16 hi = (_Float16)w;
            v = part < 2 ? hi : (_Float16)(w - (float)hi);
        }
        out_v[p] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// L2: CSR segmented mean.  32 lanes x 16 B cover one 512-B message row; a wave reduces two nodes at a
// time, a 256-thread block eight.  Rows of one node are summed in CSR order (fixed -> bitwise
// reproducible); loads are independent of each other, so the wave keeps `deg` rows in flight.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_mean_kernel(const float* __restrict__ msg, const int* __restrict__ rowptr,
                                                           long n_nodes, float* __restrict__ agg, int* status) {
    const int sub = threadIdx.x & 31;
    const long node = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (node >= n_nodes) return;
    const int r0 = rowptr[node], r1 = rowptr[node + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* p = reinterpret_cast<const f32x4*>(msg) + (size_t)r0 * (H / 4) + sub;
    int r = r0;
    for (; r + 4 <= r1; r += 4, p += 4 * (H / 4)) {
        const f32x4 a0 = p[0], a1 = p[H / 4], a2 = p[2 * (H / 4)], a3 = p[3 * (H / 4)];
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; r < r1; ++r, p += H / 4) s += p[0];
    const float inv = 1.0f / (float)max(r1 - r0, 1);
    const f32x4 res = s * inv;
    reinterpret_cast<f32x4*>(agg)[(size_t)node * (H / 4) + sub] = res;
    if (out_of_range(res)) status_raise(status, MSMP_STATUS_NODE_SATURATED);
}

// ----------------------------------------------------------------------------------------------
// L4 / L5: per-graph statistics.  One 256-thread workgroup per graph: thread = (16-B channel group
// cg = tid & 31, row slice rs = tid >> 5); two passes (mean, then centred second moment), as PyG's
// InstanceNorm does; a graph's rows (nx x 512 B ~ 51 KB) stay in L2 between the passes.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instance_norm_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                                            float eps, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    f32x4 mean, rstd;
    graph_stats(x, n0, n1, cg, rs, red, eps, mean, rstd);
    const f32x4* xp = reinterpret_cast<const f32x4*>(x);
    f32x4* op = reinterpret_cast<f32x4*>(out);
    for (int r = n0 + rs; r < n1; r += 8) op[(size_t)r * (H / 4) + cg] = (xp[(size_t)r * (H / 4) + cg] - mean) * rstd;
}

__global__ __launch_bounds__(256) void gate_blend_kernel(const float* __restrict__ h, const float* __restrict__ gate,
                                                         const float* __restrict__ mainp, const int* __restrict__ graph_ptr,
                                                         float eps, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    f32x4 gm, gr, mm, mr;
    graph_stats(gate, n0, n1, cg, rs, red, eps, gm, gr);
    graph_stats(mainp, n0, n1, cg, rs, red, eps, mm, mr);
    const f32x4* hp = reinterpret_cast<const f32x4*>(h);
    const f32x4* gp = reinterpret_cast<const f32x4*>(gate);
    const f32x4* mp = reinterpret_cast<const f32x4*>(mainp);
    f32x4* op = reinterpret_cast<f32x4*>(out);
    for (int r = n0 + rs; r < n1; r += 8) {
        const size_t o = (size_t)r * (H / 4) + cg;
        const f32x4 g = (gp[o] - gm) * gr, mnv = (mp[o] - mm) * mr, hv = hp[o];
        f32x4 res;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float tau = sigmoidf_(g[m]);
            res[m] = (1.0f - tau) * hv[m] + tau * swishf(mnv[m]);
        }
        op[o] = res;
    }
}

// Register-resident editions for graphs of at most 8*RPT nodes (E2: 100 -> RPT = 13): every row of the graph is
// read from HBM ONCE and kept in registers across the mean pass, the centred-variance pass and the apply pass
// (the generic kernels above re-read it three times: with 2048 graphs in flight the rows do not survive in L2).
// Same arithmetic in the same order (results agree with the generic kernels to the last bit or two: the compiler
// contracts the two loop forms differently).
template <int RPT>
__device__ __forceinline__ void stats_from_regs(const f32x4 (&v)[RPT], int cnt, int n_rows, int cg, int rs, f32x4* red, float eps,
                                                f32x4& mean, f32x4& rstd) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) s += v[i];
    const float inv = 1.0f / (float)max(n_rows, 1);
    mean = block_colsum(s, red, cg, rs) * inv;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) {
            const f32x4 d = v[i] - mean;
            q += d * d;
        }
    const f32x4 var = block_colsum(q, red, cg, rs) * inv;
#pragma unroll
    for (int m = 0; m < 4; ++m) rstd[m] = 1.0f / sqrtf(var[m] + eps);
}

template <int RPT>
__global__ __launch_bounds__(256) void gate_blend_reg_kernel(const float* __restrict__ h, const float* __restrict__ gate,
                                                             const float* __restrict__ mainp, const int* __restrict__ graph_ptr,
                                                             float eps, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    const int cnt = (n1 - n0 - rs + 7) / 8;              // rows n0 + rs + 8 i, i < cnt
    const f32x4* gp = reinterpret_cast<const f32x4*>(gate);
    const f32x4* mp = reinterpret_cast<const f32x4*>(mainp);
    f32x4 gv[RPT], mv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) {
            const size_t o = (size_t)(n0 + rs + 8 * i) * (H / 4) + cg;
            gv[i] = gp[o];
            mv[i] = mp[o];
        }
    f32x4 gm, gr, mm, mr;
    stats_from_regs<RPT>(gv, cnt, n1 - n0, cg, rs, red, eps, gm, gr);
    stats_from_regs<RPT>(mv, cnt, n1 - n0, cg, rs, red, eps, mm, mr);
    const f32x4* hp = reinterpret_cast<const f32x4*>(h);
    f32x4* op = reinterpret_cast<f32x4*>(out);
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) {
            const size_t o = (size_t)(n0 + rs + 8 * i) * (H / 4) + cg;
            const f32x4 g = (gv[i] - gm) * gr, mnv = (mv[i] - mm) * mr, hv = hp[o];
            f32x4 res;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float tau = sigmoidf_(g[m]);
                res[m] = (1.0f - tau) * hv[m] + tau * swishf(mnv[m]);
            }
            op[o] = res;
        }
}

template <int RPT>
__global__ __launch_bounds__(256) void instance_norm_reg_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                                                float eps, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    const int cg = threadIdx.x & 31, rs = threadIdx.x >> 5;
    const int n0 = graph_ptr[blockIdx.x], n1 = graph_ptr[blockIdx.x + 1];
    const int cnt = (n1 - n0 - rs + 7) / 8;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x);
    f32x4 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) v[i] = xp[(size_t)(n0 + rs + 8 * i) * (H / 4) + cg];
    f32x4 mean, rstd;
    stats_from_regs<RPT>(v, cnt, n1 - n0, cg, rs, red, eps, mean, rstd);
    f32x4* op = reinterpret_cast<f32x4*>(out);
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (i < cnt) op[(size_t)(n0 + rs + 8 * i) * (H / 4) + cg] = (v[i] - mean) * rstd;
}

}  // namespace msmp

using namespace msmp;

extern "C" int msmp_timing_enable(int kernel_mask) {
    g_timing_mask = kernel_mask & ((1 << MSMP_K_COUNT) - 1);
    return MSMP_OK;
}

extern "C" int msmp_timing_reset(void) {
    for (int k = 0; k < MSMP_K_COUNT; ++k) g_used[k] = 0;
    return MSMP_OK;
}

extern "C" int msmp_timing_read(int kernel, int64_t* launches_out, double* total_ms_out) {
    MSMP_REQUIRE(kernel >= 0 && kernel < MSMP_K_COUNT && launches_out && total_ms_out, MSMP_ERR_ARG, "msmp_timing_read: bad argument");
    double tot = 0.0;
    for (size_t i = 0; i < g_used[kernel]; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_events[kernel][i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_events[kernel][i].a, g_events[kernel][i].b);
        if (e != hipSuccess) {
            set_error("msmp_timing_read: %s", hipGetErrorString(e));
            return MSMP_ERR_HIP;
        }
        tot += ms;
    }
    *launches_out = (int64_t)g_used[kernel];
    *total_ms_out = tot;
    return MSMP_OK;
}

// ---- sticky range status ----------------------------------------------------------------------------------------------
// One int in host-mapped (fine-grained) memory: kernels reach it with system-scope atomics only when something is out of range,
// the host reads it like any variable.  If the mapping cannot be made the word lives in device memory and msmp_last_status pays
// a blocking copy.
static int* g_status_host = nullptr;
static int* g_status_dev = nullptr;
static int g_status_mode = 0;       // 0 not created, 1 host-mapped, 2 device memory, -1 unavailable
namespace msmp {
int* status_ptr() {
    if (g_status_mode == 0) {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            g_status_host = static_cast<int*>(hp);
            *g_status_host = 0;
            g_status_dev = static_cast<int*>(dp);
            g_status_mode = 1;
        } else {
            (void)hipGetLastError();
            if (hipMalloc(&dp, sizeof(int)) == hipSuccess && hipMemset(dp, 0, sizeof(int)) == hipSuccess) {
                g_status_dev = static_cast<int*>(dp);
                g_status_mode = 2;
            } else {
                (void)hipGetLastError();
                g_status_mode = -1;
            }
        }
    }
    return g_status_dev;
}
}  // namespace msmp

extern "C" int msmp_last_status(int* flags_out, int reset) {
    MSMP_REQUIRE(flags_out, MSMP_ERR_ARG, "msmp_last_status: null pointer");
    int* dp = status_ptr();
    *flags_out = 0;
    if (g_status_mode == 1) {
        *flags_out = __atomic_load_n(g_status_host, __ATOMIC_RELAXED);
        if (reset) __atomic_store_n(g_status_host, 0, __ATOMIC_RELAXED);
    } else if (g_status_mode == 2) {
        hipError_t e = hipMemcpy(flags_out, dp, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess && reset) e = hipMemset(dp, 0, sizeof(int));
        MSMP_REQUIRE(e == hipSuccess, MSMP_ERR_HIP, "msmp_last_status: %s", hipGetErrorString(e));
    } else {
        MSMP_REQUIRE(false, MSMP_ERR_HIP, "msmp_last_status: no status word (allocation failed)");
    }
    return MSMP_OK;
}

extern "C" int msmp_version(void) { return MSMP_ABI_VERSION; }
extern "C" const char* msmp_last_error(void) { return g_err; }

extern "C" int64_t msmp_packed_layer_floats(int tw, int nv) {
    if (tw <= 0 || nv < 1 || nv > MSMP_MAX_VARS) return -1;
    return packed_layout(tw, nv).total;
}

extern "C" int msmp_pack_layer_f32(const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* w3, const float* b3, const float* w4, const float* b4,
                                   int tw, int nv, float* packed_out, msmp_stream_t stream) {
    MSMP_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4 && packed_out, MSMP_ERR_ARG, "msmp_pack_layer_f32: null pointer");
    MSMP_REQUIRE(tw > 0 && nv >= 1 && nv <= MSMP_MAX_VARS, MSMP_ERR_ARG, "msmp_pack_layer_f32: bad tw=%d nv=%d", tw, nv);
    PackArgs a{w1, b1, w2, b2, w3, b3, w4, b4, tw, nv, packed_out};
    hipLaunchKernelGGL(pack_layer_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pack_layer_scale_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pack_layer_split_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("pack_layer_kernel");
}

extern "C" int msmp_scatter_mean_f32(const float* msg, const int32_t* rowptr, int64_t n_nodes, float* agg_out,
                                     msmp_stream_t stream) {
    MSMP_REQUIRE(rowptr && agg_out, MSMP_ERR_ARG, "msmp_scatter_mean_f32: null pointer");
    MSMP_REQUIRE(n_nodes > 0 && n_nodes < (1L << 31), MSMP_ERR_ARG, "msmp_scatter_mean_f32: bad n_nodes");
    const unsigned grid = (unsigned)((n_nodes + 7) / 8);
    timing_begin(MSMP_K_SCATTER_MEAN, (hipStream_t)stream);
    hipLaunchKernelGGL(scatter_mean_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, msg, rowptr, (long)n_nodes, agg_out, msmp_tune_get("split") ? status_ptr() : nullptr);
    timing_end(MSMP_K_SCATTER_MEAN, (hipStream_t)stream);
    return check_launch("scatter_mean_kernel");
}

// largest graph (nodes) the register-resident norm kernels are built for: 8 row slices x 16 rows per thread
static const int kRegNormMaxNodes = 128;

extern "C" int msmp_instance_norm_f32(const float* x, const int32_t* graph_ptr, int64_t n_graphs, int max_graph_nodes, float eps,
                                      float* out, msmp_stream_t stream) {
    const int g_max_graph_nodes = max_graph_nodes;
    MSMP_REQUIRE(x && graph_ptr && out, MSMP_ERR_ARG, "msmp_instance_norm_f32: null pointer");
    MSMP_REQUIRE(n_graphs > 0 && n_graphs < (1L << 31), MSMP_ERR_ARG, "msmp_instance_norm_f32: bad n_graphs");
    timing_begin(MSMP_K_NORM, (hipStream_t)stream);
    if (g_max_graph_nodes > 0 && g_max_graph_nodes <= 104)
        hipLaunchKernelGGL(instance_norm_reg_kernel<13>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, graph_ptr, eps, out);
    else if (g_max_graph_nodes > 0 && g_max_graph_nodes <= kRegNormMaxNodes)
        hipLaunchKernelGGL(instance_norm_reg_kernel<16>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, graph_ptr, eps, out);
    else
    hipLaunchKernelGGL(instance_norm_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, graph_ptr, eps, out);
    timing_end(MSMP_K_NORM, (hipStream_t)stream);
    return check_launch("instance_norm_kernel");
}

extern "C" int msmp_gate_blend_f32(const float* h, const float* gate_pre, const float* main_pre, const int32_t* graph_ptr,
                                   int64_t n_graphs, int max_graph_nodes, float eps, float* out, msmp_stream_t stream) {
    const int g_max_graph_nodes = max_graph_nodes;
    MSMP_REQUIRE(h && gate_pre && main_pre && graph_ptr && out, MSMP_ERR_ARG, "msmp_gate_blend_f32: null pointer");
    MSMP_REQUIRE(n_graphs > 0 && n_graphs < (1L << 31), MSMP_ERR_ARG, "msmp_gate_blend_f32: bad n_graphs");
    timing_begin(MSMP_K_NORM, (hipStream_t)stream);
    if (g_max_graph_nodes > 0 && g_max_graph_nodes <= 104)
        hipLaunchKernelGGL(gate_blend_reg_kernel<13>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, h, gate_pre,
                           main_pre, graph_ptr, eps, out);
    else if (g_max_graph_nodes > 0 && g_max_graph_nodes <= kRegNormMaxNodes)
        hipLaunchKernelGGL(gate_blend_reg_kernel<16>, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, h, gate_pre,
                           main_pre, graph_ptr, eps, out);
    else
    hipLaunchKernelGGL(gate_blend_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, h, gate_pre, main_pre,
                       graph_ptr, eps, out);
    timing_end(MSMP_K_NORM, (hipStream_t)stream);
    return check_launch("gate_blend_kernel");
}

// Workspace of the chained layer: [msg [E,128] unless the fused edge kernel applies] | agg [N,128] |
// pre_main [N,128] | pre_gate [N,128] | P [N,128] | Q [N,128] | Q' [N,128] (second head's Q when a gated pair is projected in one launch;
// its P takes the pre_main slot, unused on that path)
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static bool fused_ok(int max_in_degree) { return max_in_degree >= 0 && max_in_degree <= 256; }

extern "C" size_t msmp_mp_layer_workspace_bytes(int64_t n_nodes, int64_t n_edges, int gated, int max_in_degree) {
    const size_t msg = fused_ok(max_in_degree) ? 0 : align256((size_t)n_edges * H * sizeof(float));
    const size_t nod = align256((size_t)n_nodes * H * sizeof(float));
    (void)gated;
    return msg + nod * 6 + 256;
}

static int mp_layer_impl(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                         const int32_t* rowptr, const int32_t* col, const int32_t* tgt, const msmp_tiles_t* tiles,
                         const int32_t* graph_ptr,
                         int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int max_in_degree, int max_graph_nodes,
                         int tw, int nv, const float* packed_main, const float* packed_gate, int mode, float eps, float* h_out,
                         const msmp_decoder_t* dec, void* workspace, size_t workspace_bytes, msmp_stream_t stream);

extern "C" int msmp_mp_layer_f32(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                                 const int32_t* rowptr, const int32_t* col, const int32_t* tgt, const msmp_tiles_t* tiles,
                                 const int32_t* graph_ptr,
                                 int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int max_in_degree, int max_graph_nodes,
                                 int tw, int nv, const float* packed_main, const float* packed_gate, int mode, float eps, float* h_out,
                                 void* workspace, size_t workspace_bytes, msmp_stream_t stream) {
    return mp_layer_impl(h, u, pos, vars, feat, rowptr, col, tgt, tiles, graph_ptr, n_nodes, n_edges, n_graphs, max_in_degree, max_graph_nodes, tw, nv,
                         packed_main, packed_gate, mode, eps, h_out, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int msmp_mp_layer_decode_f32(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                                        const int32_t* rowptr, const int32_t* col, const int32_t* tgt, const msmp_tiles_t* tiles,
                                        const int32_t* graph_ptr,
                                        int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int max_in_degree, int max_graph_nodes,
                                        int tw, int nv, const float* packed_main, const float* packed_gate, int mode, float eps, float* h_out,
                                        const msmp_decoder_t* dec, void* workspace, size_t workspace_bytes, msmp_stream_t stream) {
    MSMP_REQUIRE(dec, MSMP_ERR_ARG, "msmp_mp_layer_decode_f32: null decoder description");
    MSMP_REQUIRE(dec->time_window == 25 && msmp_tune_get("split") && msmp_tune_get("tail") && max_graph_nodes > 0 && max_graph_nodes <= 128,
                 MSMP_ERR_UNSUPPORTED, "msmp_mp_layer_decode_f32: the fused tail does not apply (time_window %d, max_graph_nodes %d)", dec->time_window,
                 max_graph_nodes);
    return mp_layer_impl(h, u, pos, vars, feat, rowptr, col, tgt, tiles, graph_ptr, n_nodes, n_edges, n_graphs, max_in_degree, max_graph_nodes, tw, nv,
                         packed_main, packed_gate, mode, eps, h_out, dec, workspace, workspace_bytes, stream);
}

static int mp_layer_impl(const float* h, const float* u, const float* pos, const float* vars, const float* feat,
                         const int32_t* rowptr, const int32_t* col, const int32_t* tgt, const msmp_tiles_t* tiles,
                         const int32_t* graph_ptr,
                         int64_t n_nodes, int64_t n_edges, int64_t n_graphs, int max_in_degree, int max_graph_nodes,
                         int tw, int nv, const float* packed_main, const float* packed_gate, int mode, float eps, float* h_out,
                         const msmp_decoder_t* dec, void* workspace, size_t workspace_bytes, msmp_stream_t stream) {
    MSMP_REQUIRE(h && u && pos && vars && rowptr && col && tgt && graph_ptr && packed_main && h_out && workspace,
                 MSMP_ERR_ARG, "msmp_mp_layer_f32: null pointer");
    MSMP_REQUIRE(h_out != h, MSMP_ERR_ARG, "msmp_mp_layer_f32: h_out may not alias h");
    MSMP_REQUIRE(!tiles || msmp_tiles_ok(tiles, n_nodes), MSMP_ERR_ARG, "msmp_mp_layer_f32: tile descriptor does not cover %ld nodes", (long)n_nodes);
    const int gated = packed_gate != nullptr;
    const bool dense = (mode & MSMP_LAYER_DENSE_MESSAGE) != 0;
    mode &= ~MSMP_LAYER_DENSE_MESSAGE;
    MSMP_REQUIRE(!gated || mode == MSMP_LAYER_LIN, MSMP_ERR_ARG, "msmp_mp_layer_f32: the gated pair uses GNN_LayerLin layers");
    const size_t need = msmp_mp_layer_workspace_bytes(n_nodes, n_edges, gated, max_in_degree);
    MSMP_REQUIRE(workspace_bytes >= need, MSMP_ERR_WORKSPACE, "msmp_mp_layer_f32: workspace %zu < %zu", workspace_bytes, need);
    const bool fused = fused_ok(max_in_degree);
    char* ws = (char*)workspace;
    float* msg = (float*)ws;
    if (!fused) ws += align256((size_t)n_edges * H * sizeof(float));
    const size_t nod = align256((size_t)n_nodes * H * sizeof(float));
    float* const agg = (float*)ws;
    float* pre_main = (float*)(ws + nod);
    float* pre_gate = (float*)(ws + 2 * nod);
    float* pbuf = (float*)(ws + 3 * nod);
    float* qbuf = (float*)(ws + 4 * nod);
    int rc;
    // message + mean (rows L1 + L2): one fused launch when every target's in-edges fit a workgroup tile
    // Tiles pay when they are full: a tile is a 128-edge block of matrix work whatever it holds.  Structures whose tiles stay half
    // empty (32 node slots are used up before 128 edges: knn graphs of in-degree 3 on a scattered grid, RPU: 66 edges per tile)
    // run the untiled kernels (measured at 2048 graphs, ms per rollout step, tiled / untiled: RPU 7.00 / 6.77, WE3 at 84 edges
    // per tile 5.72 / 5.92, E2 at 123: 6.6 / 7.1).  "tile" 3 forces the tiles.
    int tile_mode = (tiles && n_edges > 0 && msmp_tune_get("split")) ? msmp_tune_get("tile") : 0;
    if (tile_mode == 2 && n_edges < (int64_t)76 * tiles->n_tiles) tile_mode = 0;
    if (tile_mode == 3) tile_mode = 2;
    auto aggregate = [&](const float* packed, float* agg) -> int {
        if (fused && !dense && tile_mode == 2)       // node tiles staged in LDS, P / Q computed in the workgroup
            return msmp_edge_aggregate_tiled_f32(h, u, pos, vars, feat, nullptr, nullptr, rowptr, tiles, n_nodes, n_edges, tw, nv, packed, agg, stream);
        if (fused && !dense && tile_mode == 1) {
            const int r = msmp_node_project_f32(h, u, pos, vars, n_nodes, tw, nv, packed, pbuf, qbuf, stream);
            return r ? r : msmp_edge_aggregate_tiled_f32(nullptr, nullptr, nullptr, nullptr, nullptr, pbuf, qbuf, rowptr, tiles, n_nodes, n_edges, tw,
                                                         nv, packed, agg, stream);
        }
        if (fused && !dense) {
            const int r = msmp_node_project_f32(h, u, pos, vars, n_nodes, tw, nv, packed, pbuf, qbuf, stream);
            return r ? r : msmp_edge_aggregate_projected_f32(pbuf, qbuf, rowptr, col, tgt, n_nodes, n_edges, max_in_degree, tw,
                                                              nv, packed, agg, stream);
        }
        if (fused)
            return msmp_edge_aggregate_f32(h, u, pos, vars, rowptr, col, tgt, n_nodes, n_edges, max_in_degree, tw, nv, packed,
                                           agg, stream);
        const int r = msmp_edge_mlp_f32(h, u, pos, vars, tgt, col, n_nodes, n_edges, tw, nv, packed, msg, stream);
        return r ? r : msmp_scatter_mean_f32(msg, rowptr, n_nodes, agg, stream);
    };
    // node tail (rows L3-L5): one launch per layer when the graphs fit a workgroup (update head(s) + InstanceNorm + blend)
    if (msmp_tune_get("split") && msmp_tune_get("tail") && max_graph_nodes > 0 && max_graph_nodes <= 128) {
        rc = MSMP_ERR_UNSUPPORTED;
        if (gated && fused && !dense && tile_mode == 2 && msmp_tune_get("pair") && (msmp_tune_get("pair") == 2 || n_nodes <= 65536))
            rc = msmp_edge_aggregate_tiled_pair(h, u, pos, vars, feat, rowptr, tiles, n_nodes, n_edges, tw, nv, packed_gate, packed_main, pre_gate,
                                                agg, stream);      // small batches: both heads in one launch
        else if (gated && fused && !dense && !tile_mode)       // small batches: both heads per launch (projection, then message + mean)
            rc = msmp_pair_project_aggregate(h, u, pos, vars, rowptr, col, tgt, n_nodes, n_edges, max_in_degree, tw, nv, packed_gate,
                                             packed_main, pbuf, qbuf, pre_main, (float*)(ws + 5 * nod), pre_gate, agg, stream);
        if (rc == MSMP_ERR_UNSUPPORTED) {
            if (gated && (rc = aggregate(packed_gate, pre_gate))) return rc;       // pre_gate doubles as the gate head's aggregate
            if ((rc = aggregate(packed_main, agg))) return rc;
        } else if (rc) return rc;
        return msmp_node_tail_impl(h, agg, gated ? pre_gate : nullptr, vars, graph_ptr, n_nodes, n_graphs, max_graph_nodes, nv,
                                   packed_main, packed_gate, mode, eps, h_out, dec, stream);
    }
    MSMP_REQUIRE(!dec, MSMP_ERR_UNSUPPORTED, "msmp_mp_layer_decode_f32: the fused tail does not apply");
    if (gated) {
        if ((rc = aggregate(packed_gate, agg))) return rc;
        if ((rc = msmp_node_update_f32(h, agg, vars, n_nodes, nv, packed_gate, MSMP_LAYER_LIN, pre_gate, stream))) return rc;
    }
    if ((rc = aggregate(packed_main, agg))) return rc;
    if ((rc = msmp_node_update_f32(h, agg, vars, n_nodes, nv, packed_main, mode, pre_main, stream))) return rc;
    if (gated) return msmp_gate_blend_f32(h, pre_gate, pre_main, graph_ptr, n_graphs, max_graph_nodes, eps, h_out, stream);
    return msmp_instance_norm_f32(pre_main, graph_ptr, n_graphs, max_graph_nodes, eps, h_out, stream);
}
