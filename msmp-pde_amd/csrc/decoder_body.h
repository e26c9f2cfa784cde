// The 1-D decoder split by position, eight lanes per node (decoder_kernel.hip: decoder_split_kernel) as a device function, so that
// the node tail of the LAST layer pair can run it as its epilogue on the rows it has just written (VERDICT r03 item 4 / SURVEY 8f.4:
// "decoder fusion into the last layer's epilogue").  experiments/models_gnn.py:210-224, 275-279.
#pragma once
#include "msmp_common.h"

namespace msmp {

template <int TW, int K1, int S1, int K2>
struct DecSplit {
    static constexpr int L1 = (H - K1) / S1 + 1;
    static constexpr int PP = (L1 + 7) / 8;                          // intermediate positions per lane
    static constexpr int XW = (PP - 1) * S1 + K1;                    // row values a lane needs
    static constexpr int OPL = TW > 32 ? 8 : 4;                      // consecutive outputs per lane (multiple of 4: aligned 16-byte LDS reads)
    static constexpr int MW = OPL + K2 - 1;                          // intermediate values per channel a lane needs for them
    static constexpr int MW4 = (MW + 3) / 4;
    static constexpr int LP = ((7 * OPL + 4 * MW4 > L1 ? 7 * OPL + 4 * MW4 : L1) + 3) / 4 * 4 + 4;      // padded row of the LDS table
    static constexpr int NODES = 8 * LP * 4 * 32 <= 49152 ? 32 : 16;  // nodes per workgroup (LDS <= 48 KB)
    static_assert(L1 - K2 + 1 == TW && 8 * OPL >= TW, "decoder geometry");
};

struct DecW {
    const float* w1;     // [8][k1]
    const float* b1;     // [8]
    const float* w2;     // [8][k2]
    const float* b2;     // [1]
    const float* u;      // [N, tw] (nullptr: the decoder output alone)
    float dt;
    float* out;          // [N, tw]
};

// One node by eight lanes: lane q (0..7) of node n (row `row` of 128 floats; mrow: the node's [8][LP] LDS table; the callers'
// barrier() separates the two halves: every lane of a node's table must have written before any reads).  L2_ROWS: the row was
// written by OTHER lanes of this workgroup just before (the fused tail): read it past the L1 (agent-scope loads).
// The taps of every sum are added in decoder_kernel's order: the same bits wherever this runs.
template <int TW, int K1, int S1, int K2, bool L2_ROWS, typename Barrier>
__device__ __forceinline__ void decoder_split_node(const float* row, float* mrow, int q, bool live, long n, const DecW& a, Barrier barrier) {
    using G = DecSplit<TW, K1, S1, K2>;
    constexpr int L1 = G::L1, PP = G::PP, XW = G::XW, OPL = G::OPL, MW4 = G::MW4, LP = G::LP;
    const int p0 = q * PP;
    float x[XW];
    {
        const int x0 = p0 * S1;
#pragma unroll
        for (int i = 0; i < XW; ++i) {
            const float* px = row + (x0 + i < H ? x0 + i : H - 1);
            x[i] = L2_ROWS ? __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *px;
        }
    }
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        float w1c[K1];
#pragma unroll
        for (int j = 0; j < K1; ++j) w1c[j] = a.w1[c * K1 + j];
        const float bc = a.b1[c];
        float s[PP];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) s[pp] = bc;
#pragma unroll
        for (int j = 0; j < K1; ++j)
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) s[pp] = fmaf(w1c[j], x[pp * S1 + j], s[pp]);
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
            if (p0 + pp < L1) mrow[c * LP + p0 + pp] = swishf(s[pp]);
    }
    barrier();
    // ---- outputs t0 .. t0 + OPL - 1 of this lane -------------------------------------------------------------------------
    const int t0 = q * OPL;
    float o[OPL];
    const float bias2 = a.b2[0];
#pragma unroll
    for (int i = 0; i < OPL; ++i) o[i] = bias2;
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        float w2c[K2];
#pragma unroll
        for (int j = 0; j < K2; ++j) w2c[j] = a.w2[c * K2 + j];
        float m[4 * MW4];
#pragma unroll
        for (int i = 0; i < MW4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(mrow + c * LP + t0 + 4 * i);
            m[4 * i] = v[0]; m[4 * i + 1] = v[1]; m[4 * i + 2] = v[2]; m[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < K2; ++j)
#pragma unroll
            for (int i = 0; i < OPL; ++i) o[i] = fmaf(w2c[j], m[i + j], o[i]);
    }
    if (!live || t0 >= TW) return;
    float* op = a.out + (size_t)n * TW;
    if (a.u == nullptr) {
#pragma unroll
        for (int i = 0; i < OPL; ++i)
            if (t0 + i < TW) op[t0 + i] = o[i];
        return;
    }
    const float ul = a.u[(size_t)n * TW + TW - 1];
    float tcum = 0.f;
    for (int t = 0; t < t0; ++t) tcum += a.dt;          // cumsum of a constant, float32 partial sums like torch.cumsum on the device
#pragma unroll
    for (int i = 0; i < OPL; ++i) {
        tcum += a.dt;
        if (t0 + i < TW) op[t0 + i] = ul + tcum * o[i];
    }
}

}  // namespace msmp
