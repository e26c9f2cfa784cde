// Microbenchmark: how does v_mfma_f32_32x32x16_f16 round?  (1) directed cases on the accumulator, (2) statistics of a K = 128
// dot product accumulated over 8 chained MFMAs against the exact (double) result, next to a sequential fmaf chain and a
// pairwise fp32 sum.   Build: hipcc --offload-arch=gfx950 -O3 -o mfma_round.bin mfma_round.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// D[32][32] = A[32][K] B[K][32] + C, K = 16 * steps; A row-major [32][K], B as [K][32]; one wave.
__global__ void mm(const _Float16* A, const _Float16* B, const float* C, float* D, int steps) {
    const int lane = threadIdx.x, c = lane & 31, hh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + c];
    for (int s = 0; s < steps; ++s) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A[c * 16 * steps + 16 * s + 8 * hh + j];
            b[j] = B[(16 * s + 8 * hh + j) * 32 + c];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + c] = acc[r];
}

int main() {
    const int steps = 8, K = 16 * steps;
    std::vector<_Float16> A(32 * K), B(K * 32);
    std::vector<float> C(1024), D(1024);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    auto run = [&](int st) {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, st);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    };
    auto clear = [&] { for (auto& x : A) x = (_Float16)0.f; for (auto& x : B) x = (_Float16)0.f; for (auto& x : C) x = 1.0f; };
    // directed cases, one MFMA (steps = 1, row 0 / col 0 inspected), accumulator 1.0, ulp(1) = 2^-23
    struct Case { const char* name; int n; float a, b; float extra_a, extra_b; double exact; };
    const float p13 = ldexpf(1.f, -13), p12 = ldexpf(1.f, -12);
    Case cases[] = {
        {"16 products of 2^-25 (sum 2^-21 = 4 ulp)", 16, p13, p12, 0, 0, 1.0 + ldexp(1.0, -21)},
        {"4 products of 2^-25 (sum 2^-23 = 1 ulp)", 4, p13, p12, 0, 0, 1.0 + ldexp(1.0, -23)},
        {"2 products of 2^-25 (sum = half ulp: tie)", 2, p13, p12, 0, 0, 1.0 + ldexp(1.0, -24)},
        {"3 products of 2^-25 (0.75 ulp)", 3, p13, p12, 0, 0, 1.0 + 3 * ldexp(1.0, -25)},
        {"1 product of 2^-25 (0.25 ulp)", 1, p13, p12, 0, 0, 1.0 + ldexp(1.0, -25)},
        {"2 x 2^-25 + 2^-34 (just above the tie)", 2, p13, p12, ldexpf(1.f, -17), ldexpf(1.f, -17), 1.0 + ldexp(1.0, -24) + ldexp(1.0, -34)},
        {"1 product of -2^-26 (1 - 0.25 ulp_below)", 1, -ldexpf(1.f, -13), p13, 0, 0, 1.0 - ldexp(1.0, -26)},
        {"1 product of -2^-25 (1 - 0.5 ulp_below: tie)", 1, -p13, p12, 0, 0, 1.0 - ldexp(1.0, -25)},
    };
    for (auto& cs : cases) {
        clear();
        for (int k = 0; k < cs.n; ++k) { A[k] = (_Float16)cs.a; B[k * 32] = (_Float16)cs.b; }
        if (cs.extra_a != 0) { A[15] = (_Float16)cs.extra_a; B[15 * 32] = (_Float16)cs.extra_b; }
        A.resize(32 * 16 * 8);
        // steps = 1 uses row stride 16
        std::vector<_Float16> A1(32 * 16, (_Float16)0.f);
        for (int k = 0; k < 16; ++k) A1[k] = A[k];
        std::vector<_Float16> keep = A; A = A1; A.resize(32 * K, (_Float16)0.f); for (int k = 0; k < 16; ++k) A[k] = A1[k];
        run(1);
        const double rne = (double)(float)cs.exact;
        printf("%-50s exact-1 = %+.3f ulp   mfma-1 = %+.3f ulp   (RNE of exact: %+.3f ulp)\n", cs.name, (cs.exact - 1.0) / ldexp(1.0, -23),
               ((double)D[0] - 1.0) / ldexp(1.0, -23), (rne - 1.0) / ldexp(1.0, -23));
        A = keep;
    }
    // statistics: K = 128, values like the kernels' (activations ~N(0,1) * 64 as fp16, weights ~U * 24), C = bias-like
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    double se_m = 0, se_seq = 0, se_pair = 0, bias_m = 0, bias_seq = 0, cnt = 0;
    for (int rep = 0; rep < 50; ++rep) {
        for (auto& x : A) x = (_Float16)(nd(rng) * 16.f);
        for (auto& x : B) x = (_Float16)(ud(rng) * 24.f);
        for (auto& x : C) x = nd(rng) * 100.f;
        run(steps);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ex = C[i * 32 + j];
                float seq = C[i * 32 + j];
                float part[16];
                for (int k = 0; k < K; ++k) {
                    const float a = (float)A[i * K + k], b = (float)B[k * 32 + j];
                    ex += (double)a * (double)b;
                    seq = fmaf(a, b, seq);
                }
                // blocked sum like a BLAS micro-kernel: 16 independent partial sums, then a tree
                for (int l = 0; l < 16; ++l) part[l] = 0.f;
                for (int k = 0; k < K; ++k) part[k & 15] = fmaf((float)A[i * K + k], (float)B[k * 32 + j], part[k & 15]);
                for (int w = 8; w >= 1; w >>= 1) for (int l = 0; l < w; ++l) part[l] += part[l + w];
                const float pair = part[0] + C[i * 32 + j];
                // scale: ulp of the result magnitude ~ use |ex| floor at the typical sum magnitude
                const double u = ldexp(1.0, -23) * fmax(fabs(ex), 1e-30);
                const double em = ((double)D[i * 32 + j] - ex) / u, es = ((double)seq - ex) / u, ep = ((double)pair - ex) / u;
                se_m += em * em; se_seq += es * es; se_pair += ep * ep; bias_m += em * (ex > 0 ? 1 : -1); bias_seq += es * (ex > 0 ? 1 : -1); cnt += 1;
            }
    }
    printf("K = 128 dot + C, relative error in units of 2^-23 |result|:  mfma chain rms %.3f (signed toward-magnitude bias %+.3f)   fmaf chain rms %.3f (bias %+.3f)   blocked fp32 rms %.3f\n",
           sqrt(se_m / cnt), bias_m / cnt, sqrt(se_seq / cnt), bias_seq / cnt, sqrt(se_pair / cnt));
    return 0;
}
