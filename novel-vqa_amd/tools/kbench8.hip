// kbench8.hip -- operand-value sensitivity of the level kernel (derived from kbench7): does splitting the batch into NCH
// independent chains of level kernels, one HIP stream each, let one chain's launch / prologue / epilogue
// latency hide under another chain's MFMA phase?  Forward level (layer-0 step K = 512; layer-1 step
// K = 512 + 512), B = 512 rows in all, register-staged and LDS-DMA ring kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_ring.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static const int B = 512, R = 512, T = 26;
static float *dW, *dH, *dC, *dG, *dU;
static int *dN, *dSI;
static MultiArgs<EpiLstmFwd> fwd_args(int r0, int rows)
{
    MultiArgs<EpiLstmFwd> ma;
    for (int p = 0; p < 2; ++p) {
        EpiLstmFwd e{};
        e.gx = dG + (size_t)p * B * 4 * R + (size_t)r0 * 4 * R; e.c_prev = dC + (size_t)r0 * R; e.c = dC + (size_t)B * R + (size_t)r0 * R;
        e.h = dH + (size_t)(2 + p) * B * R + (size_t)r0 * R; e.u_next = p == 0 ? dU + (size_t)r0 * R : nullptr;
        e.bias1 = p ? dW : nullptr; e.bias2 = p ? dW + 4 * R : nullptr;
        e.nrows = dN; e.sort_idx = dSI + r0; e.R = R; e.B = B; e.T = T; e.t = 3; e.lnext_m1 = 0; e.dr = Drop{1, 0.5f, 2.0f, 123, 1};
        GemmArgs g = {};
        g.A = dH + (size_t)r0 * R; g.B = dW; g.lda = R; g.ldb = R; g.M = rows; g.N = R; g.K = p ? R : 0; g.kslice = R; g.R = R; g.mlimit = dN;
        g.A2 = dH + (size_t)B * R + (size_t)r0 * R; g.B2 = dW + (size_t)4 * R * R; g.lda2 = R; g.ldb2 = R; g.K2 = R;
        ma.g[p] = g; ma.e[p] = e;
    }
    return ma;
}
typedef Cfg<16, 64, 64, 64, 4, 1, 2, 1> Prod;
template <int RING> static float chains(int nch, int iters, hipStream_t *st)
{
    const int rows = B / nch;
    std::vector<MultiArgs<EpiLstmFwd>> ma;
    for (int c = 0; c < nch; ++c) ma.push_back(fwd_args(c * rows, rows));
    auto go = [&](int n) {
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < nch; ++c) {
                if (RING) launch_gemm_ring_multi<true, EpiLstmFwd, 1>(st[c], ma[c], 2);
                else launch_gemm_multi<Prod, A_KC, B_KC, true, EpiLstmFwd, 1>(st[c], ma[c], 2);
            }
    };
    go(3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEvent_t ej[8];
    hipEventRecord(e0, st[0]);
    for (int c = 1; c < nch; ++c) hipStreamWaitEvent(st[c], e0, 0);
    go(iters);
    for (int c = 1; c < nch; ++c) { hipEventCreate(&ej[c]); hipEventRecord(ej[c], st[c]); hipStreamWaitEvent(st[0], ej[c], 0); }
    hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
int main()
{
    hipMalloc(&dW, (size_t)2 * 4 * R * R * 4); hipMalloc(&dH, (size_t)4 * B * R * 4); hipMalloc(&dC, (size_t)2 * B * R * 4);
    hipMalloc(&dG, (size_t)2 * B * 4 * R * 4); hipMalloc(&dU, (size_t)B * R * 4); hipMalloc(&dN, 4); hipMalloc(&dSI, B * 4);
    int n = B; hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice); std::vector<int> si(B); for (int i = 0; i < B; ++i) si[i] = i;
    hipMemcpy(dSI, si.data(), B * 4, hipMemcpyHostToDevice);
    hipStream_t st[1]; hipStreamCreate(&st[0]);
    std::vector<float> h((size_t)2 * 4 * R * R);
    printf("is the level kernel slowed by its operand VALUES (power / clock) rather than by its loads?  us per level\n");
    for (int mode = 0; mode < 4; ++mode) {
        for (size_t i = 0; i < h.size(); ++i) {
            const float rnd = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
            h[i] = mode == 0 ? rnd : mode == 1 ? 0.f : mode == 2 ? 1.0f : (float)((i * 2654435761u) >> 9) * (1.0f / 8388608.f) - 0.5f;
        }
        hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, h.data(), (size_t)4 * B * R * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, h.data(), (size_t)2 * B * R * 4, hipMemcpyHostToDevice); hipMemcpy(dG, h.data(), (size_t)2 * B * 4 * R * 4, hipMemcpyHostToDevice);
        const char *nm[] = {"3-digit pseudo-random (kbench data)", "all zero", "all 1.0", "full-mantissa pseudo-random"};
        printf("  %-38s register-staged %7.2f us   LDS-DMA ring %7.2f us\n", nm[mode], chains<0>(1, 300, st), chains<1>(1, 300, st));
    }
    return 0;
}
