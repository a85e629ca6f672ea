// kbench13.hip -- one LSTM forward wavefront level as production runs it (layer 0: K = 200 + 512 folded input projection; layer 1: K = 512 + 512)
// as the multi-problem fused-cell launch, for several tile configurations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static const int B = 512, R = 512, T = 26;
static float *dW, *dH, *dC, *dG, *dU;
static int *dN, *dSI;
template <class C> float level(int iters)
{
    MultiArgs<EpiLstmFwd> ma;
    for (int p = 0; p < 2; ++p) {
        EpiLstmFwd e{};
        e.gx = dG + (size_t)p * B * 4 * R; e.c_prev = dC; e.c = dC + B * R; e.h = dH + (size_t)(2 + p) * B * R; e.u_next = p == 0 ? dU : nullptr;
        e.bias1 = dW; e.bias2 = dW;
        e.nrows = dN; e.sort_idx = dSI; e.R = R; e.B = B; e.T = T; e.t = 3; e.lnext_m1 = 0; e.dr = Drop{1, 0.5f, 2.0f, 123, 1};
        GemmArgs g = {};
        g.A = dH; g.B = dW; g.lda = p ? R : 200; g.ldb = p ? R : 200; g.M = B; g.N = R; g.K = p ? R : 200; g.kslice = R; g.R = R; g.mlimit = dN;
        g.A2 = dH + (size_t)B * R; g.B2 = dW + (size_t)4 * R * R; g.lda2 = R; g.ldb2 = R; g.K2 = R;
        ma.g[p] = g; ma.e[p] = e;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm_multi<C, A_KC, B_KC, true, EpiLstmFwd, 1>(0, ma, 2);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm_multi<C, A_KC, B_KC, true, EpiLstmFwd, 1>(0, ma, 2);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
int main()
{
    hipMalloc(&dW, (size_t)2 * 4 * R * R * 4); hipMalloc(&dH, (size_t)4 * B * R * 4); hipMalloc(&dC, (size_t)2 * B * R * 4);
    hipMalloc(&dG, (size_t)2 * B * 4 * R * 4); hipMalloc(&dU, (size_t)B * R * 4); hipMalloc(&dN, 4); hipMalloc(&dSI, B * 4);
    std::vector<float> h((size_t)2 * 4 * R * R);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, h.data(), (size_t)4 * B * R * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, h.data(), (size_t)2 * B * R * 4, hipMemcpyHostToDevice); hipMemcpy(dG, h.data(), (size_t)2 * B * 4 * R * 4, hipMemcpyHostToDevice);
    int n = B; hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice); std::vector<int> si(B); for (int i = 0; i < B; ++i) si[i] = i;
    hipMemcpy(dSI, si.data(), B * 4, hipMemcpyHostToDevice);
    const int it = 100;
    printf("forward level: 3.2 GFLOP, floor 20.5 us at 157 TF\n");
#define L_(name, ...) printf("%-52s %7.2f us\n", name, level<Cfg<__VA_ARGS__>>(it));
    for (int rep = 0; rep < 2; ++rep) {
    L_("64x16u BK64 wm4 wk2 pf1 (prod, 512 WG, 8 waves)", 16, 64, 64, 64, 4, 1, 2, 1)
    L_("64x16u BK64 wm2 wk4 pf1 (NTM=2, 512 WG, 8 waves)", 16, 64, 64, 64, 2, 1, 4, 1)
    L_("64x16u BK64 wm2 wk2 pf1 (NTM=2, 512 WG, 4 waves)", 16, 64, 64, 64, 2, 1, 2, 1)
    L_("64x16u BK64 wm2 wk2 pf2 (NTM=2, 512 WG, 4 waves)", 16, 64, 64, 64, 2, 1, 2, 2)
    L_("64x16u BK32 wm2 wk2 pf2 (NTM=2, 512 WG, 4 waves)", 16, 64, 64, 32, 2, 1, 2, 2)
    L_("128x16u BK64 wm4 wk2 pf1 (NTM=2, 256 WG, 8 waves)", 16, 128, 64, 64, 4, 1, 2, 1)
    L_("128x16u BK64 wm4 wk2 pf2 (NTM=2, 256 WG, 8 waves)", 16, 128, 64, 64, 4, 1, 2, 2)
    L_("128x16u BK32 wm4 wk2 pf2 (NTM=2, 256 WG, 8 waves)", 16, 128, 64, 32, 4, 1, 2, 2)
    L_("128x16u BK64 wm4 wk4 pf1 (NTM=2, 256 WG, 16 waves)", 16, 128, 64, 64, 4, 1, 4, 1)
    L_("128x16u BK64 wm2 wk4 pf1 (NTM=4, 256 WG, 8 waves)", 16, 128, 64, 64, 2, 1, 4, 1)
    L_("128x16u BK64 wm8 wk1 pf1 (NTM=1, 256 WG, 8 waves)", 16, 128, 64, 64, 8, 1, 1, 1)
    L_("128x16u BK64 wm8 wk2 pf1 (NTM=1, 256 WG, 16 waves)", 16, 128, 64, 64, 8, 1, 2, 1)
    }
    return 0;
}
