// kbench10.hip -- tile configuration sweep for the M = B (head) products and the d(input) product of the step:
// which of the template's configurations is fastest per (shape, operand layout).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static float *dA, *dB, *dC;
template <class C, int AM, int BMo> float run(int M, int N, int K, int iters = 50)
{
    GemmArgs g = {};
    g.A = dA; g.B = dB; g.lda = AM == A_KC ? K : M; g.ldb = BMo == B_KC ? K : N; g.M = M; g.N = N; g.K = K; g.kslice = K; g.xcd = 1;
    EpiStore e{dC, N, (size_t)M * N};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm<C, AM, BMo, false, EpiStore>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm<C, AM, BMo, false, EpiStore>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
template <int AM, int BMo> void sweep(const char *name, int M, int N, int K)
{
    const double gf = 2.0 * M * N * K / 1e3;
    printf("%-44s", name);
    float t;
#define T_(...) t = run<Cfg<__VA_ARGS__>, AM, BMo>(M, N, K); printf(" %6.1f us (%5.1f TF)", t, gf / t / 1e3);
    T_(32, 64, 64, 32, 2, 2, 1, 2)      // MED
    T_(16, 64, 64, 32, 2, 2, 2, 1)      // mf16 64x64 wk2
    T_(16, 64, 64, 32, 4, 2, 1, 1)      // mf16 64x64 4x2 (8 waves 16x32)
    T_(32, 128, 128, 32, 2, 2, 1, 1)    // BIG
    T_(16, 128, 128, 32, 4, 2, 1, 1)    // mf16 128x128 4x2
    T_(16, 128, 64, 32, 4, 2, 1, 1)     // mf16 128x64 4x2
    T_(16, 32, 64, 32, 2, 2, 1, 1)      // mf16 32x64 2x2 (4 waves 16x32)
    T_(16, 32, 64, 32, 2, 2, 2, 1)      // mf16 32x64 2x2 wk2 (8 waves)
    T_(16, 64, 32, 32, 4, 2, 1, 1)      // mf16 64x32 4x2 (8 waves 16x16)
    printf("\n");
}
int main()
{
    const size_t n = (size_t)13312 * 2048;
    hipMalloc(&dA, n * 4); hipMalloc(&dB, n * 4); hipMalloc(&dC, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 9) * (1.0f / 8388608.f) - 0.5f;
    hipMemcpy(dA, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("%-44s %-24s%-24s%-24s%-24s%-24s%-24s\n", "shape (M x N x K), layout", "MED mf32 64x64 pf2", "mf16 64x64 wk2", "mf16 64x64 4x2", "BIG mf32 128x128", "mf16 128x128 4x2", "mf16 128x64 4x2"); printf("  (+ mf16 32x64 2x2, 32x64 2x2 wk2, 64x32 4x2)\n");
    sweep<A_MC, B_NC>("dW_o  1000 x 1024 x 512   MC/NC", 1000, 1024, 512);
    sweep<A_KC, B_NC>("dzd   512 x 1024 x 1000   KC/NC", 512, 1024, 1000);
    sweep<A_MC, B_NC>("dW_q  1024 x 2048 x 512   MC/NC", 1024, 2048, 512);
    sweep<A_MC, B_NC>("dW_v  1024 x 4096 x 512   MC/NC", 1024, 4096, 512);
    sweep<A_KC, B_NC>("dqd   512 x 2048 x 1024   KC/NC", 512, 2048, 1024);
    sweep<A_KC, B_KC>("W_o   512 x 1000 x 1024   KC/KC", 512, 1000, 1024);
    sweep<A_KC, B_NC>("dX0   13312 x 200 x 2048  KC/NC", 13312, 200, 2048);
    sweep<A_KC, B_KC>("i2h   13312 x 2048 x 200  KC/KC", 13312, 2048, 200);
    return 0;
}
