// kbench3.hip -- ablation of the time-batched GEMM (i2h shape 13312 x 2048 x 512, 27.9 GFLOP, floor 177 us)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
template <class C, int AM, int BMo> float run(int M, int N, int K, int Z, float *A, float *Bm, float *Cc, int iters)
{
    GemmArgs g = {};
    g.A = A; g.B = Bm; g.lda = AM == A_KC ? K : M; g.ldb = BMo == B_KC ? K : N; g.M = M; g.N = N; g.K = K; g.kslice = K / Z;
    EpiStore e{Cc, N, (size_t)M * N};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_gemm<C, AM, BMo, false, EpiStore>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm<C, AM, BMo, false, EpiStore>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
int main()
{
    const int M = 13312, N = 2048, K = 512;
    float *A, *W, *C;
    hipMalloc(&A, (size_t)M * 2048 * 4); hipMalloc(&W, (size_t)M * 2048 * 4); hipMalloc(&C, (size_t)8 * M * N * 4);
    std::vector<float> h((size_t)M * 2048);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {   // correctness of the DMA path against the register-staged path (same k order => bitwise equal)
        float *C2; hipMalloc(&C2, (size_t)M * N * 4);
        std::vector<float> r1((size_t)M * N), r2((size_t)M * N);
        int bad = 0;
        run<Cfg<32, 128, 128, 32, 2, 2, 1, 1, 0, 0>, A_KC, B_KC>(M, N, K, 1, A, W, C, 1); hipMemcpy(r1.data(), C, r1.size() * 4, hipMemcpyDeviceToHost);
        run<Cfg<32, 128, 128, 32, 2, 2, 1, 1, 0, 1>, A_KC, B_KC>(M, N, K, 1, A, W, C2, 1); hipMemcpy(r2.data(), C2, r2.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < r1.size(); ++i) bad += r1[i] != r2[i];
        printf("DMA vs staged KC/KC: %d mismatches of %zu\n", bad, r1.size());
        bad = 0;
        run<Cfg<32, 128, 128, 32, 2, 2, 1, 1, 0, 0>, A_MC, B_NC>(2048, 512, 13312, 1, A, W, C, 1); hipMemcpy(r1.data(), C, (size_t)2048 * 512 * 4, hipMemcpyDeviceToHost);
        run<Cfg<32, 128, 128, 32, 2, 2, 1, 1, 0, 1>, A_MC, B_NC>(2048, 512, 13312, 1, A, W, C2, 1); hipMemcpy(r2.data(), C2, (size_t)2048 * 512 * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < (size_t)2048 * 512; ++i) bad += r1[i] != r2[i];
        printf("DMA vs staged MC/NC: %d mismatches of %d\n", bad, 2048 * 512);
    }
    const int it = 20;
    const double gf = 2.0 * M * N * K / 1e3; // us * TF
#define R_(name, AM, BM_, ...) { float t = run<Cfg<__VA_ARGS__>, AM, BM_>(M, N, K, 1, A, W, C, it); printf("%-44s %8.1f us  %6.1f TF\n", name, t, gf / t / 1e3); }
    R_("KC/KC 128x128x32 pf1 (prod BIG)", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 1, 0)
    R_("   no loads", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 1, 1)
    R_("   no mfma", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 1, 2)
    R_("   no epilogue", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 1, 4)
    R_("   no loads no epilogue", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 1, 5)
    R_("KC/KC 128x128x32 mf16 wk2 pf1 (8 waves)", A_KC, B_KC, 16, 128, 128, 32, 2, 2, 2, 1, 0)
    R_("KC/KC 128x128x32 mf16 4x2 pf1 (8 waves)", A_KC, B_KC, 16, 128, 128, 32, 4, 2, 1, 1, 0)
    R_("KC/KC 128x128x32 mf16 4x4 pf1 (16 waves)", A_KC, B_KC, 16, 128, 128, 32, 4, 4, 1, 1, 0)
    R_("KC/KC 128x64x32 mf16 wk2 pf1 (8 waves)", A_KC, B_KC, 16, 128, 64, 32, 2, 2, 2, 1, 0)
    R_("KC/KC 128x64x32 mf16 4x2 pf1 (8 waves)", A_KC, B_KC, 16, 128, 64, 32, 4, 2, 1, 1, 0)
    R_("KC/KC 64x64x32 mf16 wk2 pf1 (8 waves)", A_KC, B_KC, 16, 64, 64, 32, 2, 2, 2, 1, 0)
    R_("KC/NC 128x128x32 mf16 wk2 pf1 (8 waves)", A_KC, B_NC, 16, 128, 128, 32, 2, 2, 2, 1, 0)
    R_("KC/NC 128x128x32 mf16 4x2 pf1 (8 waves)", A_KC, B_NC, 16, 128, 128, 32, 4, 2, 1, 1, 0)
    R_("MC/NC 128x128x32 mf16 wk2 pf1 (8 waves)", A_MC, B_NC, 16, 128, 128, 32, 2, 2, 2, 1, 0)
    R_("MC/NC 128x128x32 mf16 4x2 pf1 (8 waves)", A_MC, B_NC, 16, 128, 128, 32, 4, 2, 1, 1, 0)
    R_("MC/NC 128x128x32 mf32 wk2 pf1 (8 waves)", A_MC, B_NC, 32, 128, 128, 32, 2, 2, 2, 1, 0)
    R_("KC/KC 128x128x32 pf2", A_KC, B_KC, 32, 128, 128, 32, 2, 2, 1, 2, 0)
    R_("KC/KC 128x128x64 pf1", A_KC, B_KC, 32, 128, 128, 64, 2, 2, 1, 1, 0)
    R_("KC/KC 128x128x64 wk2 pf1 (8 waves)", A_KC, B_KC, 32, 128, 128, 64, 2, 2, 2, 1, 0)
    R_("KC/KC 256x128x32 pf1 (8 waves 4x2)", A_KC, B_KC, 32, 256, 128, 32, 4, 2, 1, 1, 0)
    R_("KC/KC 128x256x32 pf1 (8 waves 2x4)", A_KC, B_KC, 32, 128, 256, 32, 2, 4, 1, 1, 0)
    R_("KC/KC 64x64x32 pf2 (MED)", A_KC, B_KC, 32, 64, 64, 32, 2, 2, 1, 2, 0)
    R_("KC/KC 128x64x32 pf2", A_KC, B_KC, 32, 128, 64, 32, 2, 2, 1, 2, 0)
    R_("KC/NC 128x128x32 pf1 (dgrad form)", A_KC, B_NC, 32, 128, 128, 32, 2, 2, 1, 1, 0)
    R_("MC/NC 128x128x32 pf1 (wgrad form)", A_MC, B_NC, 32, 128, 128, 32, 2, 2, 1, 1, 0)
    R_("MC/NC 128x128x32 pf2", A_MC, B_NC, 32, 128, 128, 32, 2, 2, 1, 2, 0)
    return 0;
}
