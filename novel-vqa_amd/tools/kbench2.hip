// kbench2.hip -- would cross-CU split-K (bigger tiles, partial slabs, separate finisher) beat the fused
// small-tile step kernels?  Times one wavefront level as a multi-problem plain GEMM into slabs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static const int B = 512, R = 512;
template <class C, int AM, int BMo> float run(int nprob, int M, int N, int K, int Z, float *A, float *Bm, float *slabs, int iters)
{
    MultiArgs<EpiStore> ma;
    for (int p = 0; p < nprob; ++p) {
        GemmArgs g = {};
        g.A = A; g.B = Bm; g.lda = K; g.ldb = BMo == B_KC ? K : N; g.M = M; g.N = N; g.K = K; g.kslice = K / Z;
        ma.g[p] = g;
        ma.e[p] = EpiStore{slabs + (size_t)p * Z * M * N, N, (size_t)M * N};
    }
    ma.zsplit = Z;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm_multi<C, AM, BMo, false, EpiStore, 0>(0, ma, nprob);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm_multi<C, AM, BMo, false, EpiStore, 0>(0, ma, nprob);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
int main()
{
    float *A, *W, *S;
    hipMalloc(&A, (size_t)B * 4 * R * 4); hipMalloc(&W, (size_t)4 * R * R * 4); hipMalloc(&S, (size_t)3 * 8 * B * 4 * R * 4);
    std::vector<float> h((size_t)B * 4 * R);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), (size_t)4 * R * R * 4, hipMemcpyHostToDevice);
    const int it = 100;
    printf("bwd level = 3 products of 512x512x2048 (3.2 GFLOP, MFMA floor 20.5 us); fused small-tile kernel today: 53 us\n");
    printf("  64x64x32 mf32 wk1 pf2 Z=4 (prod)     %7.2f us\n", run<Cfg<32, 64, 64, 32, 2, 2, 1, 2>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 wk2 pf1 Z=4 (8 waves)  %7.2f us\n", run<Cfg<16, 64, 64, 32, 2, 2, 2, 1>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 wk1 pf1 Z=4 (4 waves)  %7.2f us\n", run<Cfg<16, 64, 64, 32, 2, 2, 1, 1>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 4x2 wk1 pf1 Z=4 (8 w)  %7.2f us\n", run<Cfg<16, 64, 64, 32, 4, 2, 1, 1>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 4x4 wk1 pf1 Z=4 (16 w) %7.2f us\n", run<Cfg<16, 64, 64, 32, 4, 4, 1, 1>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x64 mf16 wk4 pf1 Z=4 (16 waves) %7.2f us\n", run<Cfg<16, 64, 64, 64, 2, 2, 4, 1>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 wk2 pf2 Z=4 (8 waves)  %7.2f us\n", run<Cfg<16, 64, 64, 32, 2, 2, 2, 2>, A_KC, B_NC>(3, B, R, 4 * R, 4, A, W, S, it));
    printf("  64x64x32 mf16 wk2 pf1 Z=2 (384 blk)  %7.2f us\n", run<Cfg<16, 64, 64, 32, 2, 2, 2, 1>, A_KC, B_NC>(3, B, R, 4 * R, 2, A, W, S, it));
    printf("  64x64x32 mf16 wk2 pf1 Z=8 (1536 blk) %7.2f us\n", run<Cfg<16, 64, 64, 32, 2, 2, 2, 1>, A_KC, B_NC>(3, B, R, 4 * R, 8, A, W, S, it));
    printf("  128x64x32 mf16 wk2 pf1 Z=8 (8 waves) %7.2f us\n", run<Cfg<16, 128, 64, 32, 2, 2, 2, 1>, A_KC, B_NC>(3, B, R, 4 * R, 8, A, W, S, it));
    printf("  128x64x32 mf16 4x2 pf1 Z=8 (8 waves) %7.2f us\n", run<Cfg<16, 128, 64, 32, 4, 2, 1, 1>, A_KC, B_NC>(3, B, R, 4 * R, 8, A, W, S, it));
    printf("  64x128x32 mf16 wk2 pf1 Z=8 (8 waves) %7.2f us\n", run<Cfg<16, 64, 128, 32, 2, 2, 2, 1>, A_KC, B_NC>(3, B, R, 4 * R, 8, A, W, S, it));
    printf("fwd level = 3 products of 512x2048x512 (3.2 GFLOP); fused small-tile kernel today: 40 us\n");
    printf("  64x128x32 mf32 wk1 pf2 Z=1 (384 blk) %7.2f us\n", run<Cfg<32, 64, 128, 32, 2, 2, 1, 2>, A_KC, B_KC>(3, B, 4 * R, R, 1, A, W, S, it));
    printf("  64x128x64 mf32 wk2 pf2 Z=1 (384 blk) %7.2f us\n", run<Cfg<32, 64, 128, 64, 2, 2, 2, 2>, A_KC, B_KC>(3, B, 4 * R, R, 1, A, W, S, it));
    printf("  64x128x32 mf32 wk1 pf2 Z=2 (768 blk) %7.2f us\n", run<Cfg<32, 64, 128, 32, 2, 2, 1, 2>, A_KC, B_KC>(3, B, 4 * R, R, 2, A, W, S, it));
    printf("  128x128x32 mf32 wk1 pf1 Z=2 (384 blk)%7.2f us\n", run<Cfg<32, 128, 128, 32, 2, 2, 1, 1>, A_KC, B_KC>(3, B, 4 * R, R, 2, A, W, S, it));
    printf("  128x128x32 mf32 wk1 pf1 Z=4 (768 blk)%7.2f us\n", run<Cfg<32, 128, 128, 32, 2, 2, 1, 1>, A_KC, B_KC>(3, B, 4 * R, R, 4, A, W, S, it));
    printf("  64x64x32 mf32 wk1 pf2 Z=1 (768 blk)  %7.2f us\n", run<Cfg<32, 64, 64, 32, 2, 2, 1, 2>, A_KC, B_KC>(3, B, 4 * R, R, 1, A, W, S, it));
    return 0;
}
