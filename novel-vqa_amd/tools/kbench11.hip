// kbench11.hip -- the time-batched weight gradients (dW = dG^T X, MC/NC operands, K = T*B = 13312): tile
// configuration x split-K sweep (GEMM into slabs only; the slab sum costs ~6 us per 8 x 4 MB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static float *dA, *dB, *dC;
template <class C> float run(int M, int N, int K, int Z, int iters = 20)
{
    GemmArgs g = {};
    int kslice = ((K + Z - 1) / Z + 31) / 32 * 32;
    g.A = dA; g.B = dB; g.lda = M; g.ldb = N; g.M = M; g.N = N; g.K = K; g.kslice = kslice; g.xcd = 1;
    EpiStore e{dC, N, (size_t)M * N};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_gemm<C, A_MC, B_NC, false, EpiStore>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm<C, A_MC, B_NC, false, EpiStore>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
template <class C> void line(const char *name, int M, int N, int K)
{
    const double gf = 2.0 * M * N * K / 1e3;
    printf("  %-34s", name);
    for (int Z : {4, 8, 16}) { float t = run<C>(M, N, K, Z); printf("  Z=%-2d %6.1f us (%5.1f TF)", Z, t, gf / t / 1e3); }
    printf("\n");
}
void shape(int M, int N, int K)
{
    printf("dW %d x %d, K = %d\n", M, N, K);
    line<Cfg<32, 128, 128, 32, 2, 2, 1, 1>>("mf32 128x128x32 4w pf1 (prod)", M, N, K);
    line<Cfg<16, 128, 128, 32, 4, 2, 1, 1>>("mf16 128x128x32 8w 4x2", M, N, K);
    line<Cfg<32, 128, 128, 64, 2, 2, 1, 1>>("mf32 128x128x64 4w pf1", M, N, K);
    line<Cfg<32, 128, 128, 32, 2, 2, 2, 1>>("mf32 128x128x32 8w wk2", M, N, K);
    line<Cfg<32, 128, 64, 32, 2, 2, 1, 1>>("mf32 128x64x32 4w", M, N, K);
    line<Cfg<32, 64, 128, 32, 2, 2, 1, 1>>("mf32 64x128x32 4w", M, N, K);
    line<Cfg<16, 128, 64, 32, 4, 2, 1, 1>>("mf16 128x64x32 8w 4x2", M, N, K);
    line<Cfg<32, 256, 128, 32, 4, 2, 1, 1>>("mf32 256x128x32 8w 4x2", M, N, K);
    line<Cfg<32, 128, 256, 32, 2, 4, 1, 1>>("mf32 128x256x32 8w 2x4", M, N, K);
    line<Cfg<32, 64, 64, 32, 2, 2, 1, 1>>("mf32 64x64x32 4w", M, N, K);
    line<Cfg<16, 64, 32, 32, 4, 2, 1, 1>>("mf16 64x32x32 8w 4x2", M, N, K);
    line<Cfg<32, 128, 32, 32, 4, 1, 1, 1>>("mf32 128x32x32 4w 4x1", M, N, K);
    line<Cfg<16, 128, 32, 32, 4, 2, 1, 1>>("mf16 128x32x32 8w 4x2", M, N, K);
    line<Cfg<32, 128, 32, 64, 4, 1, 1, 1>>("mf32 128x32x64 4w 4x1", M, N, K);
}
// round 4: N = 200 leaves 22 % of a 256-wide tiling empty; tiles that span the 200 columns once (208 = 13 x 16, 224 = 7 x 32)
template <class C> void line2(const char *name, int M, int N, int K)
{
    const double gf = 2.0 * M * N * K / 1e3;
    printf("  %-34s", name);
    for (int Z : {8, 16, 32}) { float t = run<C>(M, N, K, Z); printf("  Z=%-2d %6.1f us (%5.1f TF)", Z, t, gf / t / 1e3); }
    printf("\n");
}
void shape200(int M, int N, int K)
{
    printf("dW %d x %d, K = %d (one column tile)\n", M, N, K);
    line2<Cfg<32, 128, 128, 32, 2, 2, 1, 1>>("mf32 128x128x32 4w pf1 (prod)", M, N, K);
    line2<Cfg<16, 128, 208, 32, 4, 1, 1, 1>>("mf16 128x208x32 4w 4x1", M, N, K);
    line2<Cfg<16, 128, 208, 32, 8, 1, 1, 1>>("mf16 128x208x32 8w 8x1", M, N, K);
    line2<Cfg<32, 128, 224, 32, 4, 1, 1, 1>>("mf32 128x224x32 4w 4x1", M, N, K);
    line2<Cfg<32, 64, 224, 32, 2, 1, 1, 1>>("mf32 64x224x32 2w 2x1", M, N, K);
    line2<Cfg<16, 64, 208, 32, 4, 1, 1, 1>>("mf16 64x208x32 4w 4x1", M, N, K);
    line2<Cfg<16, 128, 208, 32, 4, 1, 1, 2>>("mf16 128x208x32 4w 4x1 pf2", M, N, K);
}
int main(int argc, char **)
{
    const size_t n = (size_t)13312 * 2048;
    hipMalloc(&dA, n * 4); hipMalloc(&dB, n * 4); hipMalloc(&dC, (size_t)16 * 2048 * 512 * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 9) * (1.0f / 8388608.f) - 0.5f;
    hipMemcpy(dA, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), n * 4, hipMemcpyHostToDevice);
    if (argc > 1) { shape200(2048, 200, 13312); return 0; }
    shape(2048, 512, 13312);
    shape(2048, 200, 13312);
    return 0;
}
