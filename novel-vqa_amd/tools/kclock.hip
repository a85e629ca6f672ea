// kclock.hip -- what clock does the chip hold under the fp32 MFMA GEMM?  One long wgrad-form GEMM
// (random operands); run under  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
int main()
{
    const int M = 8192, N = 8192, K = 4096; // 550 GFLOP: ~4.5 ms
    float *A, *B, *C;
    hipMalloc(&A, (size_t)K * M * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)K * M);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.0f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    GemmArgs g = {};
    g.A = A; g.B = B; g.lda = M; g.ldb = N; g.M = M; g.N = N; g.K = K; g.kslice = K;
    EpiStore e{C, N, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 40; ++i) launch_gemm<Cfg<32, 128, 128, 32, 2, 2, 1, 1>, A_MC, B_NC, false, EpiStore>(0, g, e); // ~0.2 s warm
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) launch_gemm<Cfg<32, 128, 128, 32, 2, 2, 1, 1>, A_MC, B_NC, false, EpiStore>(0, g, e);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("MC/NC 128x128x32 mf32: %.3f ms per GEMM, %.1f TF\n", ms / 10, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    }
    for (int i = 0; i < 10; ++i) launch_gemm<Cfg<16, 128, 128, 32, 4, 2, 1, 1>, A_MC, B_NC, false, EpiStore>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch_gemm<Cfg<16, 128, 128, 32, 4, 2, 1, 1>, A_MC, B_NC, false, EpiStore>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("MC/NC 128x128x32 mf16 4x2: %.3f ms per GEMM, %.1f TF\n", ms / 10, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
