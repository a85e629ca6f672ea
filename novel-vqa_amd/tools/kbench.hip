// kbench.hip -- ablation microbenchmark of the per-step LSTM kernels (not part of libnvqa).
// Times the forward / backward step GEMM (+fused cell) at B=512, R=512 with pieces removed:
//   DBG 0 full, 1 no global loads, 2 no MFMA loop, 4 no epilogue (and combinations).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../include tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../csrc/gemm_f32.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static const int B = 512, R = 512, T = 26;
static float *dW, *dH, *dC, *dG, *dU, *dDC;
static int *dN, *dSI;

template <class C> float time_fwd(int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    EpiLstmFwd e{}; e.gx = dG; e.c_prev = dC; e.c = dC + B * R; e.h = dH + B * R; e.u_next = dU; e.nrows = dN; e.sort_idx = dSI;
    e.R = R; e.B = B; e.T = T; e.t = 3; e.lnext_m1 = 0; e.dr = Drop{1, 0.5f, 2.0f, 123, 1};
    GemmArgs g{dH, dW, R, R, B, R, R, R, R, dN};
    for (int i = 0; i < 3; ++i) launch_gemm<C, A_KC, B_KC, true, EpiLstmFwd>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm<C, A_KC, B_KC, true, EpiLstmFwd>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
template <class C> float time_bwd(int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    EpiLstmBwd e{}; e.gates = dG; e.c_prev = dC; e.c = dC + B * R; e.dc = dDC; e.dh_ext = dU; e.dh_ext2 = nullptr; e.nrows = dN; e.R = R;
    GemmArgs g{dG + (size_t)B * 4 * R, dW, 4 * R, R, B, R, 4 * R, 4 * R, 0, dN};
    for (int i = 0; i < 3; ++i) launch_gemm<C, A_KC, B_NC, false, EpiLstmBwd>(0, g, e);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_gemm<C, A_KC, B_NC, false, EpiLstmBwd>(0, g, e);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    CK(hipMalloc(&dW, 4 * R * R * 4)); CK(hipMalloc(&dH, 2 * B * R * 4)); CK(hipMalloc(&dC, 2 * B * R * 4));
    CK(hipMalloc(&dG, 2 * (size_t)B * 4 * R * 4)); CK(hipMalloc(&dU, B * R * 4)); CK(hipMalloc(&dDC, B * R * 4));
    CK(hipMalloc(&dN, 4)); CK(hipMalloc(&dSI, B * 4));
    std::vector<float> h(2 * (size_t)B * 4 * R, 0.01f);
    CK(hipMemcpy(dW, h.data(), 4 * R * R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, h.data(), 2 * B * R * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, h.data(), 2 * B * R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dG, h.data(), 2 * (size_t)B * 4 * R * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, h.data(), B * R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dDC, h.data(), B * R * 4, hipMemcpyHostToDevice));
    int n = B; CK(hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice)); std::vector<int> si(B); for (int i = 0; i < B; ++i) si[i] = i;
    CK(hipMemcpy(dSI, si.data(), B * 4, hipMemcpyHostToDevice));
    if (argc > 1) { // counter runs: production configs only
        printf("fwd prod %7.2f us\n", time_fwd<Cfg<16, 32, 128, 32, 2, 2, 2, 2, 0>>(20));
        printf("bwd prod %7.2f us\n", time_bwd<Cfg<16, 32, 32, 128, 2, 2, 4, 2, 0>>(20));
        return 0;
    }
    const int it = 200;
    printf("ideal MFMA time per step: 6.8 us (1.07 GFLOP at 157.3 TF)\n");
#define F(name, ...) printf("fwd %-34s %7.2f us\n", name, time_fwd<Cfg<__VA_ARGS__>>(it));
#define Bw(name, ...) printf("bwd %-34s %7.2f us\n", name, time_bwd<Cfg<__VA_ARGS__>>(it));
    F("mf16 32x128x32 wk2 pf2 (prod)", 16, 32, 128, 32, 2, 2, 2, 2, 0)
    F("mf16 32x128x64 wk2 pf2", 16, 32, 128, 64, 2, 2, 2, 2, 0)
    F("mf16 32x128x64 wk4 pf2", 16, 32, 128, 64, 2, 2, 4, 2, 0)
    F("mf32 32x128x32 wk4 pf2", 32, 32, 128, 32, 1, 1, 4, 2, 0)
    F("mf32 32x128x64 wk4 pf2", 32, 32, 128, 64, 1, 1, 4, 2, 0)
    F("mf32 32x128x64 wk8 pf2", 32, 32, 128, 64, 1, 1, 8, 2, 0)
    F("mf32 64x128x32 wk4 pf2 (128blk)", 32, 64, 128, 32, 2, 1, 4, 2, 0)
    Bw("mf16 32x32x128 wk4 pf2 (prod)", 16, 32, 32, 128, 2, 2, 4, 2, 0)
    Bw("mf16 32x32x128 wk4 pf1", 16, 32, 32, 128, 2, 2, 4, 1, 0)
    Bw("mf32 32x32x128 wk8 pf2", 32, 32, 32, 128, 1, 1, 8, 2, 0)
    Bw("mf32 32x32x128 wk16 pf2", 32, 32, 32, 128, 1, 1, 16, 2, 0)
    Bw("mf32 32x32x64 wk8 pf2", 32, 32, 32, 64, 1, 1, 8, 2, 0)
    Bw("mf32 64x32x64 wk8 pf2 (128blk)", 32, 64, 32, 64, 2, 1, 8, 2, 0)
    Bw("mf32 64x64x64 wk4 pf2 (64blk)", 32, 64, 64, 64, 2, 2, 4, 2, 0)
    return 0;
}
