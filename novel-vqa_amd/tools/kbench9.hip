// kbench9.hip -- in-kernel clock and per-workgroup cycle counts of the level kernels (s_memtime / s_memrealtime stamps); derived from kbench7: does splitting the batch into NCH
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

template <int DBG> struct ProdCfg { typedef Cfg<16, 64, 64, 64, 4, 1, 2, 1, DBG> type; };
// variant 0..2: register-staged kernel with DBG 0 (full), 1 (no global loads), 2 (no MFMA); 3: ring S = 4
template <int V>
__global__ __launch_bounds__(V == 3 ? 256 : 512) void stamped(MultiArgs<EpiLstmFwd> a, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int p = blockIdx.z;
    if constexpr (V == 3) gemm_ring_body<true, EpiLstmFwd, 1, 4>(a.g[p], a.e[p], blockIdx.x, blockIdx.y, 0);
    else gemm_f32_body<typename ProdCfg<V>::type, A_KC, B_KC, true, EpiLstmFwd, 1>(a.g[p], a.e[p], blockIdx.x, blockIdx.y, 0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        out[b * 2] = t1 - t0; out[b * 2 + 1] = r1 - r0;
    }
}
#include <algorithm>
template <int V> static void run(const char *name, unsigned long long *dS)
{
    auto ma = fwd_args(0, B);
    dim3 grid(R / 16, B / 64, 2);
    const int it = 30000; // > 1 s of back-to-back launches: the clock has settled
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(stamped<V>, grid, dim3(V == 3 ? 256 : 512), 0, 0, ma, dS);
    hipEventRecord(e0, 0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(stamped<V>, grid, dim3(V == 3 ? 256 : 512), 0, 0, ma, dS);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); return; }
    std::vector<unsigned long long> s(1024 * 2);
    hipMemcpy(s.data(), dS, s.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc[2];
    for (int b = 0; b < 1024; ++b) {
        if (!s[b * 2 + 1]) continue;
        clk.push_back((double)s[b * 2] / (double)s[b * 2 + 1] * 100.0); // MHz (s_memrealtime ticks at 100 MHz)
        cyc[b / 256].push_back((double)s[b * 2]);
    }
    if (clk.empty()) { printf("%s: no stamps (launch error: %s)\n", name, hipGetErrorString(hipGetLastError())); return; }
    std::sort(clk.begin(), clk.end());
    for (auto &c : cyc) std::sort(c.begin(), c.end());
    printf("%-34s %6.2f us/level  clock median %5.0f MHz (p10 %5.0f, p90 %5.0f)  WG cycles median: layer-0 step %6.0f, layer-1 step %6.0f\n", name,
           ms * 1e3 / it, clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], cyc[0][cyc[0].size() / 2], cyc[1][cyc[1].size() / 2]);
}
int main()
{
    hipMalloc(&dW, (size_t)2 * 4 * R * R * 4); hipMalloc(&dH, (size_t)4 * B * R * 4); hipMalloc(&dC, (size_t)2 * B * R * 4);
    hipMalloc(&dG, (size_t)2 * B * 4 * R * 4); hipMalloc(&dU, (size_t)B * R * 4); hipMalloc(&dN, 4); hipMalloc(&dSI, B * 4);
    std::vector<float> h((size_t)2 * 4 * R * R);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 9) * (1.0f / 8388608.f) * 0.2f - 0.1f;
    hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, h.data(), (size_t)4 * B * R * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, h.data(), (size_t)2 * B * R * 4, hipMemcpyHostToDevice); hipMemcpy(dG, h.data(), (size_t)2 * B * 4 * R * 4, hipMemcpyHostToDevice);
    int n = B; hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice); std::vector<int> si(B); for (int i = 0; i < B; ++i) si[i] = i;
    hipMemcpy(dSI, si.data(), B * 4, hipMemcpyHostToDevice);
    unsigned long long *dS; hipMalloc(&dS, 1024 * 2 * 8); hipMemset(dS, 0, 1024 * 2 * 8);
    run<0>("register-staged, full", dS);
    run<1>("register-staged, no global loads", dS);
    run<2>("register-staged, no MFMA", dS);
    run<3>("LDS-DMA ring S=4", dS);
    return 0;
}
