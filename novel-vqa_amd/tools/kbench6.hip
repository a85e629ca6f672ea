// kbench6.hip -- the LDS-DMA ring level kernel (gemm_ring.h) against the register-staged one (gemm_f32.h):
// one LSTM forward wavefront level (layer-0 step K = 512; layer-1 step K = 512 + 512) and one BPTT level
// (3 products 512 x 512 x 2048, split-K 4), outputs compared element by element.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../csrc/gemm_ring.h"
#include "../csrc/epilogues.h"
using namespace nvqa;
static const int B = 512, R = 512, T = 26;
static float *dW, *dH, *dC, *dG0, *dG, *dU, *dOut[2];
static int *dN, *dSI;
// outputs of variant v live in dOut[v]: gates 2 x [B][4R], c 2 x [B][R], h 2 x [B][R], u [B][R]
static const size_t OUTF = (size_t)2 * B * 4 * R + 5 * (size_t)B * R;
static MultiArgs<EpiLstmFwd> fwd_args(int v)
{
    MultiArgs<EpiLstmFwd> ma;
    float *o = dOut[v];
    for (int p = 0; p < 2; ++p) {
        EpiLstmFwd e{};
        e.gx = o + (size_t)p * B * 4 * R; e.c_prev = dC; e.c = o + (size_t)2 * B * 4 * R + (size_t)p * B * R;
        e.h = o + (size_t)2 * B * 4 * R + (size_t)(2 + p) * B * R; e.u_next = p == 0 ? o + (size_t)2 * B * 4 * R + (size_t)4 * B * R : nullptr;
        e.bias1 = p ? dW : nullptr; e.bias2 = p ? dW + 4 * R : nullptr;
        e.nrows = dN; e.sort_idx = dSI; e.R = R; e.B = B; e.T = T; e.t = 3; e.lnext_m1 = 0; e.dr = Drop{1, 0.5f, 2.0f, 123, 1};
        GemmArgs g = {};
        g.A = dH; g.B = dW; g.lda = R; g.ldb = R; g.M = B; g.N = R; g.K = p ? R : 0; g.kslice = R; g.R = R; g.mlimit = dN;
        g.A2 = dH + (size_t)B * R; g.B2 = dW + (size_t)4 * R * R; g.lda2 = R; g.ldb2 = R; g.K2 = R;
        ma.g[p] = g; ma.e[p] = e;
    }
    return ma;
}
template <class F> float timeit(int iters, F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / iters;
}
static void reset_gx()
{
    for (int v = 0; v < 2; ++v) hipMemcpy(dOut[v], dG0, (size_t)2 * B * 4 * R * 4, hipMemcpyDeviceToDevice);
}
static double compare(size_t n, const char *what)
{
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), dOut[0], n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dOut[1], n * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        double d = fabs((double)a[i] - b[i]); if (d > md) md = d; if (fabs(a[i]) > mx) mx = fabs(a[i]);
        if (!(d <= 1e-5 * (1 + fabs(a[i])))) ++bad;
    }
    printf("%s: max |diff| %.3g (max |ref| %.3g), %zu of %zu beyond 1e-5\n", what, md, mx, bad, n);
    return md;
}
int main()
{
    hipMalloc(&dW, (size_t)2 * 4 * R * R * 4); hipMalloc(&dH, (size_t)4 * B * R * 4); hipMalloc(&dC, (size_t)2 * B * R * 4);
    hipMalloc(&dG0, (size_t)2 * B * 4 * R * 4); hipMalloc(&dG, (size_t)B * 4 * R * 4); hipMalloc(&dU, (size_t)B * R * 4); hipMalloc(&dN, 4); hipMalloc(&dSI, B * 4);
    for (int v = 0; v < 2; ++v) hipMalloc(&dOut[v], std::max(OUTF, (size_t)3 * 4 * B * R) * 4);
    std::vector<float> h((size_t)2 * 4 * R * R);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dH, h.data() + 777, (size_t)4 * B * R * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, h.data() + 31, (size_t)2 * B * R * 4, hipMemcpyHostToDevice); hipMemcpy(dG0, h.data() + 5, (size_t)2 * B * 4 * R * 4, hipMemcpyHostToDevice);
    hipMemcpy(dG, h.data() + 99, (size_t)B * 4 * R * 4, hipMemcpyHostToDevice);
    int n = B - 37; hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice); std::vector<int> si(B); for (int i = 0; i < B; ++i) si[i] = i;
    hipMemcpy(dSI, si.data(), B * 4, hipMemcpyHostToDevice);
    typedef Cfg<16, 64, 64, 64, 4, 1, 2, 1> Prod;
    // ---- forward level: correctness (one launch each from the same state), then timing
    reset_gx();
    { auto a0 = fwd_args(0), a1 = fwd_args(1);
      launch_gemm_multi<Prod, A_KC, B_KC, true, EpiLstmFwd, 1>(0, a0, 2);
      hipError_t e = launch_gemm_ring_multi<true, EpiLstmFwd, 1>(0, a1, 2);
      hipError_t e2 = hipDeviceSynchronize();
      printf("ring fwd launch: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
      if (e2 != hipSuccess) return 1;
      compare(OUTF, "forward level (gates, c, h, u)"); }
    n = B; hipMemcpy(dN, &n, 4, hipMemcpyHostToDevice);
    const int it = 200;
    { auto a0 = fwd_args(0), a1 = fwd_args(1);
      printf("forward level, register-staged (prod)   %7.2f us\n", timeit(it, [&] { launch_gemm_multi<Prod, A_KC, B_KC, true, EpiLstmFwd, 1>(0, a0, 2); }));
      printf("forward level, LDS-DMA ring             %7.2f us\n", timeit(it, [&] { launch_gemm_ring_multi<true, EpiLstmFwd, 1>(0, a1, 2); })); }
    // ---- BPTT level: 3 products dG [B x 4R] x W^T, W stored transposed [R][4R] for the ring (K-contiguous),
    // [4R][R] for the production kernel (B_NC); split-K 4 into slabs
    {
        float *dWt; hipMalloc(&dWt, (size_t)4 * R * R * 4);
        std::vector<float> wt((size_t)4 * R * R);
        for (int k = 0; k < 4 * R; ++k) for (int j = 0; j < R; ++j) wt[(size_t)j * 4 * R + k] = h[(size_t)k * R + j];
        hipMemcpy(dWt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice);
        const int Z = 4;
        MultiArgs<EpiStore> m0, m1;
        for (int p = 0; p < 3; ++p) {
            GemmArgs g = {};
            g.A = dG; g.lda = 4 * R; g.M = B; g.N = R; g.K = 4 * R; g.kslice = 4 * R / Z; g.mlimit = dN;
            m0.g[p] = g; m0.g[p].B = dW; m0.g[p].ldb = R;
            m1.g[p] = g; m1.g[p].B = dWt; m1.g[p].ldb = 4 * R;
            m0.e[p] = EpiStore{dOut[0] + (size_t)p * Z * B * R, R, (size_t)B * R};
            m1.e[p] = EpiStore{dOut[1] + (size_t)p * Z * B * R, R, (size_t)B * R};
        }
        m0.zsplit = m1.zsplit = Z;
        typedef Cfg<16, 64, 64, 32, 4, 2, 1, 1> ProdB;
        launch_gemm_multi<ProdB, A_KC, B_NC, false, EpiStore, 0>(0, m0, 3);
        launch_gemm_ring_multi<false, EpiStore, 0>(0, m1, 3);
        hipError_t e2 = hipDeviceSynchronize();
        printf("ring bwd launch: %s\n", hipGetErrorString(e2));
        if (e2 != hipSuccess) return 1;
        compare((size_t)3 * Z * B * R, "BPTT level slabs");
        printf("BPTT level, register-staged B_NC (prod) %7.2f us\n", timeit(it, [&] { launch_gemm_multi<ProdB, A_KC, B_NC, false, EpiStore, 0>(0, m0, 3); }));
        printf("BPTT level, LDS-DMA ring (W^T, K-contig) %6.2f us\n", timeit(it, [&] { launch_gemm_ring_multi<false, EpiStore, 0>(0, m1, 3); }));
    }
    return 0;
}
