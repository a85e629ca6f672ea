// wgrad_bf16_check.hip -- TEST INFRASTRUCTURE (tests/test_gpu_wgrad_bf16.py): the bf16 k-major weight-gradient kernel of the
// library (novel-vqa_amd/csrc/wgrad_bf16.h, compiled into this program as the library compiles it) on the shapes of the
// step -- dW[2048 x N] = dG^T X, K = 14336, N = 512 / 200, operands as f32 rows or as the persistent kernels' bf16 images,
// split-K 8 and 16 -- against a DOUBLE-PRECISION sum of the bf16-rounded operands on the host (its own statement of the
// product: round-to-nearest-even to bf16, exact products, f64 accumulation).  Every output element of a sample of rows and
// ALL their columns is checked; prints one line per configuration and exits non-zero on a mismatch.
// (Promoted from tools/kbench15, which only timed the kernel and sampled 200 outputs.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../novel-vqa_amd/csrc/wgrad_bf16.h"
using namespace nvqa;

static float bf16r(float x)
{
    unsigned u; memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float y; memcpy(&y, &u, 4);
    return y;
}

__global__ void k_sum_slabs(const float *slabs, int ks, size_t n, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < ks; ++z) s += slabs[(size_t)z * n + i];
    out[i] = s;
}

int main()
{
    const int M = 2048, K = 14336, lda = 2048;
    std::vector<float> hA((size_t)K * lda), hB((size_t)K * 512);
    srand(1);
    for (auto &v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto &v : hB) v = (rand() / (float)RAND_MAX - 0.5f);
    float *dA, *dB, *dS, *dO;
    if (hipMalloc(&dA, hA.size() * 4) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMalloc(&dB, hB.size() * 4);
    (void)hipMalloc(&dS, (size_t)16 * M * 512 * 4); (void)hipMalloc(&dO, (size_t)M * 512 * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    std::vector<unsigned short> hA16(hA.size()), hB16(hB.size());
    std::vector<float> rA(hA.size()), rB(hB.size()); // the rounded operands, as f32
    for (size_t i = 0; i < hA.size(); ++i) { rA[i] = bf16r(hA[i]); unsigned u; memcpy(&u, &rA[i], 4); hA16[i] = (unsigned short)(u >> 16); }
    for (size_t i = 0; i < hB.size(); ++i) { rB[i] = bf16r(hB[i]); unsigned u; memcpy(&u, &rB[i], 4); hB16[i] = (unsigned short)(u >> 16); }
    unsigned short *dA16, *dB16;
    (void)hipMalloc(&dA16, hA16.size() * 2); (void)hipMalloc(&dB16, hB16.size() * 2);
    (void)hipMemcpy(dA16, hA16.data(), hA16.size() * 2, hipMemcpyHostToDevice);
    int bad = 0;
    for (int N : {512, 200}) {
        const int ldb = N;
        // B with row stride N: repack the host matrix (the first N columns of each 512-wide row) and its image
        std::vector<float> hBn((size_t)K * N), rBn((size_t)K * N);
        std::vector<unsigned short> hBn16((size_t)K * N);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < N; ++n) { hBn[(size_t)k * N + n] = hB[(size_t)k * 512 + n]; rBn[(size_t)k * N + n] = rB[(size_t)k * 512 + n]; hBn16[(size_t)k * N + n] = hB16[(size_t)k * 512 + n]; }
        (void)hipMemcpy(dB, hBn.data(), hBn.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB16, hBn16.data(), hBn16.size() * 2, hipMemcpyHostToDevice);
        // reference: 24 rows of dW, all N columns, f64 accumulation of exact products of the rounded operands
        const int NR = 24;
        std::vector<int> rows(NR);
        for (int r = 0; r < NR; ++r) rows[r] = (r * 977 + 13) % M;
        std::vector<double> ref((size_t)NR * N, 0.0);
        for (int k = 0; k < K; ++k)
            for (int r = 0; r < NR; ++r) {
                const double a = rA[(size_t)k * lda + rows[r]];
                const float *b = &rBn[(size_t)k * N];
                double *o = &ref[(size_t)r * N];
                for (int n = 0; n < N; ++n) o[n] += a * (double)b[n];
            }
        double scale = 0;
        for (double v : ref) scale = fmax(scale, fabs(v));
        for (int mode = 0; mode < 3; ++mode)
            for (int ks : {8, 16}) {
                const int kslice = ((K + ks - 1) / ks + 63) / 64 * 64;
                const int kz = (K + kslice - 1) / kslice;
                WgradBf16Args g{dA, dB, dA16, dB16, dS, (size_t)M * N, lda, ldb, N, M, N, K, kslice};
                const dim3 grid((M + 127) / 128, (N + 127) / 128, kz);
                (void)hipMemset(dS, 0xff, (size_t)16 * M * 512 * 4); // NaNs: an unwritten slab element shows
                if (mode == 0) hipLaunchKernelGGL((k_wgrad_bf16<false, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                else if (mode == 1) hipLaunchKernelGGL((k_wgrad_bf16<true, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                else hipLaunchKernelGGL((k_wgrad_bf16<true, true>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                hipLaunchKernelGGL(k_sum_slabs, dim3(((size_t)M * N + 255) / 256), dim3(256), 0, 0, dS, kz, (size_t)M * N, dO);
                std::vector<float> hO((size_t)M * N);
                if (hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 3; }
                double worst = 0;
                for (int r = 0; r < NR; ++r)
                    for (int n = 0; n < N; ++n) {
                        const double e = fabs(ref[(size_t)r * N + n] - (double)hO[(size_t)rows[r] * N + n]);
                        worst = (e == e) ? fmax(worst, e) : 1e30;
                    }
                const bool ok = worst <= 2e-5 * scale; // f32 accumulation of 14336 exact products, split-K 8 / 16: measured 2-4e-6
                bad += ok ? 0 : 1;
                printf("A %s B %s N=%d splitK=%d: max err %.3g of %.3g (%.2g relative) %s\n", mode ? "image" : "f32", mode == 2 ? "image" : "f32", N, kz,
                       worst, scale, worst / scale, ok ? "OK" : "MISMATCH");
            }
    }
    return bad ? 1 : 0;
}
