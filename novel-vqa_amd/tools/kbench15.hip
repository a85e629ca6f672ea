// kbench15.hip -- the bf16 k-major weight-gradient kernel (csrc/wgrad_bf16.h) on the shapes of the step:
// dW[2048 x N] = dG^T X with K = 14336 rows, N = 512 (arch2, arch1 layer 1 / recurrent) and N = 200 (arch1 layer 0);
// checks a sample of outputs against a double-precision sum of the bf16-rounded operands, then times split-K 4 / 8 / 16.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../csrc/wgrad_bf16.h"
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
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4);
    (void)hipMalloc(&dS, (size_t)16 * M * 512 * 4); (void)hipMalloc(&dO, (size_t)M * 512 * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)k_wgrad_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES);
    // bf16 images of the operands (what the persistent bf16 kernels leave behind)
    std::vector<unsigned short> hA16(hA.size()), hB16(hB.size());
    for (size_t i = 0; i < hA.size(); ++i) { float r = bf16r(hA[i]); unsigned u; memcpy(&u, &r, 4); hA16[i] = (unsigned short)(u >> 16); }
    for (size_t i = 0; i < hB.size(); ++i) { float r = bf16r(hB[i]); unsigned u; memcpy(&u, &r, 4); hB16[i] = (unsigned short)(u >> 16); }
    unsigned short *dA16, *dB16;
    (void)hipMalloc(&dA16, hA16.size() * 2); (void)hipMalloc(&dB16, hB16.size() * 2);
    (void)hipMemcpy(dA16, hA16.data(), hA16.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB16, hB16.data(), hB16.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
    for (int N : {512, 200}) {
        const int ldb = N;
        for (int ks : {8, 16}) {
            int kslice = ((K + ks - 1) / ks + 63) / 64 * 64;
            const int kz = (K + kslice - 1) / kslice;
            WgradBf16Args g{dA, dB, dA16, dB16, dS, (size_t)M * N, lda, ldb, N, M, N, K, kslice};
            const dim3 grid((M + 127) / 128, (N + 127) / 128, kz);
            auto go = [&] {
                if (mode == 0) hipLaunchKernelGGL((k_wgrad_bf16<false, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                else if (mode == 1) hipLaunchKernelGGL((k_wgrad_bf16<true, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                else hipLaunchKernelGGL((k_wgrad_bf16<true, true>), grid, dim3(256), NVQA_WB_LDS_BYTES, 0, g);
                hipLaunchKernelGGL(k_sum_slabs, dim3(((size_t)M * N + 255) / 256), dim3(256), 0, 0, dS, kz, (size_t)M * N, dO);
            };
            go(); go();
            (void)hipEventRecord(e0, 0);
            const int it = 20;
            for (int i = 0; i < it; ++i) go();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> hO((size_t)M * N);
            (void)hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, scale = 0;
            for (int s = 0; s < 200; ++s) {
                const int m = rand() % M, n = rand() % N;
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)bf16r(hA[(size_t)k * lda + m]) * (double)bf16r(hB[(size_t)k * ldb + n]);
                worst = fmax(worst, fabs(ref - hO[(size_t)m * N + n]));
                scale = fmax(scale, fabs(ref));
            }
            printf("A %s B %s N=%d splitK=%d grid=%dx%dx%d: %.1f us (kernel + slab sum), %.0f TF; max err %.3g of %.3g %s\n", mode ? "image" : "f32", mode == 2 ? "image" : "f32", N, kz, grid.x, grid.y, grid.z,
                   ms / it * 1000, 2.0 * M * N * K / (ms / it * 1e-3) / 1e12, worst, scale, worst <= 2e-5 * scale ? "OK" : "MISMATCH");
        }
    }
    return 0;
}
