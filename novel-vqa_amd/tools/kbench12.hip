// kbench12.hip -- does a stream of global loads overlap a stream of f32 MFMAs on one CU?
// Skeleton of the forward level kernel (8 waves, 64x64x64 tiles: 32 MFMAs 16x16x4 per wave and tile, 4 float4 global
// loads per thread and tile) with each part switchable:
//   LD  : 0 none | 1 global_load_dwordx4 into registers (waited for at the end of the tile) | 2 global_load_lds (LDS-DMA)
//   FR  : 0 MFMA operands stay in registers | 1 fragments re-read from LDS (5 ds_read_b128 per 16 MFMAs)
//   ST  : 0 nothing | 1 barrier, ds_write of the loaded registers, barrier (the PF = 1 structure)
//   MF  : 0 no MFMAs | 1 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LD, int FR, int ST, int MF>
__global__ __launch_bounds__(512) void k(const float *A, const float *B, float *out, int nt, int ldk)
{
    __shared__ __attribute__((aligned(16))) float sm[2 * 64 * 64 * 2]; // two stages of (A 64x64 | B 64x64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    // each thread stages 2 float4 of A and 2 of B per tile: row = f / 16, chunk = f % 16
    const int f0 = tid, f1 = tid + 512;
    const float *pa0 = A + (size_t)(blockIdx.y * 64 + f0 / 16) * ldk + 4 * (f0 % 16);
    const float *pa1 = A + (size_t)(blockIdx.y * 64 + f1 / 16) * ldk + 4 * (f1 % 16);
    const float *pb0 = B + (size_t)(blockIdx.x * 64 + f0 / 16) * ldk + 4 * (f0 % 16);
    const float *pb1 = B + (size_t)(blockIdx.x * 64 + f1 / 16) * ldk + 4 * (f1 % 16);
    for (int i = tid; i < 2 * 64 * 64 * 2; i += 512) sm[i] = 0.001f * (i % 97);
    __syncthreads();
    float4 a4 = make_float4(0.1f * lane, 0.2f, 0.3f, 0.4f), b4[4];
    for (int j = 0; j < 4; ++j) b4[j] = make_float4(0.01f * j, 0.02f * lane, 0.03f, 0.04f);
    const int wm = wave & 3, wk = wave >> 2, li = lane & 15, lh = lane >> 4;
    typedef __attribute__((address_space(3))) void *lds_t;
    typedef const __attribute__((address_space(1))) void *glb_t;
    for (int t = 0; t < nt; ++t) {
        f32x4 r0, r1, r2, r3;
        const int k0 = (t * 64) % ldk;
        float *stage = sm + (t & 1) * (64 * 64 * 2);
        if (LD == 1) {
            r0 = *reinterpret_cast<const f32x4 *>(pa0 + k0);
            r1 = *reinterpret_cast<const f32x4 *>(pa1 + k0);
            r2 = *reinterpret_cast<const f32x4 *>(pb0 + k0);
            r3 = *reinterpret_cast<const f32x4 *>(pb1 + k0);
        } else if (LD == 2) {
            float *nxt = sm + ((t + 1) & 1) * (64 * 64 * 2);
            __builtin_amdgcn_global_load_lds((glb_t)(pa0 + k0), (lds_t)(nxt + wave * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_t)(pa1 + k0), (lds_t)(nxt + 2048 + wave * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_t)(pb0 + k0), (lds_t)(nxt + 4096 + wave * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_t)(pb1 + k0), (lds_t)(nxt + 6144 + wave * 256), 16, 0, 0);
        }
        if (MF) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (FR) {
                    const int qq = wk * 2 + q;
                    a4 = *reinterpret_cast<const float4 *>(&stage[(wm * 16 + li) * 64 + 4 * ((4 * qq + lh) ^ (li & 15))]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        b4[j] = *reinterpret_cast<const float4 *>(&stage[4096 + (j * 16 + li) * 64 + 4 * ((4 * qq + lh) ^ (li & 15))]);
                }
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float av = w == 0 ? a4.x : w == 1 ? a4.y : w == 2 ? a4.z : a4.w;
                        const float bv = w == 0 ? b4[j].x : w == 1 ? b4[j].y : w == 2 ? b4[j].z : b4[j].w;
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
                    }
            }
        }
        if (LD == 1) {
            if (ST) {
                __syncthreads();
                float *nxt = sm + ((t + 1) & 1) * (64 * 64 * 2);
                *reinterpret_cast<f32x4 *>(&nxt[4 * f0]) = r0;
                *reinterpret_cast<f32x4 *>(&nxt[4 * f1]) = r1;
                *reinterpret_cast<f32x4 *>(&nxt[4096 + 4 * f0]) = r2;
                *reinterpret_cast<f32x4 *>(&nxt[4096 + 4 * f1]) = r3;
                __syncthreads();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(r0), "v"(r1), "v"(r2), "v"(r3));
            }
        } else if (LD == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ST) __syncthreads();
        } else if (ST) {
            __syncthreads();
            __syncthreads();
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) out[tid] = s + sm[tid];
}

template <int LD, int FR, int ST, int MF> void run(const char *name, const float *A, const float *B, float *out, int gx, int gy, int nt)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LD, FR, ST, MF>), dim3(gx, gy), dim3(512), 0, 0, A, B, out, nt, 1024);
    hipEventRecord(e0, 0);
    const int it = 50;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<LD, FR, ST, MF>), dim3(gx, gy), dim3(512), 0, 0, A, B, out, nt, 1024);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it, wgs = (double)gx * gy, fl = MF ? wgs * nt * 2.0 * 64 * 64 * 64 : 0;
    printf("%-44s grid %3dx%-2d nt %3d : %8.2f us  %6.1f TF  %5.0f clk/tile/WG\n", name, gx, gy, nt, us, fl / us * 1e-6, us * 2380.0 / nt);
}

int main()
{
    float *A, *B, *out;
    hipMalloc(&A, (size_t)8192 * 1024 * 4); hipMalloc(&B, (size_t)8192 * 1024 * 4); hipMalloc(&out, 4096);
    std::vector<float> h((size_t)8192 * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int nt : {16, 64}) {
        for (int g = 0; g < 2; ++g) {
            const int gx = 32, gy = g ? 16 : 8; // 256 WGs (1 per CU) or 512 (2 per CU); A rows = gy*64 <= 8192, B rows = 2048
            run<0, 0, 0, 1>("mfma only, operands in registers", A, B, out, gx, gy, nt);
            run<0, 1, 0, 1>("mfma + LDS fragment reads", A, B, out, gx, gy, nt);
            run<0, 1, 1, 1>("mfma + frag reads + 2 barriers", A, B, out, gx, gy, nt);
            run<1, 0, 0, 0>("global loads only (regs, wait per tile)", A, B, out, gx, gy, nt);
            run<1, 0, 0, 1>("mfma(regs) + global loads (regs)", A, B, out, gx, gy, nt);
            run<1, 1, 0, 1>("mfma + frag reads + global loads", A, B, out, gx, gy, nt);
            run<1, 1, 1, 1>("full PF=1 structure (loads, ds_write, 2 bar)", A, B, out, gx, gy, nt);
            run<1, 1, 1, 0>("PF=1 structure without MFMAs", A, B, out, gx, gy, nt);
            run<2, 0, 0, 0>("LDS-DMA loads only", A, B, out, gx, gy, nt);
            run<2, 1, 1, 1>("mfma + frag reads + LDS-DMA + 1 barrier", A, B, out, gx, gy, nt);
        }
    }
    return 0;
}
