// kbench14.hip -- core loop of a weight-stationary ("persistent RNN") LSTM level: every wave keeps its slice of the
// weight matrix in REGISTERS across time steps (B fragments of v_mfma_f32_16x16x4_f32: 16 gate columns x K), the
// activations come from LDS by ds_read_b128 (one read feeds 4 MFMAs per row tile).  Question answered here: does
// hipcc keep KTOT/4 B fragments (256 VGPRs at K = 1024) resident and issue the MFMAs back to back?
// One workgroup = 4 waves (one per SIMD, 512-register budget), tile = (16*MT rows) x 64 gate columns x KTOT.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KTOT, int MT>
__global__ __launch_bounds__(256) void k_core(const float *W, const float *A, float *out, int steps)
{
    constexpr int ROWS = 16 * MT;
    __shared__ __attribute__((aligned(16))) float As[ROWS * 64]; // one K = 64 chunk of the activations, chunk-swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    // B fragments: column = 16*wave + li of this workgroup's 64, k = 64c + 16g + 4lh + w  ->  b[(c*4+g)*4+w]
    float b[KTOT / 4];
    const float *wp = W + ((size_t)blockIdx.x * 64 + 16 * wave + li) * KTOT + 4 * lh;
#pragma unroll
    for (int i = 0; i < KTOT / 16; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(wp + 16 * i);
        b[4 * i + 0] = v[0]; b[4 * i + 1] = v[1]; b[4 * i + 2] = v[2]; b[4 * i + 3] = v[3];
    }
    for (int i = tid; i < ROWS * 64; i += 256) As[i] = A[i % 4096] ;
    __syncthreads();
    float sum = 0.f;
    for (int s = 0; s < steps; ++s) {
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KTOT / 64; ++c) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    a[m] = *reinterpret_cast<const f32x4 *>(&As[(m * 16 + li) * 64 + 4 * ((4 * g + lh) ^ li)]);
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][w], b[(c * 4 + g) * 4 + w], acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) sum += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
        __syncthreads(); // stands for the step boundary
    }
    if (sum == 1234.5f) out[tid] = sum;
}

template <int KTOT, int MT> void run(const float *W, const float *A, float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 26;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_core<KTOT, MT>), dim3(256), dim3(256), 0, 0, W, A, out, steps);
    (void)hipEventRecord(e0, 0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k_core<KTOT, MT>), dim3(256), dim3(256), 0, 0, W, A, out, steps);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it / steps, fl = 256.0 * 2.0 * 16 * MT * 64 * KTOT;
    printf("K %4d, %3d rows x 64 cols per CU: %7.2f us per step  %6.1f TF  (floor %.2f us)\n", KTOT, 16 * MT, us, fl / us * 1e-6, fl / 157.3e6);
}

int main()
{
    float *W, *A, *out;
    (void)hipMalloc(&W, (size_t)256 * 64 * 1024 * 4); (void)hipMalloc(&A, (size_t)4096 * 4); (void)hipMalloc(&out, 4096);
    std::vector<float> h((size_t)256 * 64 * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f;
    (void)hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(A, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<1024, 8>(W, A, out);
    run<768, 8>(W, A, out);
    run<1024, 7>(W, A, out);
    run<512, 8>(W, A, out);
    return 0;
}
