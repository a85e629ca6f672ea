// gemm_f32.h -- LDS-tiled fp32 GEMM on the gfx950 f32-input MFMA
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32 fma chains at the
// f32 vector rate, MI355X_MICROARCH.md "Matrix cores").  Every dense product of the
// VQA step runs through this template; what differs per call site is the operand
// layout, the tile shape and the fused epilogue.
//
// Replaces the cuBLAS sgemm calls behind nn.Linear forward / updateGradInput /
// accGradParameters (reference: 002_train_vqa_arch1/misc/LSTM.lua:41-42,
// misc/netdef.lua:10-11, 002_train_baseline.lua:142,154).
//
//   C[m][n] = sum_k A(m,k) * B(k,n),  k in [z*kslice, min(K,(z+1)*kslice))   (z = blockIdx.z)
//
// Operand layouts (all row-major storage, ld in floats, ld % 4 == 0):
//   A_KC: A stored [M][K]  (k contiguous)     activations / dY for dgrad
//   A_MC: A stored [K][M]  (m contiguous)     dY^T for wgrad, without a transpose pass
//   B_KC: B stored [N][K]  (k contiguous)     Torch nn.Linear weight [out][in] in forward
//   B_NC: B stored [K][N]  (n contiguous)     the same weight in dgrad, activations in wgrad
// LDS images are k-major ([BK][BM+pad]) so that a wave's MFMA operand read is 32 (or 16)
// consecutive floats per k: conflict-free ds_read_b32.  K-contiguous sources are transposed
// on the LDS write (pad chosen so the 4 scalar writes of a float4 hit distinct banks).
//
// GATES mode (fused LSTM cell): N indexes hidden units; the block's B tile holds the
// rows {g*R + u} of the [4R][K] weight for its units u and all 4 gates g, and each lane
// ends up with the 4 gate pre-activations of one (row, unit) in its own registers, so
// the cell update needs no cross-lane traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvqa {

enum { A_KC = 0, A_MC = 1 };
enum { B_KC = 0, B_NC = 1 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *A;
    const float *B;
    int lda, ldb;
    int M, N, K;
    int kslice;        // K range per blockIdx.z (multiple of BK); >= K for no split
    int R;             // GATES: gate stride inside the [4R][K] weight
    const int *mlimit; // optional device int: rows >= *mlimit are inactive (MFMA work skipped)
};

template <int MF> struct AccT;
template <> struct AccT<32> { typedef f32x16 type; };
template <> struct AccT<16> { typedef f32x4 type; };

template <int MF>
__device__ __forceinline__ typename AccT<MF>::type mfma(float a, float b, typename AccT<MF>::type c);
template <>
__device__ __forceinline__ f32x16 mfma<32>(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma<16>(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Epilogue concept:
//   plain : void operator()(int z, int m, int n, float v) const
//   GATES : void operator()(int m, int unit, const float (&a)[4]) const
// Called only for m < M, n < N.

template <int MF, int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool GATES, class Epi>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmArgs g, Epi epi)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int KI = 64 / MF;       // k per MFMA
    constexpr int NREG = MF * MF / 64; // accumulator registers per MFMA tile
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int NTM = TM / MF, NTN = TN / MF;
    constexpr int BU = BN / 4; // GATES: units per block
    static_assert(TM % MF == 0 && TN % MF == 0, "wave tile must be a multiple of the MFMA tile");
    static_assert(!GATES || (NTN == 4 && BMODE == B_KC), "GATES: 4 gate sub-tiles per wave, weight [4R][K]");
    static_assert(BK % 4 == 0 && BK % KI == 0, "BK");
    // pad: transposing (K-contiguous) writes want LD % 8 == 1; float4 row writes want LD % 4 == 0
    constexpr int LDA = AMODE == A_KC ? BM + 1 : BM + 4;
    constexpr int LDB = BMODE == B_KC ? BN + 1 : BN + 4;
    constexpr int A_F4 = BM * BK / 4, B_F4 = BN * BK / 4;
    constexpr int NA = (A_F4 + NT - 1) / NT, NB = (B_F4 + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane % MF, lh = lane / MF;
    const int m0 = blockIdx.y * BM;
    const int n0 = GATES ? blockIdx.x * BU : blockIdx.x * BN; // GATES: first unit
    const int z = blockIdx.z;
    const int kbeg = z * g.kslice;
    const int kend = min(g.K, kbeg + g.kslice);
    const int mlim = g.mlimit ? min(g.M, *g.mlimit) : g.M;
    const bool active = m0 < mlim;

    typename AccT<MF>::type acc[NTM][NTN];
#pragma unroll
    for (int a = 0; a < NTM; ++a)
#pragma unroll
        for (int b = 0; b < NTN; ++b)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[a][b][r] = 0.0f;

    float4 ra[NA], rb[NB];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int f = tid + j * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < A_F4) {
                if (AMODE == A_KC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    const int m = m0 + row, k = k0 + 4 * kq;
                    if (m < mlim && k < kend) v = *reinterpret_cast<const float4 *>(g.A + (size_t)m * g.lda + k);
                } else {
                    const int kr = f / (BM / 4), mq = f % (BM / 4);
                    const int k = k0 + kr, m = m0 + 4 * mq;
                    if (k < kend && m < g.M) v = *reinterpret_cast<const float4 *>(g.A + (size_t)k * g.lda + m);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int f = tid + j * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < B_F4) {
                if (BMODE == B_KC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    const int k = k0 + 4 * kq;
                    int n;
                    bool ok;
                    if (GATES) {
                        const int u = n0 + row % BU;
                        n = (row / BU) * g.R + u;
                        ok = u < g.N;
                    } else {
                        n = n0 + row;
                        ok = n < g.N;
                    }
                    if (ok && k < kend) v = *reinterpret_cast<const float4 *>(g.B + (size_t)n * g.ldb + k);
                } else {
                    const int kr = f / (BN / 4), nq = f % (BN / 4);
                    const int k = k0 + kr, n = n0 + 4 * nq;
                    if (k < kend && n < g.N) v = *reinterpret_cast<const float4 *>(g.B + (size_t)k * g.ldb + n);
                }
            }
            rb[j] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int f = tid + j * NT;
            if (f < A_F4) {
                if (AMODE == A_KC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    As[(4 * kq + 0) * LDA + row] = ra[j].x;
                    As[(4 * kq + 1) * LDA + row] = ra[j].y;
                    As[(4 * kq + 2) * LDA + row] = ra[j].z;
                    As[(4 * kq + 3) * LDA + row] = ra[j].w;
                } else {
                    const int kr = f / (BM / 4), mq = f % (BM / 4);
                    *reinterpret_cast<float4 *>(&As[kr * LDA + 4 * mq]) = ra[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int f = tid + j * NT;
            if (f < B_F4) {
                if (BMODE == B_KC) {
                    const int row = f / (BK / 4), kq = f % (BK / 4);
                    Bs[(4 * kq + 0) * LDB + row] = rb[j].x;
                    Bs[(4 * kq + 1) * LDB + row] = rb[j].y;
                    Bs[(4 * kq + 2) * LDB + row] = rb[j].z;
                    Bs[(4 * kq + 3) * LDB + row] = rb[j].w;
                } else {
                    const int kr = f / (BN / 4), nq = f % (BN / 4);
                    *reinterpret_cast<float4 *>(&Bs[kr * LDB + 4 * nq]) = rb[j];
                }
            }
        }
    };

    if (active && kbeg < kend) {
        load_tiles(kbeg);
        store_tiles();
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = k0 + BK < kend;
            if (more) load_tiles(k0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK / KI; ++kk) {
                float a[NTM], b[NTN];
                const int krow = kk * KI + lh;
#pragma unroll
                for (int t = 0; t < NTM; ++t) a[t] = As[krow * LDA + wm * TM + t * MF + li];
#pragma unroll
                for (int t = 0; t < NTN; ++t) {
                    const int nl = GATES ? t * BU + wn * MF + li : wn * TN + t * MF + li;
                    b[t] = Bs[krow * LDB + nl];
                }
#pragma unroll
                for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTN; ++tb) acc[ta][tb] = mfma<MF>(a[ta], b[tb], acc[ta][tb]);
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    // epilogue: C/D map (cdna_hip_programming.md section 3): col = lane % MF,
    // row = 8*(reg>>2) + 4*(lane / MF) + (reg & 3)
#pragma unroll
    for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const int m = m0 + wm * TM + ta * MF + 8 * (r >> 2) + 4 * lh + (r & 3);
            if (m >= g.M) continue;
            if constexpr (GATES) {
                const int u = n0 + wn * MF + li;
                if (u < g.N) {
                    const float av[4] = {acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]};
                    epi(m, u, av);
                }
            } else {
#pragma unroll
                for (int tb = 0; tb < NTN; ++tb) {
                    const int n = n0 + wn * TN + tb * MF + li;
                    if (n < g.N) epi(z, m, n, acc[ta][tb][r]);
                }
            }
        }
}

template <int MF, int BM, int BN, int BK, int WM, int WN, int AMODE, int BMODE, bool GATES, class Epi>
inline hipError_t launch_gemm(hipStream_t s, const GemmArgs &g, const Epi &epi)
{
    const int ksplit = (g.K + g.kslice - 1) / g.kslice;
    dim3 grid(GATES ? (g.N + BN / 4 - 1) / (BN / 4) : (g.N + BN - 1) / BN, (g.M + BM - 1) / BM,
              ksplit < 1 ? 1 : ksplit);
    hipLaunchKernelGGL((gemm_f32_kernel<MF, BM, BN, BK, WM, WN, AMODE, BMODE, GATES, Epi>), grid,
                       dim3(64 * WM * WN), 0, s, g, epi);
    return hipGetLastError();
}

} // namespace nvqa
