// gemm_ring.h -- fp32 MFMA GEMM for the latency-critical LSTM wavefront levels: operands travel
// global -> LDS by direct DMA (global_load_lds_dwordx4) into a ring of S stages that runs S-1
// K-tiles ahead of the MFMAs.
//
// Why a second GEMM body: the level kernels of gemm_f32.h stage one K-tile ahead through registers.
// On a level every workgroup of the chip asks for its tile at the same moment and each XCD re-reads
// ~7 MB through the fabric, so a tile takes longer to arrive than it takes to multiply; the ablation
// of tools/kbench5 shows the two costs adding up (43 us = 8 fixed + 20 MFMA + 12 loads + 3 epilogue)
// instead of overlapping.  A deeper register pipeline does not fit (114 VGPRs at 4 waves/SIMD); the
// DMA ring needs no staging registers at all.
//
// hipcc orders every LDS access it can see behind an outstanding LDS-DMA with s_waitcnt vmcnt(0)
// (the DMA variant of gemm_f32.h is serialised that way), so the fragment reads here are inline
// ds_read_b128 and the waits are explicit:
//   * a wave's DMAs complete in issue order: "s_waitcnt vmcnt(4 * tiles still allowed in flight)"
//     means this wave's pieces of tile kt have landed; the s_barrier that follows extends that to the
//     other waves' pieces and also says every wave is done reading tile kt-1, whose stage is refilled
//     right after it (one barrier per K-tile);
//   * "s_waitcnt lgkmcnt(n)" before the MFMAs names the fragment registers as in/out operands so the
//     compiler cannot move an MFMA above it.
//
// Shapes: both operands K-contiguous (A [M][K], B [N][K]), 64 x 64 output tile, BK = 32, 4 waves of
// 16 rows x 64 columns (4 MFMA 16x16x4 column tiles; in GATES mode those are the 4 gates of 16 hidden
// units, as in gemm_f32.h).  K, K2 and kslice must be multiples of 32; rows / columns beyond M / N are
// clamped on the load (their products are never stored).  Same GemmArgs / MultiArgs / epilogue
// interface as gemm_f32.h.  LDS images and the k order are those of gemm_f32.h (16-byte chunks
// XOR-swizzled by the row; the swizzle is applied to the DMA's per-lane SOURCE address).
#pragma once
#include <type_traits>
#include "gemm_f32.h"

namespace nvqa {

template <int OFF> __device__ __forceinline__ f32x4 lds_read_b128(uint32_t addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF));
    return v;
}

#define NVQA_RING_S 4
#define NVQA_RING_BK 32

template <bool GATES, class Epi, int SEG, int S = NVQA_RING_S>
__device__ __forceinline__ void gemm_ring_body(const GemmArgs &g, const Epi &epi, const int bx, const int by,
                                               const int bz)
{
    constexpr int BM = 64, BN = 64, BK = NVQA_RING_BK;
    static_assert(S >= 2 && S <= 4, "ring stages");
    constexpr int STAGE = BM * BK + BN * BK; // floats per stage: A image then B image (16 KiB)
    constexpr int BU = BN / 4;               // GATES: units per block
    static_assert(SEG == 0 || SEG == 1, "SEG");
    __shared__ __attribute__((aligned(1024))) float smem[S * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;
    const int m0 = by * BM;
    const int n0 = GATES ? bx * BU : bx * BN;
    const int z = bz;
    const int kbeg = z * g.kslice;
    const int kend = min(g.K, kbeg + g.kslice);
    const int mlim = g.mlimit ? min(g.M, *g.mlimit) : g.M;
    const bool active = m0 < mlim;
    const int nk1 = active && kbeg < kend ? (kend - kbeg) / BK : 0;
    const int nk2 = SEG > 0 && active ? g.K2 / BK : 0;
    const int nk = nk1 + nk2;

    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[b][r] = 0.0f;

    // epilogue inputs that do not depend on the accumulators are requested now (older than every DMA,
    // so the in-order vmcnt waits of the loop cover them too)
    constexpr bool PRE = EpiTraits<Epi>::prefetch;
    typename EpiTraits<Epi>::Pre pre[PRE && !GATES ? 4 : 1][PRE ? 4 : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + 4 * lh + r;
            if (m >= g.M) continue;
            if constexpr (GATES) {
                const int u = n0 + li;
                if (u < g.N) pre[0][r] = epi.preload(m, u);
            } else {
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    const int n = n0 + tb * 16 + li;
                    if (n < g.N) pre[tb][r] = epi.preload(m, n);
                }
            }
        }
    }

    // DMA pieces: a piece is one wave-instruction = 64 lanes x 16 B = 1 KiB of consecutive LDS = 8 rows
    // of a [rows][32] image.  Wave w delivers pieces w and w + 4 of the A image and of the B image.
    constexpr int NSEG = SEG > 0 ? 2 : 1;
    const float *srcA[NSEG][2], *srcB[NSEG][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int P = (wave + 4 * j) * 64 + lane;
        const int row = P >> 3, kq = (P & 7) ^ ((row >> 1) & 7);
        const int m = min(m0 + row, g.M - 1);
        int n;
        if (GATES) n = (row / BU) * g.R + min(n0 + row % BU, g.N - 1);
        else n = min(n0 + row, g.N - 1);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            srcA[sg][j] = (sg ? g.A2 + (size_t)m * g.lda2 : g.A + (size_t)m * g.lda) + 4 * kq;
            srcB[sg][j] = (sg ? g.B2 + (size_t)n * g.ldb2 : g.B + (size_t)n * g.ldb) + 4 * kq;
        }
    }
    typedef __attribute__((address_space(3))) void *lds_t;
    typedef const __attribute__((address_space(1))) void *glb_t;
    // operand = 0: A image, 1: B image; j = 0, 1: this wave's two pieces of it
    auto issue_piece = [&](int kt, int operand, int j) {
        const bool s2 = SEG > 0 && kt >= nk1;
        const int k0 = s2 ? (kt - nk1) * BK : kbeg + kt * BK;
        float *st = smem + (kt % S) * STAGE + operand * (BM * BK) + (wave + 4 * j) * 256;
        const float *src = operand == 0 ? (SEG > 0 && s2 ? srcA[NSEG - 1][j] : srcA[0][j])
                                        : (SEG > 0 && s2 ? srcB[NSEG - 1][j] : srcB[0][j]);
        __builtin_amdgcn_global_load_lds((glb_t)(src + k0), (lds_t)st, 16, 0, 0);
    };
    auto issue = [&](int kt) {
        issue_piece(kt, 0, 0);
        issue_piece(kt, 0, 1);
        issue_piece(kt, 1, 0);
        issue_piece(kt, 1, 1);
    };

    // fragment addresses (bytes): row * 128 + 16 * ((4 q + lh) ^ x), x = (row >> 1) & 7 = (li >> 1) & 7 for
    // every row this lane reads (rows are li + a multiple of 16)
    const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)smem; // LDS byte address of the ring
    const int x = (li >> 1) & 7;
    const uint32_t fa0 = (wave * 16 + li) * (BK * 4) + 16 * ((0 + lh) ^ x);
    const uint32_t fa1 = (wave * 16 + li) * (BK * 4) + 16 * ((4 + lh) ^ x);
    const uint32_t fb0 = BM * BK * 4 + li * (BK * 4) + 16 * ((0 + lh) ^ x);
    const uint32_t fb1 = BM * BK * 4 + li * (BK * 4) + 16 * ((4 + lh) ^ x);
    constexpr int TB = 16 * BK * 4; // bytes between the B rows of consecutive column tiles

    // One K-tile of the pipelined loop: the 32 MFMAs of tile kt (fragments fc, already in registers) with,
    // pinned between them by sched_barrier, the 10 fragment reads of tile kt+1 (into fn) and the 4 DMAs of
    // tile kt+S (into the stage tile kt has just vacated).  A wave alone on its SIMD issues in order, so
    // whatever is not placed inside the MFMA stream is exposed; placed there it costs (almost) nothing
    // because an MFMA occupies the issue port for 8 of its 32 cycles.
    struct Frag { f32x4 a[2], b[2][4]; };
    auto tile = [&](auto next_c, auto issue_c, const Frag &fc, Frag &fn, int kt) {
        constexpr bool NEXT = decltype(next_c)::value, ISSUE = decltype(issue_c)::value;
        const uint32_t sb = sbase + (uint32_t)((kt + 1) % S) * (STAGE * 4);
        int nm = 0; // MFMAs issued so far (compile-time after unrolling)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc.a[q][w], fc.b[q][tb][w], acc[tb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ++nm;
                    if constexpr (NEXT) {
                        if (nm == 1) fn.a[0] = lds_read_b128<0>(sb + fa0);
                        if (nm == 2) fn.b[0][0] = lds_read_b128<0>(sb + fb0);
                        if (nm == 3) fn.b[0][1] = lds_read_b128<TB>(sb + fb0);
                        if (nm == 4) fn.b[0][2] = lds_read_b128<2 * TB>(sb + fb0);
                        if (nm == 5) fn.b[0][3] = lds_read_b128<3 * TB>(sb + fb0);
                        if (nm == 7) fn.a[1] = lds_read_b128<0>(sb + fa1);
                        if (nm == 8) fn.b[1][0] = lds_read_b128<0>(sb + fb1);
                        if (nm == 9) fn.b[1][1] = lds_read_b128<TB>(sb + fb1);
                        if (nm == 10) fn.b[1][2] = lds_read_b128<2 * TB>(sb + fb1);
                        if (nm == 11) fn.b[1][3] = lds_read_b128<3 * TB>(sb + fb1);
                    }
                    if constexpr (ISSUE) {
                        if (nm == 13) issue_piece(kt + S, 0, 0);
                        if (nm == 16) issue_piece(kt + S, 0, 1);
                        if (nm == 19) issue_piece(kt + S, 1, 0);
                        if (nm == 22) issue_piece(kt + S, 1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        if constexpr (NEXT)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fn.a[0]), "+v"(fn.b[0][0]), "+v"(fn.b[0][1]), "+v"(fn.b[0][2]), "+v"(fn.b[0][3]), "+v"(fn.a[1]),
                           "+v"(fn.b[1][0]), "+v"(fn.b[1][1]), "+v"(fn.b[1][2]), "+v"(fn.b[1][3]));
    };

    if (nk > 0) {
        const int npre = min(S, nk);
        for (int kt = 0; kt < npre; ++kt) issue(kt);
        // tile 0 has landed once at most npre - 1 newer tiles (4 DMAs each, in issue order) are in flight
        if (npre >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (npre == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (npre == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        Frag f[2];
        {
            const uint32_t sb = sbase;
            f[0].a[0] = lds_read_b128<0>(sb + fa0);
            f[0].b[0][0] = lds_read_b128<0>(sb + fb0);
            f[0].b[0][1] = lds_read_b128<TB>(sb + fb0);
            f[0].b[0][2] = lds_read_b128<2 * TB>(sb + fb0);
            f[0].b[0][3] = lds_read_b128<3 * TB>(sb + fb0);
            f[0].a[1] = lds_read_b128<0>(sb + fa1);
            f[0].b[1][0] = lds_read_b128<0>(sb + fb1);
            f[0].b[1][1] = lds_read_b128<TB>(sb + fb1);
            f[0].b[1][2] = lds_read_b128<2 * TB>(sb + fb1);
            f[0].b[1][3] = lds_read_b128<3 * TB>(sb + fb1);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0].a[0]), "+v"(f[0].b[0][0]), "+v"(f[0].b[0][1]), "+v"(f[0].b[0][2]), "+v"(f[0].b[0][3]),
                           "+v"(f[0].a[1]), "+v"(f[0].b[1][0]), "+v"(f[0].b[1][1]), "+v"(f[0].b[1][2]), "+v"(f[0].b[1][3]));
        }
        auto step = [&](const Frag &fc, Frag &fn, int kt) {
            if (kt + 1 < nk) {
                // tile kt+1 must have landed: tiles up to min(nk, kt+S) - 1 are issued, those after kt+1 may stay in flight
                const int newer = min(nk, kt + S) - (kt + 2);
                if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // every wave's pieces of tile kt+1 are in LDS, and every wave has the fragments of tile kt in
                // registers (its reads were waited for at the end of the previous step): stage kt % S is free
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + S < nk) tile(std::true_type{}, std::true_type{}, fc, fn, kt);
                else tile(std::true_type{}, std::false_type{}, fc, fn, kt);
            } else {
                tile(std::false_type{}, std::false_type{}, fc, fn, kt);
            }
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(f[0], f[1], kt);
            if (kt + 1 < nk) step(f[1], f[0], kt + 1);
        }
    }

    // epilogue: C/D map of the 16x16x4 MFMA: col = lane % 16, row = 4 * (lane / 16) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 16 + 4 * lh + r;
        if (m >= g.M) continue;
        if constexpr (GATES) {
            const int u = n0 + li;
            if (u < g.N) {
                const float av[4] = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                if constexpr (PRE) epi(m, u, av, pre[0][r]);
                else epi(m, u, av);
            }
        } else {
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                const int n = n0 + tb * 16 + li;
                if (n < g.N) {
                    if constexpr (PRE) epi(z, m, n, acc[tb][r], 0.f, pre[tb][r]);
                    else epi(z, m, n, acc[tb][r]);
                }
            }
        }
    }
}

// true when a product can go through the ring kernel
inline bool ring_ok(const GemmArgs &g, bool gates)
{
    if (g.K % NVQA_RING_BK || (g.K > 0 && g.kslice % NVQA_RING_BK) || g.K2 % NVQA_RING_BK) return false;
    if (g.lda % 4 || g.ldb % 4 || (g.K2 && (g.lda2 % 4 || g.ldb2 % 4))) return false;
    if (g.M < 1 || g.N < 1) return false;
    if (gates && g.N % 16) return false;
    return true;
}

template <bool GATES, class Epi, int SEG, int S>
__global__ __launch_bounds__(256) void gemm_ring_multi_kernel(MultiArgs<Epi> a)
{
    const int p = blockIdx.z / a.zsplit, z = blockIdx.z % a.zsplit;
    gemm_ring_body<GATES, Epi, SEG, S>(a.g[p], a.e[p], blockIdx.x, blockIdx.y, z);
}
template <bool GATES, class Epi, int SEG, int S = NVQA_RING_S>
inline hipError_t launch_gemm_ring_multi(hipStream_t s, const MultiArgs<Epi> &a, int nprob)
{
    const GemmArgs &g = a.g[0]; // all problems share M and N (grid shape)
    dim3 grid(GATES ? (g.N + 15) / 16 : (g.N + 63) / 64, (g.M + 63) / 64, nprob * a.zsplit);
    hipLaunchKernelGGL((gemm_ring_multi_kernel<GATES, Epi, SEG, S>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace nvqa
