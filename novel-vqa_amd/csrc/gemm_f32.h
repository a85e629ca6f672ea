// gemm_f32.h -- LDS-tiled fp32 GEMM on the gfx950 f32-input MFMA
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact f32 fma chains at the
// f32 vector rate, MI355X_MICROARCH.md "Matrix cores").  Every dense product of the
// VQA step runs through this template; what differs per call site is the operand
// layout, the tile shape, the pipeline depth and the fused epilogue.
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
//   A_IM2COL: implicit im2col of an NHWC tensor (3x3 conv), K-contiguous like A_KC
//   B_KC: B stored [N][K]  (k contiguous)     Torch nn.Linear weight [out][in] in forward
//   B_NC: B stored [K][N]  (n contiguous)     the same weight in dgrad, activations in wgrad
// LDS images: K-contiguous operands are stored [rows][BK] with the 16-byte chunk index XOR-swizzled
// by the row (conflict-free ds_read_b128 for the hardware's lane groups); one such read feeds 4
// MFMAs, with the k order permuted identically for A and B: k(q, w, h) = QK*q + 4*h + w (h = lane
// group, w = 0..3).  M/N-contiguous operands are stored k-major [BK][rows + 4] and read with
// ds_read_b32.  Either way the global -> LDS copy is float4 in, float4 out: no transposing writes.
//
// Pipeline: register-staged prefetch (PF = 1: next tile in flight while the current one is
// multiplied, single LDS buffer; PF = 2: two tiles in flight, two LDS buffers, one barrier per
// tile), fragment reads software-pipelined one q step ahead of the MFMAs.  (One tile in flight with two LDS buffers -- the
// next tile stored behind the current tile's MFMAs, one barrier per tile -- was measured in round 3 and is SLOWER than
// PF = 1 on every product of the step and on the convolutions: weight gradients 0.795 -> 0.832 ms, VGG batch 128
// 36.2 -> 38.4 ms; the second barrier is cheaper than the second buffer's LDS footprint.)  Cfg::DMA selects a
// direct global -> LDS variant (global_load_lds_dwordx4, swizzle on the source address); it is
// bit-identical and measured equal or slower, so it is off.  WK > 1 splits each K-tile over WK wave
// groups (more waves per SIMD) and sums the partial accumulators through LDS in a fixed order; the
// epilogue is then spread over the groups.  blockIdx.z selects a K slice (split-K into slabs) or,
// in gemm_f32_multi_kernel, one of several same-shape problems.
//
// GATES mode (fused LSTM cell): N indexes hidden units; the block's B tile holds the
// rows {g*R + u} of the [4R][K] weight for its units u and all 4 gates g, and each lane
// ends up with the 4 gate pre-activations of one (row, unit) in its own registers, so
// the cell update needs no cross-lane traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvqa {
template <class E> struct EpiTraits; // epilogues.h

enum { A_KC = 0, A_MC = 1, A_IM2COL = 2, A_IM2COLF = 3 }; // A_IM2COLF: A_IM2COL when no K-tile straddles two taps (C_in % BK == 0)
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
    // optional second K segment (SEG > 0): C += A2 * B2, same layouts as A / B.
    // SEG == 1 adds into the same accumulator ([x_t | h_{t-1}] x [W_i2h | W_h2h]^T of one LSTM
    // step); SEG == 2 keeps a second accumulator that the epilogue receives separately.
    const float *A2;
    const float *B2;
    int lda2, ldb2, K2;
    // A_IM2COL: A is an NHWC activation [n][cH][cW][cC]; row m = output pixel (n, y, x) of a 3x3,
    // stride-1, pad-1 convolution, k = (ky*3 + kx)*cC + ci (cC % 4 == 0): implicit GEMM, no im2col buffer
    int cH, cW, cC;
    // Ragged time-batched operands: rows come in groups of seg_rows (= B, one group per LSTM step) of
    // which only the first seg_limits[group] are active (the rest are zero padding: questions that
    // have not started yet, misc/RNNUtils.lua:136-145).  mseg: the M rows of A are grouped -> inactive
    // row tiles are skipped altogether (their outputs are never read).  kseg: the K rows of an
    // A_MC / B_NC product are grouped -> all-zero K-tiles are skipped.
    const int *mseg_limits, *kseg_limits;
    int seg_rows;
    // XCD-aware tile order (single-problem launches): workgroups are dealt round-robin over the 8 XCDs, so
    // with the natural order the tiles that share an operand block (same row block, neighbouring column
    // tiles; same K slice) land on 8 different L2s and the block is fetched from HBM up to 8 times.  With
    // xcd = 1 the hardware id is folded so that each XCD owns one contiguous range of the x-fastest tile order.
    int xcd;
};

// hardware workgroup id (x fastest) -> logical tile id, a bijection on [0, total): XCD k = id % 8 receives the
// contiguous logical range [k * per, ...) in dispatch order; the first total % 8 XCDs hold one tile more.
__device__ __forceinline__ unsigned xcd_fold(unsigned id, unsigned total)
{
    const unsigned per = (total + 7) / 8, tall = total % 8 ? total % 8 : 8;
    const unsigned x = id % 8, local = id / 8;
    return x < tall ? x * per + local : tall * per + (x - tall) * (per - 1) + local;
}

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

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x4 pack_bf16(const float4 &v)
{
    const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const u32x2 r = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
    return __builtin_bit_cast(s16x4, r);
}
template <int MF>
__device__ __forceinline__ typename AccT<MF>::type mfma_bf16(s16x4 a, s16x4 b, typename AccT<MF>::type c);
template <> __device__ __forceinline__ f32x16 mfma_bf16<32>(s16x4 a, s16x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_bf16<16>(s16x4 a, s16x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// Tile configuration
// DBG (ablation builds of tools/kbench only): 1 = no global loads, 2 = no MFMA loop, 4 = no epilogue
// DMA = 1: tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write);
// two LDS buffers, the next tile's DMA is in flight while the current one is multiplied.
// BF = 2: both operands are bf16 IN MEMORY, K-contiguous.  Every size, leading dimension and K index of GemmArgs is then in
// units of TWO bf16 (= one float of storage: the pointers are the bf16 arrays reinterpreted), so the loaders, the LDS images
// and their swizzle move 16-byte pieces exactly as in f32; BK = 32 such units = 64 k per tile = two v_mfma_f32_16x16x32_bf16
// per tile pair (MF = 16 only).
// BF = 1: bf16 matrix cores.  The LDS images stay f32; the 4 consecutive k a lane reads per q step are rounded
// to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) and go through ONE v_mfma_f32_16x16x16_bf16 /
// v_mfma_f32_32x32x8_bf16 (same lane -> (row, k) map as the four f32 MFMAs they replace), f32 accumulate.
// SB = 1: a scheduling barrier keeps the LDS reads of the next q step ahead of the MFMAs of the current one (without it the
// compiler sinks them below the MFMAs and every q step starts with an exposed LDS round trip)
// WPE > 0: register budget for WPE waves per SIMD (amdgpu_waves_per_eu), so that two workgroups share a CU
template <int MF_, int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, int PF_, int DBG_ = 0, int DMA_ = 0, int BF_ = 0, int SB_ = 0, int WPE_ = 0> struct Cfg {
    static constexpr int MF = MF_, BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_, PF = PF_;
    static constexpr int DBG = DBG_, DMA = DMA_, BF = BF_, SB = SB_, WPE = WPE_;
};
template <class C> struct WithBF { typedef Cfg<C::MF, C::BM, C::BN, C::BK, C::WM, C::WN, C::WK, C::PF, C::DBG, C::DMA, 1, C::SB, C::WPE> type; };

// Epilogue concept:
//   plain         : void operator()(int z, int m, int n, float v) const
//   plain, SEG==2 : void operator()(int z, int m, int n, float v, float v2) const
//   GATES         : void operator()(int m, int unit, const float (&a)[4]) const
// Called only for m < M, n < N.

// swizzle of the 16-byte chunk index inside a K-contiguous LDS row (conflict-free ds_read_b128
// for the lane groups of MI355X_MICROARCH.md "LDS"): rows of 256 B or more XOR the low 4 chunk
// bits with the row, 128-B rows XOR 3 bits with row/2.
template <int BK> __device__ __forceinline__ int swz_chunk(int row, int chunk)
{
    if constexpr (BK >= 64) return chunk ^ (row & 15);
    else return chunk ^ ((row >> 1) & 7);
}

// EXT: the LDS tile buffers are the caller's (smem_ext, SMEM floats as computed below) instead of a static array of this
// function -- for a caller that runs tiles of a product inside ANOTHER kernel (lstm_persist_bwd2.h: the workgroups without a
// role multiply head weight gradients under the BPTT) and must not add static LDS to it.
template <class C, int AMODE, int BMODE, bool GATES, class Epi, int SEG, bool EXT = false>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs &g, const Epi &epi, const int bx, const int by,
                                              const int bz, float *smem_ext = nullptr)
{
    constexpr int MF = C::MF, BM = C::BM, BN = C::BN, BK = C::BK, WM = C::WM, WN = C::WN, WK = C::WK,
                  PF = C::PF;
    constexpr int NT = 64 * WM * WN * WK;
    constexpr bool IM = AMODE == A_IM2COL || AMODE == A_IM2COLF, IMF = AMODE == A_IM2COLF;
    constexpr int KI = 64 / MF;        // k per MFMA (lane groups h = 0..KI-1)
    constexpr int QK = 4 * KI;         // k per "q step": 4 MFMAs, k(q,w,h) = QK*q + 4*h + w
    constexpr int NQ = BK / QK;        // q steps per K-tile
    constexpr int QPW = NQ / WK;       // q steps per wave group
    constexpr int NREG = MF * MF / 64; // accumulator registers per MFMA tile
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int NTM = TM / MF, NTN = TN / MF;
    constexpr int BU = BN / 4; // GATES: units per block
    static_assert(TM % MF == 0 && TN % MF == 0, "wave tile must be a multiple of the MFMA tile");
    static_assert(!GATES || (NTN == 4 && BMODE == B_KC), "GATES: 4 gate sub-tiles per wave, weight [4R][K]");
    static_assert(BK == 32 || BK == 64 || BK == 128, "BK");
    static_assert(NQ % WK == 0 && NREG % WK == 0, "WK must divide the q steps and the accumulator registers");
    static_assert(PF == 1 || PF == 2, "PF");
    static_assert(C::BF != 2 || (MF == 16 && AMODE != A_MC && BMODE == B_KC && !GATES && SEG == 0), "BF = 2: K-contiguous bf16 operands, 16x16x32 MFMA");
    static_assert(SEG != 2 || !GATES, "SEG 2 is for the plain epilogue");
    // K-contiguous operands: LDS image [rows][BK], chunk-swizzled, read with ds_read_b128.
    // M/N-contiguous operands: LDS image [BK][rows + 4], read with ds_read_b32 (the 4 floats of a
    // q step sit 4 rows apart for the KI lane groups: (4*LD) % 32 == 16 keeps MF=16 conflict-free).
    // (an LDS-DMA piece is 1 KiB of consecutive LDS: k-major images cannot be padded then, which is
    // conflict-free only for the 32-wide MFMA operand read)
    constexpr bool DMA = C::DMA != 0;
    static_assert(!DMA || MF == 32 || (AMODE != A_MC && BMODE == B_KC), "DMA with k-major operands needs MF = 32");
    constexpr int LDA = AMODE != A_MC ? BK : (DMA ? BM : BM + 4);
    constexpr int LDB = BMODE == B_KC ? BK : (DMA ? BN : BN + 4);
    constexpr int A_FL = AMODE != A_MC ? BM * BK : BK * LDA;
    constexpr int B_FL = BMODE == B_KC ? BN * BK : BK * LDB;
    constexpr int A_F4 = BM * BK / 4, B_F4 = BN * BK / 4;
    constexpr int NA = (A_F4 + NT - 1) / NT, NB = (B_F4 + NT - 1) / NT;
    constexpr int BUF = A_FL + B_FL; // floats per LDS buffer (A image then B image)
    constexpr int ACCN = NTM * NTN * NREG;
    constexpr int RED = WK > 1 ? WK * WM * WN * ACCN * 64 : 0;
    constexpr int NBUF = DMA ? 2 : PF;
    constexpr int SMEM = (NBUF * BUF > RED ? NBUF * BUF : RED);

    float *smem;
    if constexpr (EXT) {
        smem = smem_ext;
    } else {
        __shared__ __attribute__((aligned(16))) float smem_static[SMEM];
        smem = smem_static;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (WM * WN), wrem = wave % (WM * WN);
    const int wm = wrem / WN, wn = wrem % WN;
    const int li = lane % MF, lh = lane / MF;
    const int m0 = by * BM;
    const int n0 = GATES ? bx * BU : bx * BN; // GATES: first unit
    const int z = bz;
    const int kbeg = z * g.kslice;
    const int kend = min(g.K, kbeg + g.kslice);
    const int mlim = g.mlimit ? min(g.M, *g.mlimit) : g.M;
    const bool active = m0 < mlim;
    if (g.mseg_limits && g.seg_rows % BM == 0) { // block-uniform, before any barrier
        if (m0 % g.seg_rows >= g.mseg_limits[m0 / g.seg_rows]) return;
    }

    typename AccT<MF>::type acc[NTM][NTN];
#pragma unroll
    for (int a = 0; a < NTM; ++a)
#pragma unroll
        for (int b = 0; b < NTN; ++b)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[a][b][r] = 0.0f;
    typename AccT<MF>::type acc2[SEG == 2 ? NTM : 1][SEG == 2 ? NTN : 1];
    if constexpr (SEG == 2) {
#pragma unroll
        for (int a = 0; a < NTM; ++a)
#pragma unroll
            for (int b = 0; b < NTN; ++b)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc2[a][b][r] = 0.0f;
    }

    // epilogue inputs that do not depend on the accumulators are requested now (EpiTraits::prefetch)
    constexpr bool PRE = EpiTraits<Epi>::prefetch;
    typename EpiTraits<Epi>::Pre pre[PRE ? NTM : 1][PRE && !GATES ? NTN : 1][PRE ? NREG : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                if (WK > 1 && r % WK != wk) continue;
                const int m = m0 + wm * TM + ta * MF + 8 * (r >> 2) + 4 * lh + (r & 3);
                if (m >= g.M) continue;
                if constexpr (GATES) {
                    const int u = n0 + wn * MF + li;
                    if (u < g.N) pre[ta][0][r] = epi.preload(m, u);
                } else {
#pragma unroll
                    for (int tb = 0; tb < NTN; ++tb) {
                        const int n = n0 + wn * TN + tb * MF + li;
                        if (n < g.N) pre[ta][tb][r] = epi.preload(m, n);
                    }
                }
            }
    }

    // tile index -> (segment, k range)
    const bool kskip = SEG == 0 && g.kseg_limits != nullptr && g.seg_rows % BK == 0 && kbeg % BK == 0;
    int nk1 = active && kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
    // kseg: only the K-tiles that hold active rows are multiplied, and they are dealt to the K slices ROUND-ROBIN (tile
    // i of the active list goes to slice i % slices): a ragged batch has its active rows in the late steps, so
    // contiguous slices would leave the last slice as long as ever and the others idle.  kit_* walks the active list
    // (load_tiles is called with kt = 0, 1, 2, ... exactly once each and advances `slices` tiles per call).
    int kit_seg = -1, kit_next = 0, kit_end = 0;
    const int kz = gridDim.z > 1 ? (int)gridDim.z : 1; // slices (single-problem launches only carry kseg)
    if (kskip) {
        int total = 0;
        const int nseg = (g.K + g.seg_rows - 1) / g.seg_rows;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const int lim = min(g.kseg_limits[sgm], g.K - sgm * g.seg_rows);
            if (lim > 0) total += (lim + BK - 1) / BK;
        }
        nk1 = active && total > z ? (total - z + kz - 1) / kz : 0;
    }
    const int nk2 = SEG > 0 && active ? (g.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;

    // ---- per-thread invariants of the tile loop (hoisted: the step kernels spend as many issue
    // cycles on address arithmetic as on MFMAs otherwise) ---------------------------------------
    // global source pointer of each staged float4 at tile 0 of a segment (nullptr: row/col out of
    // range -> zeros), its k offset inside the tile, and its LDS destination offset
    constexpr int NSEG = SEG > 0 ? 2 : 1;
    const float *pA[NSEG][NA], *pB[NSEG][NB];
    int kA[NA], kB[NB], sA[NA], sB[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int f = tid + j * NT;
        const bool inb = A_F4 % NT == 0 || f < A_F4;
        if (AMODE != A_MC) {
            const int row = f / (BK / 4), kq = f % (BK / 4), m = m0 + row;
            kA[j] = 4 * kq;
            sA[j] = row * BK + 4 * swz_chunk<BK>(row, kq);
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg)
                pA[sg][j] = (inb && m < mlim && AMODE == A_KC) ? (sg ? g.A2 + (size_t)m * g.lda2 : g.A + (size_t)m * g.lda) + 4 * kq : nullptr;
        } else {
            const int kr = f / (BM / 4), mq = f % (BM / 4), m = m0 + 4 * mq;
            kA[j] = kr;
            sA[j] = kr * LDA + 4 * mq;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg)
                pA[sg][j] = (inb && m < g.M) ? (sg ? g.A2 + (size_t)kr * g.lda2 : g.A + (size_t)kr * g.lda) + m : nullptr;
        }
        if (!inb) sA[j] = -1;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int f = tid + j * NT;
        const bool inb = B_F4 % NT == 0 || f < B_F4;
        if (BMODE == B_KC) {
            const int row = f / (BK / 4), kq = f % (BK / 4);
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
            kB[j] = 4 * kq;
            sB[j] = row * BK + 4 * swz_chunk<BK>(row, kq);
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg)
                pB[sg][j] = (inb && ok) ? (sg ? g.B2 + (size_t)n * g.ldb2 : g.B + (size_t)n * g.ldb) + 4 * kq : nullptr;
        } else {
            const int kr = f / (BN / 4), nq = f % (BN / 4), n = n0 + 4 * nq;
            kB[j] = kr;
            sB[j] = kr * LDB + 4 * nq;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg)
                pB[sg][j] = (inb && n < g.N) ? (sg ? g.B2 + (size_t)kr * g.ldb2 : g.B + (size_t)kr * g.ldb) + n : nullptr;
        }
        if (!inb) sB[j] = -1;
    }

    // A_IM2COL: output pixel (n, y, x) of each staged row, decomposed ONCE (rows do not change over the K
    // loop); imP = address of that pixel's channel 0 in the NHWC input, NULL for rows beyond the image batch
    const float *imP[IM ? NA : 1];
    int imY[IM && !IMF ? NA : 1], imX[IM && !IMF ? NA : 1];
    // fast path (a K-tile never straddles two taps: C_in % BK == 0, every VGG layer but the first): the tap of a tile is
    // uniform, so (ky, kx, channel offset) live in scalar registers, advanced tile by tile, and a row's share of the work is
    // one bit test against imMask (bit t: tap t of this output pixel lies inside the image) and one add
    unsigned imMask[IMF ? NA : 1];
    constexpr bool imFast = AMODE == A_IM2COLF; // the launcher checked g.cC % BK == 0 (then K = 9 C_in and the K slices are whole tiles too)
    int imTap = 0, imC0 = 0; // of the NEXT tile load_tiles is asked for (tiles are requested in order)
    if constexpr (IM) {
        imTap = kbeg / g.cC; imC0 = kbeg - imTap * g.cC;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int f = tid + j * NT;
            const int m = m0 + f / (BK / 4);
            imP[j] = nullptr;
            if constexpr (IMF) imMask[j] = 0; else { imY[j] = 0; imX[j] = 0; }
            if ((A_F4 % NT == 0 || f < A_F4) && m < mlim) {
                const int hw = g.cH * g.cW;
                const int n = m / hw, rem = m - n * hw, y = rem / g.cW, x = rem - y * g.cW;
                imP[j] = g.A + (((size_t)n * g.cH + y) * g.cW + x) * g.cC;
                if constexpr (!IMF) { imY[j] = y; imX[j] = x; }
                const unsigned rows = (y >= 1 ? 1u : 0u) | 2u | (y + 1 < g.cH ? 4u : 0u);
                const unsigned cols = (x >= 1 ? 1u : 0u) | 2u | (x + 1 < g.cW ? 4u : 0u);
                if constexpr (IMF) imMask[j] = ((rows & 1u) ? cols : 0u) | (cols << 3) | ((rows & 4u) ? cols << 6 : 0u);
            }
        }
    }

    auto load_tiles = [&](int kt, float4(&ra)[NA], float4(&rb)[NB]) {
        if constexpr (C::DBG & 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int j = 0; j < NB; ++j) rb[j] = make_float4(1.f, 1.f, 1.f, 1.f);
            return;
        }
        const bool s2 = SEG > 0 && kt >= nk1;
        int k0 = s2 ? (kt - nk1) * BK : kbeg + kt * BK;
        const int kend = s2 ? g.K2 : (kskip ? g.K : min(g.K, kbeg + g.kslice));
        if (kskip) {
            for (int adv = kt == 0 ? z + 1 : kz; adv > 0; --adv) { // to my next tile of the active list
                while (kit_seg < 0 || kit_next >= kit_end) {
                    ++kit_seg;
                    kit_next = kit_seg * g.seg_rows;
                    kit_end = min(kit_seg * g.seg_rows + g.kseg_limits[kit_seg], g.K);
                }
                k0 = kit_next;
                kit_next += BK;
            }
        }
        const int glda = s2 ? g.lda2 : g.lda, gldb = s2 ? g.ldb2 : g.ldb;
        int tap0 = 0, cK0 = 0;
        if constexpr (IM) {
            if constexpr (imFast) {
                // uniform: tap and first channel of this tile, then the state of the next one
                const int tap = imTap, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
                const ptrdiff_t off = ((ptrdiff_t)(ky - 1) * g.cW + (kx - 1)) * g.cC + imC0;
                imC0 += BK;
                if (imC0 >= g.cC) { imC0 = 0; ++imTap; }
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    // branch-free: a row outside the image reads a harmless address and is zeroed by a select
                    const bool ok = (imMask[j] >> tap) & 1u;
                    const float4 t = *reinterpret_cast<const float4 *>(ok ? imP[j] + off + kA[j] : g.A);
                    ra[j] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else { tap0 = k0 / g.cC; cK0 = k0 - tap0 * g.cC; }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (imFast) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (IM) {
                // k -> (tap, channel): k0 is uniform, so the division by the channel count is done once per tile;
                // a staged float4 crosses into the next tap(s) only when a tap is narrower than the tile (C_in < BK)
                int ci = cK0 + kA[j], tap = tap0;
                if (g.cC < BK) { tap += ci / g.cC; ci %= g.cC; }
                else if (ci >= g.cC) { ci -= g.cC; ++tap; }
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = imY[j] + ky - 1, ix = imX[j] + kx - 1;
                if (imP[j] && k0 + kA[j] < kend && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW)
                    v = *reinterpret_cast<const float4 *>(imP[j] + ((ptrdiff_t)(ky - 1) * g.cW + (kx - 1)) * g.cC + ci);
            } else {
                const float *p = SEG > 0 && s2 ? pA[NSEG - 1][j] : pA[0][j];
                if (p && k0 + kA[j] < kend)
                    v = *reinterpret_cast<const float4 *>(AMODE == A_KC ? p + k0 : p + (size_t)k0 * glda);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *p = SEG > 0 && s2 ? pB[NSEG - 1][j] : pB[0][j];
            if constexpr (IM) { // the convolutions: unconditional load + select instead of a branch per load
                const bool ok = p && k0 + kB[j] < kend;
                const float4 t = *reinterpret_cast<const float4 *>(ok ? (BMODE == B_KC ? p + k0 : p + (size_t)k0 * gldb) : g.B);
                if (ok) v = t;
            } else if (p && k0 + kB[j] < kend)
                v = *reinterpret_cast<const float4 *>(BMODE == B_KC ? p + k0 : p + (size_t)k0 * gldb);
            rb[j] = v;
        }
    };
    auto store_tiles = [&](int buf, const float4(&ra)[NA], const float4(&rb)[NB]) {
        float *As = smem + buf * BUF, *Bs = As + A_FL;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (A_F4 % NT == 0 || sA[j] >= 0) *reinterpret_cast<float4 *>(&As[sA[j]]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (B_F4 % NT == 0 || sB[j] >= 0) *reinterpret_cast<float4 *>(&Bs[sB[j]]) = rb[j];
    };

    // MFMA operand fragments: lane (li, lh) of q step q reads the 4 consecutive k of chunk KI*q + lh
    // (K-contiguous images, one ds_read_b128) or k = QK*q + 4*lh + {0..3} (k-major images, 4 ds_read_b32)
    int fA[NTM], xA[NTM], fB[NTN], xB[NTN];
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
        const int row = wm * TM + t * MF + li;
        fA[t] = AMODE != A_MC ? row * BK : 4 * lh * LDA + row;
        xA[t] = BK >= 64 ? (row & 15) : ((row >> 1) & 7);
    }
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int col = GATES ? t * BU + wn * MF + li : wn * TN + t * MF + li;
        fB[t] = BMODE == B_KC ? col * BK : 4 * lh * LDB + col;
        xB[t] = BK >= 64 ? (col & 15) : ((col >> 1) & 7);
    }
    auto read_frags = [&](const float *As, const float *Bs, int q, float4(&a)[NTM], float4(&b)[NTN]) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) {
            if constexpr (AMODE != A_MC) {
                a[t] = *reinterpret_cast<const float4 *>(&As[fA[t] + 4 * ((KI * q + lh) ^ xA[t])]);
            } else {
                const float *p = &As[fA[t] + QK * q * LDA];
                a[t] = make_float4(p[0], p[LDA], p[2 * LDA], p[3 * LDA]);
            }
        }
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            if constexpr (BMODE == B_KC) {
                b[t] = *reinterpret_cast<const float4 *>(&Bs[fB[t] + 4 * ((KI * q + lh) ^ xB[t])]);
            } else {
                const float *p = &Bs[fB[t] + QK * q * LDB];
                b[t] = make_float4(p[0], p[LDB], p[2 * LDB], p[3 * LDB]);
            }
        }
    };
    auto compute = [&](int buf, typename AccT<MF>::type(&acc)[NTM][NTN]) {
        if constexpr (C::DBG & 2) return;
        const float *As = smem + buf * BUF, *Bs = As + A_FL;
        // software-pipelined: the fragments of q step qq+1 are requested before the MFMAs of qq
        float4 a[2][NTM], b[2][NTN];
        read_frags(As, Bs, wk * QPW, a[0], b[0]);
#pragma unroll
        for (int qq = 0; qq < QPW; ++qq) {
            if (qq + 1 < QPW) read_frags(As, Bs, wk * QPW + qq + 1, a[(qq + 1) & 1], b[(qq + 1) & 1]);
            if constexpr (C::SB != 0) __builtin_amdgcn_sched_barrier(0); // the reads of q step qq+1 stay ahead of the MFMAs of qq
            if constexpr (C::BF == 2) {
                // gfx950 form: the LDS images ARE bf16 (a 16-byte piece = 8 consecutive k = one lane's fragment of
                // v_mfma_f32_16x16x32_bf16: lane (li, lh) holds k = 32 q + 8 lh + 0..7 of row / column li); one MFMA per
                // tile pair and q step, no conversion
#pragma unroll
                for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTN; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[qq & 1][ta]),
                                                                              __builtin_bit_cast(bf16x8, b[qq & 1][tb]), acc[ta][tb], 0, 0, 0);
                continue;
            } else if constexpr (C::BF != 0) {
                s16x4 ap[NTM], bp[NTN];
#pragma unroll
                for (int ta = 0; ta < NTM; ++ta) ap[ta] = pack_bf16(a[qq & 1][ta]);
#pragma unroll
                for (int tb = 0; tb < NTN; ++tb) bp[tb] = pack_bf16(b[qq & 1][tb]);
#pragma unroll
                for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTN; ++tb) acc[ta][tb] = mfma_bf16<MF>(ap[ta], bp[tb], acc[ta][tb]);
                continue;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NTN; ++tb) {
                        const float4 &av4 = a[qq & 1][ta], &bv4 = b[qq & 1][tb];
                        const float av = w == 0 ? av4.x : w == 1 ? av4.y : w == 2 ? av4.z : av4.w;
                        const float bv = w == 0 ? bv4.x : w == 1 ? bv4.y : w == 2 ? bv4.z : bv4.w;
                        acc[ta][tb] = mfma<MF>(av, bv, acc[ta][tb]);
                    }
        }
    };

    // LDS-DMA: lane l of wave-instruction i delivers the 16-byte chunk P = i*64 + l of the LDS image;
    // the (inverse-)swizzle goes on the per-lane SOURCE address (cdna_hip_programming.md rule 21).
    // Out-of-range chunks are zero-filled with an ordinary LDS store instead.
    auto dma_tiles = [&](int kt, int buf) {
        constexpr int NW = NT / 64;
        const bool s2 = SEG > 0 && kt >= nk1;
        const float *gA = s2 ? g.A2 : g.A, *gB = s2 ? g.B2 : g.B;
        const int glda = s2 ? g.lda2 : g.lda, gldb = s2 ? g.ldb2 : g.ldb;
        const int k0 = s2 ? (kt - nk1) * BK : kbeg + kt * BK;
        const int kend = s2 ? g.K2 : min(g.K, kbeg + g.kslice);
        float *As = smem + buf * BUF, *Bs = As + A_FL;
        typedef __attribute__((address_space(3))) void *lds_t;
        typedef const __attribute__((address_space(1))) void *glb_t;
#pragma unroll
        for (int i = 0; i < (A_F4 + NT - 1) / NT; ++i) {
            const int piece = i * NW + wave; // wave-uniform
            const int P = piece * 64 + lane;
            if (A_F4 % NT != 0 && piece * 64 >= A_F4) break;
            const float *src = nullptr;
            if (AMODE != A_MC) {
                const int row = P / (BK / 4), kq = swz_chunk<BK>(row, P % (BK / 4));
                const int m = m0 + row, k = k0 + 4 * kq;
                if (AMODE == A_KC) {
                    if (m < mlim && k < kend) src = gA + (size_t)m * glda + k;
                } else if (m < mlim && k < kend) {
                    const int hw = g.cH * g.cW;
                    const int n = m / hw, rem = m - n * hw, y = rem / g.cW, x = rem - y * g.cW;
                    const int tap = k / g.cC, ci = k - tap * g.cC, ky = tap / 3, kx = tap - ky * 3;
                    const int iy = y + ky - 1, ix = x + kx - 1;
                    if (iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW) src = gA + (((size_t)n * g.cH + iy) * g.cW + ix) * g.cC + ci;
                }
            } else {
                const int kr = P / (BM / 4), mq = P % (BM / 4);
                const int k = k0 + kr, m = m0 + 4 * mq;
                if (k < kend && m < g.M) src = gA + (size_t)k * glda + m;
            }
            if (src) __builtin_amdgcn_global_load_lds((glb_t)src, (lds_t)(As + piece * 256), 16, 0, 0);
            else *reinterpret_cast<float4 *>(As + (size_t)P * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < (B_F4 + NT - 1) / NT; ++i) {
            const int piece = i * NW + wave;
            const int P = piece * 64 + lane;
            if (B_F4 % NT != 0 && piece * 64 >= B_F4) break;
            const float *src = nullptr;
            if (BMODE == B_KC) {
                const int row = P / (BK / 4), kq = swz_chunk<BK>(row, P % (BK / 4));
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
                if (ok && k < kend) src = gB + (size_t)n * gldb + k;
            } else {
                const int kr = P / (BN / 4), nq = P % (BN / 4);
                const int k = k0 + kr, n = n0 + 4 * nq;
                if (k < kend && n < g.N) src = gB + (size_t)k * gldb + n;
            }
            if (src) __builtin_amdgcn_global_load_lds((glb_t)src, (lds_t)(Bs + piece * 256), 16, 0, 0);
            else *reinterpret_cast<float4 *>(Bs + (size_t)P * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    auto compute_tile = [&](int buf, int kt) {
        if constexpr (SEG == 2) {
            if (kt >= nk1) { compute(buf, acc2); return; }
        }
        compute(buf, acc);
    };
    if (nk > 0) {
        if constexpr (DMA) {
            dma_tiles(0, 0);
            __syncthreads(); // hipcc drains vmcnt before the barrier: the DMA has landed
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) dma_tiles(kt + 1, (kt + 1) & 1);
                compute_tile(kt & 1, kt);
                __syncthreads();
            }
        } else if constexpr (PF == 1) {
            float4 ra[NA], rb[NB];
            load_tiles(0, ra, rb);
            store_tiles(0, ra, rb);
            __syncthreads();
            for (int kt = 0; kt < nk; ++kt) {
                const bool more = kt + 1 < nk;
                if (more) load_tiles(kt + 1, ra, rb);
                compute_tile(0, kt);
                __syncthreads();
                if (more) {
                    store_tiles(0, ra, rb);
                    __syncthreads();
                }
            }
        } else {
            // tiles t+1 and t+2 in flight while tile t is multiplied; one barrier per tile
            float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
            load_tiles(0, ra0, rb0);
            if (nk > 1) load_tiles(1, ra1, rb1);
            store_tiles(0, ra0, rb0);
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {
                if (kt + 2 < nk) load_tiles(kt + 2, ra0, rb0);
                compute_tile(0, kt);
                if (kt + 1 < nk) store_tiles(1, ra1, rb1);
                __syncthreads();
                if (kt + 1 >= nk) break;
                if (kt + 3 < nk) load_tiles(kt + 3, ra1, rb1);
                compute_tile(1, kt + 1);
                if (kt + 2 < nk) store_tiles(0, ra0, rb0);
                __syncthreads();
            }
        }
    }

    if constexpr (WK > 1) {
        // Every wave group publishes its partial accumulators; group wk then owns the accumulator
        // registers r with r % WK == wk, sums the WK partials of those in group order (deterministic)
        // and runs the epilogue for them, so the epilogue is spread over all waves.
        auto reduce = [&](typename AccT<MF>::type(&ac)[NTM][NTN]) {
            __syncthreads();
            float *dst = smem + ((size_t)(wk * WM * WN + wrem) * ACCN) * 64 + lane;
#pragma unroll
            for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTN; ++tb)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) dst[((ta * NTN + tb) * NREG + r) * 64] = ac[ta][tb][r];
            __syncthreads();
#pragma unroll
            for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTN; ++tb)
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        if (r % WK != wk) continue;
                        float sacc = 0.f;
#pragma unroll
                        for (int gk = 0; gk < WK; ++gk)
                            sacc += smem[((size_t)(gk * WM * WN + wrem) * ACCN + (ta * NTN + tb) * NREG + r) * 64 + lane];
                        ac[ta][tb][r] = sacc;
                    }
        };
        if (nk > 0) {
            reduce(acc);
            if constexpr (SEG == 2) reduce(acc2);
        }
    }

    if constexpr (C::DBG & 4) { // keep the accumulators live, write nothing
#pragma unroll
        for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
            for (int tb = 0; tb < NTN; ++tb) asm volatile("" ::"v"(acc[ta][tb]));
        return;
    }
    // tile-native epilogue (EpiTraits::vec4, MF = 16, WK = 1): the lane's 4 accumulator registers of an MFMA tile go out
    // as ONE 16-byte store into a slab whose element order is private to the producer and its consumer
    // (no epilogue uses it since round 3 removed the fused BPTT level kernel): slot = ((wave * NTM + ta) * NTN + tb) * 64 + lane
    if constexpr (EpiTraits<Epi>::vec4) {
        static_assert(MF == 16 && WK == 1 && !GATES && SEG == 0, "vec4 epilogue: 16x16 MFMA tiles, no K-group split");
        const unsigned tile = by * ((g.N + BN - 1) / BN) + bx;
#pragma unroll
        for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
            for (int tb = 0; tb < NTN; ++tb) epi.store4(z, tile, ((wrem * NTM + ta) * NTN + tb) * 64 + lane, acc[ta][tb]);
        return;
    }
    // epilogue: C/D map (cdna_hip_programming.md section 3): col = lane % MF,
    // row = 8*(reg>>2) + 4*(lane / MF) + (reg & 3)
#pragma unroll
    for (int ta = 0; ta < NTM; ++ta)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            if (WK > 1 && r % WK != wk) continue;
            const int m = m0 + wm * TM + ta * MF + 8 * (r >> 2) + 4 * lh + (r & 3);
            if (m >= g.M) continue;
            if constexpr (GATES) {
                const int u = n0 + wn * MF + li;
                if (u < g.N) {
                    const float av[4] = {acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]};
                    if constexpr (PRE) epi(m, u, av, pre[ta][0][r]);
                    else epi(m, u, av);
                }
            } else {
#pragma unroll
                for (int tb = 0; tb < NTN; ++tb) {
                    const int n = n0 + wn * TN + tb * MF + li;
                    if (n < g.N) {
                        if constexpr (PRE) epi(z, m, n, acc[ta][tb][r], SEG == 2 ? acc2[ta][tb][r] : 0.f, pre[ta][tb][r]);
                        else if constexpr (SEG == 2) epi(z, m, n, acc[ta][tb][r], acc2[ta][tb][r]);
                        else epi(z, m, n, acc[ta][tb][r]);
                    }
                }
            }
        }
}

template <class C, int AMODE, int BMODE, bool GATES, class Epi, int SEG = 0>
__global__ __launch_bounds__(64 * C::WM * C::WN * C::WK) __attribute__((amdgpu_waves_per_eu(C::WPE > 0 ? C::WPE : 1, C::WPE > 0 ? C::WPE : 8)))
void gemm_f32_kernel(GemmArgs g, Epi epi)
{
    if (g.xcd) {
        const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
        const unsigned j = xcd_fold(blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), total);
        gemm_f32_body<C, AMODE, BMODE, GATES, Epi, SEG>(g, epi, j % nx, (j / nx) % ny, j / (nx * ny));
    } else {
        gemm_f32_body<C, AMODE, BMODE, GATES, Epi, SEG>(g, epi, blockIdx.x, blockIdx.y, blockIdx.z);
    }
}

// K slices of a level product whose rows beyond *mlimit are inactive: the fewer row tiles have work, the deeper the
// split, so that the level still fills the chip and each workgroup's K loop gets shorter (a BPTT level of a ragged batch:
// B = 512: 8 row tiles -> 4 slices, the tuned full-batch form; 4 -> 8; 1 or 2 -> 16): as deep as the workgroups of the
// launch (all row tiles x zbase slices per column tile) allow.  Producer and finisher use the same rule.
__host__ __device__ inline int zsplit_for(int active_mtiles, int all_mtiles, int zbase, int zmax)
{
    int z = zbase;
    while (z * 2 <= zmax && active_mtiles * z * 2 <= all_mtiles * zbase) z *= 2;
    return z;
}

// Several independent problems of the same shape class in ONE launch (blockIdx.z = problem):
// the LSTM steps of different layers on one wavefront diagonal.  More workgroups per launch
// (two per CU overlap each other's prologue / epilogue) and one kernel boundary per diagonal.
#define NVQA_MULTI_MAX 8
template <class Epi> struct MultiArgs {
    GemmArgs g[NVQA_MULTI_MAX];
    Epi e[NVQA_MULTI_MAX];
    int zsplit = 1; // K slices per problem (cross-CU split-K: the epilogue sees z and writes a slab)
    int zadapt = 0; // != 0 (ragged BPTT levels): the deepest split allowed; each problem then uses zsplit_for(its active row
                    // tiles, zsplit, zadapt) slices, chosen on the device from *g.mlimit, and the launch's workgroups of
                    // that problem (still row tiles x column tiles x zsplit of them) are re-dealt over (active row tile,
                    // column tile, slice): inactive row tiles are not computed, idle workgroups leave at once
    int xcd = 0;    // fold the hardware workgroup id so that each XCD owns a contiguous range of the x-fastest
                    // (tile, K slice, problem) order: one K slice of one problem per XCD shares its operand slabs in L2
};
template <class C, int AMODE, int BMODE, bool GATES, class Epi, int SEG>
__global__ __launch_bounds__(64 * C::WM * C::WN * C::WK) void gemm_f32_multi_kernel(MultiArgs<Epi> a)
{
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd == 2 && gridDim.z == 2 && gridDim.x % 4 == 0) {
        // forward wavefront level with two layers in flight: XCD k = id % 8 owns (layer k / 4, quarter k % 4 of the
        // unit tiles) for every row block, so that its L2 holds one layer's activations once and a quarter of that
        // layer's weights (2 MB + 2 MB at R = 512) instead of both layers' activations and an eighth of both weights
        const unsigned nx = gridDim.x, ny = gridDim.y;
        const unsigned id = bx + nx * (by + ny * bz), k = id % 8, local = id / 8, qx = nx / 4;
        bz = k / 4;
        bx = (k % 4) * qx + local % qx;
        by = local / qx;
    } else if (a.xcd == 3 && gridDim.x % 4 == 0 && gridDim.y % 2 == 0) {
        // BPTT level: XCD k = id % 8 owns (row half k / 4, column quarter k % 4) of every
        // product and K slice: its L2 sees half of each dG and a quarter of each W instead of all of dG and an eighth of W
        const unsigned nx = gridDim.x, ny = gridDim.y, qx = nx / 4, hy = ny / 2;
        const unsigned id = bx + nx * (by + ny * bz), k = id % 8, local = id / 8;
        bx = (k % 4) * qx + local % qx;
        by = (k / 4) * hy + (local / qx) % hy;
        bz = local / (qx * hy);
    } else if (a.xcd) {
        const unsigned nx = gridDim.x, ny = gridDim.y;
        const unsigned j = xcd_fold(bx + nx * (by + ny * bz), nx * ny * gridDim.z);
        bx = j % nx; by = (j / nx) % ny; bz = j / (nx * ny);
    }
    const int p = bz / a.zsplit, z = bz % a.zsplit;
    if (a.zadapt) {
        const int nr = min(a.g[p].M, *a.g[p].mlimit), mt = (nr + C::BM - 1) / C::BM;
        const int ze = zsplit_for(mt, (int)gridDim.y, a.zsplit, a.zadapt); // mt * ze <= row tiles * zsplit: the problem's workgroups suffice
        const unsigned nx = gridDim.x, slot = bx + nx * (by + gridDim.y * z);
        if (slot >= nx * mt * ze) return; // (nobody reads the inactive rows of the slabs: the finisher writes their zeros)
        GemmArgs g = a.g[p];
        g.kslice = (g.K / ze + C::BK - 1) / C::BK * C::BK;
        gemm_f32_body<C, AMODE, BMODE, GATES, Epi, SEG>(g, a.e[p], slot % nx, (slot / nx) % mt, slot / (nx * mt));
        return;
    }
    gemm_f32_body<C, AMODE, BMODE, GATES, Epi, SEG>(a.g[p], a.e[p], bx, by, z);
}
template <class C, int AMODE, int BMODE, bool GATES, class Epi, int SEG>
inline hipError_t launch_gemm_multi(hipStream_t s, const MultiArgs<Epi> &a, int nprob)
{
    const GemmArgs &g = a.g[0]; // all problems share M and N (grid shape)
    dim3 grid(GATES ? (g.N + C::BN / 4 - 1) / (C::BN / 4) : (g.N + C::BN - 1) / C::BN,
              (g.M + C::BM - 1) / C::BM, nprob * a.zsplit);
    hipLaunchKernelGGL((gemm_f32_multi_kernel<C, AMODE, BMODE, GATES, Epi, SEG>), grid,
                       dim3(64 * C::WM * C::WN * C::WK), 0, s, a);
    return hipGetLastError();
}

template <class C, int AMODE, int BMODE, bool GATES, class Epi, int SEG = 0>
inline hipError_t launch_gemm(hipStream_t s, const GemmArgs &g, const Epi &epi)
{
    const int ksplit = (g.K + g.kslice - 1) / g.kslice;
    dim3 grid(GATES ? (g.N + C::BN / 4 - 1) / (C::BN / 4) : (g.N + C::BN - 1) / C::BN,
              (g.M + C::BM - 1) / C::BM, ksplit < 1 ? 1 : ksplit);
    hipLaunchKernelGGL((gemm_f32_kernel<C, AMODE, BMODE, GATES, Epi, SEG>), grid,
                       dim3(64 * C::WM * C::WN * C::WK), 0, s, g, epi);
    return hipGetLastError();
}

} // namespace nvqa
