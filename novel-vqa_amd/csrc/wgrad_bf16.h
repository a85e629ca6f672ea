// wgrad_bf16.h -- the time-batched weight gradients of the bf16 operand mode (nvqa_set_precision(1)) as a gfx950 kernel:
//     dW[M x N] = A^T B,   A = dG stored [K][lda] (K = steps x rows, M gate pre-activations contiguous),
//                          B = the layer's input / previous hidden state stored [K][ldb] (N contiguous)
// Both operands are K-MAJOR: the reduction index is the slow one, so a lane's MFMA fragment (8 consecutive k of one m)
// is a COLUMN of the tile.  gemm_f32.h's BF mode reads it with 4 ds_read_b32 per 4 k from an f32 image and converts per
// MFMA (v_mfma_f32_32x32x8_bf16_1k): 0.38 ms per step, LDS- and VALU-bound, the matrix pipe 10 % busy.  Here:
//   * the f32 rows are rounded to bf16 ONCE on their way into LDS (v_cvt_pk_bf16_f32, round-to-nearest-even -- the same
//     rounding of the same values as before: bit-identical products), the image is [k][128 m] bf16, half the bytes;
//   * fragments come out of that k-major image with ds_read_b64_tr_b16, the hardware transpose read (4 k x 16 m block per
//     16-lane group, delivered column-major): two reads = the 8 k of one v_mfma_f32_16x16x32_bf16 operand;
//   * image layout (b) of cdna_hip_programming.md T10: 256-byte rows, 16-byte chunk ch of row k at
//     256 k + 16 (ch ^ (((k & 3) << 2) | ((k >> 2) & 3))): conflict-free for these reads.
// Where the step's persistent bf16 kernels have left bf16 IMAGES of an operand (dG: lstm_persist_bwd2.h's Gb; h and
// Dropout(h): lstm_persist.h's Hb / Ub -- the same values, rounded the same way) the operand is staged from the image
// instead (template flags ABF / BBF): half the L2 -> LDS bytes, which bound the f32-sourced form (32 flop per byte at
// this tile size), and no conversion.
// Tile 128 x 128 x 64 per stage, 4 waves as 2 x 2 (64 x 64 each = 4 x 4 MFMA tiles), register-prefetched global loads,
// double-buffered LDS, split-K over blockIdx.z into f32 slabs that k_reduce_slabs sums in z order (deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvqa {

typedef float wb_f32x4 __attribute__((ext_vector_type(4)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));
typedef short wb_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));

struct WgradBf16Args {
    const float *A, *B; // [K][lda], [K][ldb]
    const unsigned short *A16, *B16; // bf16 images of A / B with the same leading dimensions (ABF / BBF instances)
    float *out;         // slab z at out + z * slab_stride, [M][ldo]
    size_t slab_stride;
    int lda, ldb, ldo, M, N, K, kslice; // kslice: K rows per blockIdx.z, a multiple of 64
};

#define NVQA_WB_BM 128
#define NVQA_WB_BN 128
#define NVQA_WB_BK 64
#define NVQA_WB_STAGE_BYTES (2 * NVQA_WB_BK * 256) // A image + B image
#define NVQA_WB_LDS_BYTES (2 * NVQA_WB_STAGE_BYTES)

__device__ __forceinline__ unsigned wb_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <bool ABF, bool BBF>
__global__ __launch_bounds__(256) void k_wgrad_bf16(WgradBf16Args g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wb_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * NVQA_WB_BM, n0 = blockIdx.y * NVQA_WB_BN;
    const int kbeg = blockIdx.z * g.kslice, kend = min(g.K, kbeg + g.kslice);
    // staging, f32 source: thread -> (row = tid / 32 + 8 i, 4 columns 4 (tid % 32) ..), i = 0 .. 7;
    //          bf16 image: thread -> (row = tid / 16 + 16 i, 8 columns 8 (tid % 16) ..), i = 0 .. 3: the 16-byte chunk as it is
    const int srow = tid >> 5, sc4 = tid & 31, irow = tid >> 4, ic8 = tid & 15;
    const bool a_in = ABF ? m0 + 8 * ic8 < g.M : m0 + 4 * sc4 < g.M; // M, N multiples of 4 (f32) / 8 (image)
    const bool b_in = BBF ? n0 + 8 * ic8 < g.N : n0 + 4 * sc4 < g.N;
    const float *pa = g.A + (size_t)(kbeg + srow) * g.lda + m0 + 4 * sc4;
    const float *pb = g.B + (size_t)(kbeg + srow) * g.ldb + n0 + 4 * sc4;
    const unsigned short *pa16 = ABF ? g.A16 + (size_t)(kbeg + irow) * g.lda + m0 + 8 * ic8 : nullptr;
    const unsigned short *pb16 = BBF ? g.B16 + (size_t)(kbeg + irow) * g.ldb + n0 + 8 * ic8 : nullptr;
    wb_u32x4 ra[ABF ? 4 : 8], rb[BBF ? 4 : 8];
    auto load_stage = [&](int k0) {
        if constexpr (ABF) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ra[i] = a_in && k0 + irow + 16 * i < kend ? *reinterpret_cast<const wb_u32x4 *>(pa16 + (size_t)(k0 - kbeg + 16 * i) * g.lda) : wb_u32x4{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ra[i] = a_in && k0 + srow + 8 * i < kend ? *reinterpret_cast<const wb_u32x4 *>(pa + (size_t)(k0 - kbeg + 8 * i) * g.lda) : wb_u32x4{0, 0, 0, 0};
        }
        if constexpr (BBF) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rb[i] = b_in && k0 + irow + 16 * i < kend ? *reinterpret_cast<const wb_u32x4 *>(pb16 + (size_t)(k0 - kbeg + 16 * i) * g.ldb) : wb_u32x4{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                rb[i] = b_in && k0 + srow + 8 * i < kend ? *reinterpret_cast<const wb_u32x4 *>(pb + (size_t)(k0 - kbeg + 8 * i) * g.ldb) : wb_u32x4{0, 0, 0, 0};
        }
    };
    auto pack = [](const wb_u32x4 &u) {
        const wb_f32x4 v = __builtin_bit_cast(wb_f32x4, u);
        const wb_f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
        return wb_u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(lo, wb_bf16x2)),
                        __builtin_bit_cast(unsigned, __builtin_convertvector(hi, wb_bf16x2))};
    };
    auto store_stage = [&](int buf) {
        unsigned char *As = wb_smem + buf * NVQA_WB_STAGE_BYTES, *Bs = As + NVQA_WB_BK * 256;
        if constexpr (ABF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<wb_u32x4 *>(As + wb_off(irow + 16 * i, ic8)) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<wb_u32x2 *>(As + wb_off(srow + 8 * i, sc4 >> 1) + 8u * (sc4 & 1)) = pack(ra[i]);
        }
        if constexpr (BBF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<wb_u32x4 *>(Bs + wb_off(irow + 16 * i, ic8)) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<wb_u32x2 *>(Bs + wb_off(srow + 8 * i, sc4 >> 1) + 8u * (sc4 & 1)) = pack(rb[i]);
        }
    };
    // fragment of the 16 columns starting at c16 (tile-local), k rows kk .. kk+31: lane (li, lh) needs k = kk + 8 lh + j.
    // One transposed read covers a 4 (k) x 16 (columns) block per 16-lane group: lane 4q + p of the group supplies the
    // address of row q, columns 4p .. 4p+3 and receives column (4q + p) of the 4 rows.
    const int q = li >> 2, p = li & 3;
    auto frag = [&](const unsigned char *img, int c16, int kk) -> wb_bf16x8 {
        const int r0 = kk + 8 * lh + q;
        const int ch = (c16 >> 3) + (p >> 1);
        const wb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) wb_s16x4 *)(img + wb_off(r0, ch) + 8u * (p & 1)));
        const wb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) wb_s16x4 *)(img + wb_off(r0 + 4, ch) + 8u * (p & 1)));
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(wb_bf16x8, v);
    };

    wb_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wb_f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = (kend - kbeg + NVQA_WB_BK - 1) / NVQA_WB_BK;
    if (nst > 0) {
        load_stage(kbeg);
        store_stage(0);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int buf = st & 1;
            if (st + 1 < nst) load_stage(kbeg + (st + 1) * NVQA_WB_BK); // in flight under this stage's MFMAs
            const unsigned char *As = wb_smem + buf * NVQA_WB_STAGE_BYTES, *Bs = As + NVQA_WB_BK * 256;
#pragma unroll
            for (int ks = 0; ks < NVQA_WB_BK / 32; ++ks) {
                wb_bf16x8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = frag(As, wm * 64 + 16 * i, 32 * ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = frag(Bs, wn * 64 + 16 * j, 32 * ks);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (st + 1 < nst) store_stage(buf ^ 1); // the other buffer: its readers passed the barrier of the previous stage
            __syncthreads();
        }
    }
    // D[row = 4 lh + r][col = li] of tile (i, j): row = m (A's columns), col = n
    float *out = g.out + (size_t)blockIdx.z * g.slab_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + 16 * i + 4 * lh + r;
                if (m < g.M && n < g.N) out[(size_t)m * g.ldo + n] = acc[i][j][r];
            }
        }
}

} // namespace nvqa
