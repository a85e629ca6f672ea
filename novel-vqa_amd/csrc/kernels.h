// kernels.h -- the HBM-bound (non-GEMM) kernels of the VQA step: batch assembly,
// embedding gather, head preparation, softmax cross-entropy, deterministic embedding
// gradient, column sums, clamp+RMSprop.  64-wide wavefronts, coalesced 16-byte accesses
// where the row width allows, no atomics on floats (results are bit-reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include "ride_jobs.h"
#include "../../include/nvqa_layout.h"
#include "epilogues.h"
#include "gemm_f32.h"
#include "latch.h"

namespace nvqa {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// four f32 -> four bf16 (round to nearest even), 8 bytes: a piece of a row of the bf16 image of the layer-0 inputs (bf16 mode: the
// weight-gradient kernel and, where E = R, the persistent forward kernel read the image instead of rounding f32 rows themselves)
__device__ __forceinline__ void store_bf16x4(unsigned short *dst, const float4 &v)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint2 o;
    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v.x, v.y}, bf2));
    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v.z, v.w}, bf2));
    *reinterpret_cast<uint2 *>(dst) = o;
}

// W [rows][cols] f32 -> its transpose [cols][rows] in bf16 (round to nearest even), 32 x 32 tiles through LDS: workgroup `blk` of
// ceil(cols / 32) x ceil(rows / 32), 256 threads.  bf16 mode: the image of W_i2h[0]^T that the gfx950-form d(layer-0 input) product
// multiplies by (nvqa_api.hip: lstm_dx0); the weights do not move during a step, so it rides as extra workgroups of the
// embedding launch at the start of the step instead of being a launch of its own in front of the product.
struct TransposeJob {
    const float *W = nullptr; // nullptr: no job
    unsigned short *out = nullptr;
    int rows = 0, cols = 0, first_block = 0, nblocks = 0;
};
__device__ __forceinline__ void transpose_to_bf16_block(const TransposeJob &j, int blk, float (*t)[33])
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, nbx = (j.cols + 31) / 32;
    const int c0 = (blk % nbx) * 32, r0 = (blk / nbx) * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, cc = c0 + tx;
        t[i][tx] = r < j.rows && cc < j.cols ? j.W[(size_t)r * j.cols + cc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int cc = c0 + i, r = r0 + tx;
        if (cc < j.cols && r < j.rows) reinterpret_cast<__bf16 *>(j.out)[(size_t)cc * j.rows + r] = (__bf16)t[tx][i];
    }
}

// ---------------------------------------------------------------------------------
// dataset:next_batch() gather (002_train_baseline.lua:202-210): rows qinds of the
// HBM-resident dataset -> batch buffers.  One block per sample.
// ---------------------------------------------------------------------------------
// The sample ids travel as KERNEL ARGUMENTS (nvqa_step_indices, B <= NVQA_QARG_MAX: 32-bit ids, 3 KB of the 4 KB kernarg
// segment) -- round 3 copied them with a 4 KB H2D blit in front of every step, a 4.6 us launch of its own; reading them from
// pinned host memory inside the kernel (one PCIe read per workgroup) measured 11 us SLOWER.  Larger batches: qinds in device memory.
// (NVQA_QARG_MAX, struct QIdxArg: nvqa_ctx.h)
template <bool ARG>
__global__ void k_gather_batch(QIdxArg qa, const int64_t *qinds, const int32_t *Q, const int32_t *QL,
                               const int32_t *IP, const int32_t *ANS, const float *F, int T, int I,
                               int32_t *tok, int32_t *len, int32_t *lab, float *img)
{
    const int b = blockIdx.x;
    const int64_t q = ARG ? (int64_t)qa.q[b] : qinds[b];
    for (int t = threadIdx.x; t < T; t += blockDim.x) tok[(size_t)b * T + t] = Q[q * T + t];
    if (threadIdx.x == 0) {
        len[b] = QL ? QL[q] : T;
        lab[b] = ANS[q];
    }
    const float4 *src = reinterpret_cast<const float4 *>(F + (size_t)(IP[q] - 1) * I);
    float4 *dst = reinterpret_cast<float4 *>(img + (size_t)b * I);
    for (int i = threadIdx.x; i < I / 4; i += blockDim.x) dst[i] = src[i];
}

// row L2 normalisation (002_train_baseline.lua:117-121), one wave per row, in place
// (columns [col0, col0 + I) of rows of width ld: early fusion normalises two blocks separately,
// 003_train_ae_based_ef.lua:115-119)
__global__ void k_l2norm_rows(float *F, int64_t n, int ld, int col0, int I)
{
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float4 *p = reinterpret_cast<float4 *>(F + row * ld + col0);
    float s = 0.f;
    for (int i = lane; i < I / 4; i += 64) {
        const float4 v = p[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = sqrtf(wave_sum(s));
    for (int i = lane; i < I / 4; i += 64) {
        float4 v = p[i];
        v.x /= s; v.y /= s; v.z /= s; v.w /= s;
        p[i] = v;
    }
}

// fc7 features -> row L2 norm (002_train_baseline.lua:117-121) -> the step's image-feature buffer
// (BASELINE config 5: the extractor fused in front of the training step).  One wave per row.
__global__ void k_l2norm_copy(const float *src, int n, int I, float *dst)
{
    const int row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float4 *p = reinterpret_cast<const float4 *>(src + (size_t)row * I);
    float4 *q = reinterpret_cast<float4 *>(dst + (size_t)row * I);
    float s = 0.f;
    for (int i = lane; i < I / 4; i += 64) {
        const float4 v = p[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = sqrtf(wave_sum(s));
    for (int i = lane; i < I / 4; i += 64) {
        float4 v = p[i];
        v.x /= s; v.y /= s; v.z /= s; v.w /= s;
        q[i] = v;
    }
}

// ---------------------------------------------------------------------------------
// sort_encoding_onehot_right_align (misc/RNNUtils.lua:84-124) without the one-hot:
// stable descending counting sort of the lengths, inverse permutation, and the number of
// active rows per time column.  Single workgroup (B is a few thousand at most).
// ---------------------------------------------------------------------------------
__global__ void k_sort_lengths(const int32_t *len, int B, int T, int32_t *sort_idx, int32_t *sort_inv,
                               int32_t *nrows /*[T]*/)
{
    // hist[T+1], start[T+1], run[T+1] (rows of each length placed by earlier passes), wcnt[waves][T+1] (rows of each length
    // in each wave of this pass).  The stable rank of row b among the rows of its length = rows of that length in earlier
    // passes + in earlier waves of this pass + in lower lanes of its own wave (one ballot per length value: T + 1 ballots
    // instead of round 3's loop over every earlier row -- 9 us -> the launch floor).
    extern __shared__ int sm[];
    const int TL = T + 1, nw = blockDim.x / 64, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int *hist = sm, *start = sm + TL, *run = sm + 2 * TL, *wcnt = sm + 3 * TL;
    for (int i = threadIdx.x; i < 3 * TL; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += blockDim.x) {
        const int b = b0 + threadIdx.x;
        const int l = b < B ? min(max(len[b], 0), T) : -1;
        int inwave = 0;
        for (int v = 0; v <= T; ++v) {
            const unsigned long long m = __ballot(l == v);
            if (l == v) inwave = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[w * TL + v] = __popcll(m);
        }
        __syncthreads();
        int before = 0;
        if (l >= 0) {
            for (int ww = 0; ww < w; ++ww) before += wcnt[ww * TL + l];
            before += run[l] + inwave;
        }
        __syncthreads();
        if ((int)threadIdx.x <= T) {
            int tot = 0;
            for (int ww = 0; ww < nw; ++ww) tot += wcnt[ww * TL + threadIdx.x];
            run[threadIdx.x] += tot;
        }
        // (rank relative to the first row of its length; the bucket starts are known only after the last pass)
        if (l >= 0) sort_inv[b] = before;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int l = T; l >= 0; --l) { hist[l] = run[l]; start[l] = acc; acc += run[l]; }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int l = min(max(len[b], 0), T);
        const int pos = start[l] + sort_inv[b];
        sort_idx[pos] = b;
        sort_inv[b] = pos;
    }
    // column t (0-based) is active for rows with len >= T - t
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        int n = 0;
        for (int l = T - t; l <= T; ++l) n += hist[l];
        nrows[t] = n;
    }
}

// ---------------------------------------------------------------------------------
// arch1 embedding forward: x = tanh(Dropout(W_e[:,tok] + b_e)) (002_train_baseline.lua:141-144).
// The reference multiplies a dense one-hot [sum(len) x V] by W_e^T; every product but one per
// row is an exact zero, so this is a row gather from the transposed table WeT [V][E].
// One wave per packed row (t, r); writes zeros for inactive rows and records the token.
// ---------------------------------------------------------------------------------
__global__ void k_emb_fwd(const int32_t *tok, const int32_t *sort_idx, const int32_t *nrows,
                          const float *WeT, const float *be, int B, int T, int E, Drop dr,
                          float *X, int32_t *ptok, unsigned short *Xb /* bf16 image of X, or NULL */, TransposeJob tj,
                          const float *img, float *vd, int I, int vd_first_block)
{
    if (tj.W && (int)blockIdx.x >= tj.first_block) {
        __shared__ float tt[32][33];
        transpose_to_bf16_block(tj, blockIdx.x - tj.first_block, tt);
        return;
    }
    if (vd && (int)blockIdx.x >= vd_first_block) {
        // Dropout on the image feature (netdef.lua:11), row b = blockIdx.x - vd_first_block: k_head_prep's second half, done here --
        // before the LSTM -- when the image projection rides in the persistent forward launch (lstm_persist.h) and needs it early
        const int b = blockIdx.x - vd_first_block;
        for (int j = 4 * threadIdx.x; j < I; j += 4 * blockDim.x) {
            const uint64_t idx = (uint64_t)b * I + j;
            const float4 v = *reinterpret_cast<const float4 *>(img + idx);
            *reinterpret_cast<float4 *>(vd + idx) =
                make_float4(dr.scale(NVQA_SITE_V, idx) * v.x, dr.scale(NVQA_SITE_V, idx + 1) * v.y,
                            dr.scale(NVQA_SITE_V, idx + 2) * v.z, dr.scale(NVQA_SITE_V, idx + 3) * v.w);
        }
        return;
    }
    const int row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (row >= T * B) return;
    const int t = row / B, r = row % B;
    float4 *x4 = reinterpret_cast<float4 *>(X + (size_t)row * E);
    unsigned short *xb = Xb ? Xb + (size_t)row * E : nullptr;
    if (r >= nrows[t]) {
        for (int i = lane; i < E / 4; i += 64) {
            x4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xb) *reinterpret_cast<uint2 *>(xb + 4 * i) = make_uint2(0u, 0u);
        }
        if (lane == 0) ptok[row] = -1;
        return;
    }
    const int b = sort_idx[r];
    const int w = tok[(size_t)b * T + t] - 1;
    if (lane == 0) ptok[row] = w;
    const float4 *w4 = reinterpret_cast<const float4 *>(WeT + (size_t)w * E);
    const float4 *b4 = reinterpret_cast<const float4 *>(be);
    const uint64_t base = ((uint64_t)b * T + t) * E;
    for (int i = lane; i < E / 4; i += 64) {
        const float4 wv = w4[i], bv = b4[i];
        float4 o;
        o.x = tanhf_(dr.scale(NVQA_SITE_EMB, base + 4 * i + 0) * (wv.x + bv.x));
        o.y = tanhf_(dr.scale(NVQA_SITE_EMB, base + 4 * i + 1) * (wv.y + bv.y));
        o.z = tanhf_(dr.scale(NVQA_SITE_EMB, base + 4 * i + 2) * (wv.z + bv.z));
        o.w = tanhf_(dr.scale(NVQA_SITE_EMB, base + 4 * i + 3) * (wv.w + bv.w));
        x4[i] = o;
        if (xb) store_bf16x4(xb + 4 * i, o);
    }
}

// ---------------------------------------------------------------------------------
// head preparation: question vector [c1 h1 c2 h2 ...] un-sorted + Dropout
// (002_train_baseline.lua:306, misc/LSTM.lua:70, netdef.lua:10) and Dropout on the
// image feature (netdef.lua:11).  blockIdx.x = sample b.
// ---------------------------------------------------------------------------------
__global__ void k_head_prep(const float *Cfin /*[L][B][R]*/, const float *Hfin, size_t lstride,
                            const int32_t *sort_inv, const float *img, int B, int R, int L, int I,
                            Drop dr, float *qd, float *vd, LatchArgs latch)
{
    if ((int)blockIdx.x == B) { err_latch_block(latch); return; } // the extra workgroup: the forward launch's err latch (latch.h)
    const int b = blockIdx.x, r = sort_inv[b];
    const int Q = 2 * R * L;
    // 16-byte accesses (R % 4 == 0, I % 4 == 0 are preconditions of nvqa_create): 4 consecutive j stay inside
    // one (layer, c | h) run of R values
    for (int j = 4 * threadIdx.x; j < Q; j += 4 * blockDim.x) {
        const int l = j / (2 * R), part = (j / R) & 1, u = j % R;
        const float4 v = *reinterpret_cast<const float4 *>((part ? Hfin : Cfin) + (size_t)l * lstride + (size_t)r * R + u);
        const uint64_t idx = (uint64_t)b * Q + j;
        *reinterpret_cast<float4 *>(qd + idx) =
            make_float4(dr.scale(NVQA_SITE_Q, idx) * v.x, dr.scale(NVQA_SITE_Q, idx + 1) * v.y,
                        dr.scale(NVQA_SITE_Q, idx + 2) * v.z, dr.scale(NVQA_SITE_Q, idx + 3) * v.w);
    }
    if (!vd) return; // (made by the embedding launch already: the image projection rode in the forward launch)
    for (int j = 4 * threadIdx.x; j < I; j += 4 * blockDim.x) {
        const uint64_t idx = (uint64_t)b * I + j;
        const float4 v = *reinterpret_cast<const float4 *>(img + idx);
        *reinterpret_cast<float4 *>(vd + idx) =
            make_float4(dr.scale(NVQA_SITE_V, idx) * v.x, dr.scale(NVQA_SITE_V, idx + 1) * v.y,
                        dr.scale(NVQA_SITE_V, idx + 2) * v.z, dr.scale(NVQA_SITE_V, idx + 3) * v.w);
    }
}

// ---------------------------------------------------------------------------------
// nn.CrossEntropyCriterion forward+backward in one pass (002_train_baseline.lua:157,308-310):
// row max / sum in registers + wave shuffles, loss_b = lse - s[y], dscores = (softmax - onehot)/B.
// One wave per row.  argmax (first maximal index, 1-based) optional.
// ---------------------------------------------------------------------------------
__global__ void k_softmax_ce(const float *scores, const int32_t *labels, int B, int A, float *dscores,
                             float *rowloss, int32_t *argmax, float *h_rowloss)
{
    const int b = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float *s = scores + (size_t)b * A;
    // The row lives in registers (up to 16 values per lane: A <= 1024, the reference's 1000 answers): one read of the
    // scores instead of three dependent passes over them.  Same arithmetic in the same order as the general loops below
    // (per-lane strided partial sums, then the wave butterfly): bit-identical.
    constexpr int RV = 16;
    const bool small = A <= 64 * RV;
    float rv[RV];
    float mx = -INFINITY;
    int am = 0x7fffffff;
    if (small) {
#pragma unroll
        for (int j = 0; j < RV; ++j) rv[j] = lane + 64 * j < A ? s[lane + 64 * j] : -INFINITY;
#pragma unroll
        for (int j = 0; j < RV; ++j)
            if (rv[j] > mx) { mx = rv[j]; am = lane + 64 * j; }
    } else {
        for (int a = lane; a < A; a += 64) {
            const float v = s[a];
            if (v > mx) { mx = v; am = a; }
        }
    }
    const float wmx = wave_max(mx);
    if (argmax) {
        int cand = (mx == wmx) ? am : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) argmax[b] = cand + 1;
    }
    if (!labels) return;
    float sum = 0.f;
    if (small) {
#pragma unroll
        for (int j = 0; j < RV; ++j)
            if (lane + 64 * j < A) sum += expf(rv[j] - wmx);
    } else {
        for (int a = lane; a < A; a += 64) sum += expf(s[a] - wmx);
    }
    sum = wave_sum(sum);
    const float lse = wmx + logf(sum);
    const int y = labels[b] - 1;
    if (lane == 0) {
        const float rl = lse - s[y];
        rowloss[b] = rl;
        // The batch mean is taken by the HOST when it asks for the loss (nvqa_api.hip: loss_mean_host, the fixed order round 3's
        // one-workgroup k_loss_mean used: bit-identical): the row goes straight into the pinned host array -- a posted PCIe write --
        // instead of a second launch (4.7 us at the launch floor) and a 4-byte D2H blit (4.6 us) behind this kernel.
        if (h_rowloss) __hip_atomic_store(h_rowloss + b, rl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (dscores) {
        const float invB = 1.0f / (float)B;
        if (small) {
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                const int a = lane + 64 * j;
                if (a < A) dscores[(size_t)b * A + a] = (expf(rv[j] - lse) - (a == y ? 1.0f : 0.0f)) * invB;
            }
        } else {
            for (int a = lane; a < A; a += 64)
                dscores[(size_t)b * A + a] = (expf(s[a] - lse) - (a == y ? 1.0f : 0.0f)) * invB;
        }
    }
}

// multiple-choice answer (004_eval_model.lua:259-271): among the non-zero candidate ids of a row, the one with the
// highest score; ties go to the earlier slot (torch.max over the candidates in slot order).  One wave per row,
// n_mc <= 64 candidates (the reference's MC_ans_test has 18).
__global__ void k_mc_argmax(const float *scores, const int32_t *mc, int n, int A, int n_mc, int32_t *out)
{
    const int b = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (b >= n) return;
    float v = -INFINITY;
    int slot = 0x7fffffff;
    if (lane < n_mc) {
        const int a = mc[(size_t)b * n_mc + lane];
        if (a >= 1 && a <= A) { v = scores[(size_t)b * A + (a - 1)]; slot = lane; }
    }
    const float mx = wave_max(v);
    int cand = (slot != 0x7fffffff && v == mx) ? slot : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) out[b] = cand == 0x7fffffff ? 0 : mc[(size_t)b * n_mc + cand];
}

// ---------------------------------------------------------------------------------
// column sums (bias gradients: nn.Linear accGradParameters, gradBias += colsum(dY)).
// Stage 1: grid (N/64, S): block sums its row stripe for 64 columns -> part[s][n].
// Stage 2: sums the S partials in order.  Deterministic.
// (Round 4 tried ONE launch, the last workgroup of a column block adding the partials behind an agent-scope release /
// acquire pair: the two __threadfence() of every workgroup -- an L2 write-back and an invalidate on a chip whose L2s are
// full of the GEMMs' dirty lines -- cost more than the second launch: 33 -> 40 us per step.  Two launches stay.)
// ---------------------------------------------------------------------------------
__global__ void k_colsum_part(const float *X, int M, int N, int ld, int rows_per_split, float *part)
{
    __shared__ float sm[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int mbeg = blockIdx.y * rows_per_split, mend = min(M, mbeg + rows_per_split);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f; // 4 independent chains keep 4+ loads in flight
    if (n < N) {
        int m = mbeg + w;
        for (; m + 12 < mend; m += 16) {
            s0 += X[(size_t)m * ld + n];
            s1 += X[(size_t)(m + 4) * ld + n];
            s2 += X[(size_t)(m + 8) * ld + n];
            s3 += X[(size_t)(m + 12) * ld + n];
        }
        for (; m < mend; m += 4) s0 += X[(size_t)m * ld + n];
    }
    sm[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && n < N) part[(size_t)blockIdx.y * N + n] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
__global__ void k_colsum_final(const float *part, int S, int N, float *out, float *out2)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 3 < S; i += 4) {
        s0 += part[(size_t)i * N + n];
        s1 += part[(size_t)(i + 1) * N + n];
        s2 += part[(size_t)(i + 2) * N + n];
        s3 += part[(size_t)(i + 3) * N + n];
    }
    for (; i < S; ++i) s0 += part[(size_t)i * N + n];
    const float s = (s0 + s1) + (s2 + s3);
    out[n] = s;
    if (out2) out2[n] = s;
}

// Several SHORT column sums (M = B rows: the head's bias gradients) in one launch and one stage: a block owns
// 64 columns of one problem and walks all its rows (4 row lanes x 4 independent chains), fixed order.
// (struct ColsumBatch and the block body colsum_batch_block live in ride_jobs.h: the same body also runs inside the idle
// workgroups of the persistent BPTT launch.)
__global__ void k_colsum_batch(ColsumBatch a)
{
    __shared__ float sm[4][64];
    colsum_batch_block(a, blockIdx.x, &sm[0][0]);
}

// sum of split-K slabs (fixed order) with optional accumulate into C
__global__ void k_reduce_slabs(const float *slabs, int S, size_t n4, float4 *C)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 *p = reinterpret_cast<const float4 *>(slabs);
    float4 s = p[i];
    for (int z = 1; z < S; ++z) {
        const float4 v = p[(size_t)z * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    C[i] = s;
}

// ---------------------------------------------------------------------------------
// arch1 embedding backward (002_train_baseline.lua:319-320): d(pre) = Dropout' (tanh' dX);
// dW_e[:,tok] += d(pre) summed in ascending packed order (bit-reproducible, no atomics).
// Each wave owns 16 consecutive vocabulary rows, scans the packed token list once and
// accumulates its rows in LDS; writes every row of its range (zeros included), so the
// gradient table needs no memset.  The reference forms this as a dense
// [E x sum(len)] x [sum(len) x V] GEMM over the one-hot matrix.
// ---------------------------------------------------------------------------------
#define NVQA_EB_ROWS 16
#define NVQA_EB_COLS 128 // columns per workgroup: LDS = waves x 16 x 128 floats = 32 KB whatever E is
__global__ void k_emb_bwd(const int32_t *ptok, const float *X, const float *dX, const int32_t *sort_idx,
                          int NP /*T*B*/, int B, int T, int V, int E, Drop dr, float *dWeT /*[V][E]*/,
                          int plain /* 1: nn.LookupTable (arch2): the row gradient is dX itself */)
{
    // A workgroup owns 16 vocabulary rows x 128 columns (blockIdx.y = column block; with E = 512 one block per
    // row range needed 128 KB of LDS, one workgroup per CU, and the 512 START-token rows of arch2 went through
    // one wave 8 columns-of-64 at a time: 1.49 ms).  Its waves scan consecutive quarters of the packed token
    // list into private LDS accumulators, which are then summed in wave order: a fixed summation order
    // (ascending packed position within a wave, waves in order), so still bit-reproducible.
    extern __shared__ float acc[]; // [waves][16][EC]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int v0 = blockIdx.x * NVQA_EB_ROWS;
    const int e0 = blockIdx.y * NVQA_EB_COLS, EC = min(NVQA_EB_COLS, E - e0);
    float *my = acc + (size_t)wave * NVQA_EB_ROWS * NVQA_EB_COLS;
    for (int i = lane; i < NVQA_EB_ROWS * NVQA_EB_COLS; i += 64) my[i] = 0.f;
    const int per = ((NP + waves - 1) / waves + 63) / 64 * 64;
    const int kbeg = wave * per, kend = min(NP, kbeg + per);
    const bool c0 = lane < EC, c1 = lane + 64 < EC; // this lane's two columns of the block
    auto fetch = [&](int kk, float &a0, float &a1) {
        const size_t g = (size_t)kk * E + e0 + lane;
        a0 = c0 ? dX[g] : 0.f;
        a1 = c1 ? dX[g + 64] : 0.f;
        if (!plain) {
            const int t = kk / B, r = kk % B;
            const uint64_t base = ((uint64_t)sort_idx[r] * T + t) * E + e0 + lane;
            if (c0) { const float x = X[g]; a0 = dr.scale(NVQA_SITE_EMB, base) * (a0 * (1.0f - x * x)); }
            if (c1) { const float x = X[g + 64]; a1 = dr.scale(NVQA_SITE_EMB, base + 64) * (a1 * (1.0f - x * x)); }
        }
    };
    // the token list is read eight 64-token chunks at a time (independent loads in flight: a wave's scan is
    // a chain of L2 round trips otherwise and almost every chunk has no hit for these 16 rows)
    for (int kb = kbeg; kb < kend; kb += 8 * 64) {
      int wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
          const int k = kb + 64 * j + lane;
          wv[j] = k < kend ? ptok[k] : -1;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k0 = kb + 64 * j;
        const int w = wv[j];
        unsigned long long hit = __ballot(w >= v0 && w < v0 + NVQA_EB_ROWS);
        if (!hit) continue;
        // hits in ascending packed order, eight at a time: their rows are requested together (a frequent word's
        // chain of hits is latency-bound otherwise) and added in order
        while (hit) {
            int src[8];
            float a0[8], a1[8];
            int n = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                src[i] = -1;
                if (hit) {
                    src[i] = __ffsll((long long)hit) - 1;
                    hit &= hit - 1;
                    fetch(k0 + src[i], a0[i], a1[i]);
                    ++n;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i >= n) break;
                const int row = __shfl(w, src[i], 64) - v0;
                my[row * NVQA_EB_COLS + lane] += a0[i];
                my[row * NVQA_EB_COLS + lane + 64] += a1[i];
            }
        }
      }
    }
    __syncthreads();
    const int nr = min(NVQA_EB_ROWS, V - v0);
    for (int i = threadIdx.x; i < nr * NVQA_EB_COLS; i += blockDim.x) {
        const int row = i / NVQA_EB_COLS, col = i % NVQA_EB_COLS;
        if (col >= EC) continue;
        float s = 0.f;
        for (int wv = 0; wv < waves; ++wv) s += acc[(size_t)wv * NVQA_EB_ROWS * NVQA_EB_COLS + i];
        dWeT[(size_t)(v0 + row) * E + e0 + col] = s;
    }
}

// ---------------------------------------------------------------------------------
// Embedding / lookup-table gradient by token segments (the default; k_emb_bwd above remains for vocabularies whose
// index does not fit one workgroup's LDS).  k_emb_bwd has every workgroup scan the whole packed token list (V / 16 x 4
// scans of T*B tokens: 0.14 ms on arch2 whatever the batch) and sums a frequent word's rows in one wave, 8 at a time
// (arch2's START token has B = 512 of them in EVERY batch; real questions add "what", "is", "the").  Here
//   k_tok_index   (one workgroup, on a side stream under the forward pass) sorts the packed positions by token:
//                 seg_start[v] .. seg_start[v+1] delimit token v's positions in perm[], ascending;
//   k_emb_bwd_seg (one wave per token) sums the rows of its positions in that order;
//   k_emb_bwd_long: a token with more than NVQA_ES_SHORT occurrences is cut into NVQA_ES_CHUNKS equal chunks summed by
//                 different waves into partial rows; the wave that arrives last (agent-scope counter) adds the partials
//                 in chunk order.
// The summation order is fixed by perm and the chunk boundaries, so the result is bit-reproducible; no atomics on data.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NVQA_TI_THREADS) void k_tok_index(TokIndexArgs t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned ti_smem[];
    tok_index_body<NVQA_TI_THREADS>(ti_smem, t);
}

// rows of positions [lo, hi) of perm, summed in that order: 8 rows in flight, both column passes of a lane together
template <int NPASS>
__device__ __forceinline__ void emb_seg_sum(const uint16_t *perm, int lo, int hi, const float *X, const float *dX, const int32_t *sort_idx,
                                            int B, int T, int E, const Drop &dr, int plain, int lane, float4 (&acc)[NPASS])
{
    const int E4 = E / 4;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = lo; i0 < hi; i0 += 64) { // the positions of up to 64 rows: one coalesced load, then broadcasts
        const int mine = i0 + lane < hi ? (int)perm[i0 + lane] : 0;
        const int nb = min(64, hi - i0);
        for (int j0 = 0; j0 < nb; j0 += 8) {
            float4 a[8][NPASS];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = __shfl(mine, min(j0 + j, 63), 64);
                const bool on = j0 + j < nb;
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
                    const int c4 = lane + 64 * p;
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on && c4 < E4) {
                        const size_t g = (size_t)k * E + 4 * c4;
                        d = *reinterpret_cast<const float4 *>(dX + g);
                        if (!plain) { // arch1: through Dropout and Tanh of the embedding (002_train_baseline.lua:319-320)
                            const float4 x = *reinterpret_cast<const float4 *>(X + g);
                            const int t = k / B, r = k % B;
                            const uint64_t base = ((uint64_t)sort_idx[r] * T + t) * E + 4 * c4;
                            d.x = dr.scale(NVQA_SITE_EMB, base) * (d.x * (1.0f - x.x * x.x));
                            d.y = dr.scale(NVQA_SITE_EMB, base + 1) * (d.y * (1.0f - x.y * x.y));
                            d.z = dr.scale(NVQA_SITE_EMB, base + 2) * (d.z * (1.0f - x.z * x.z));
                            d.w = dr.scale(NVQA_SITE_EMB, base + 3) * (d.w * (1.0f - x.w * x.w));
                        }
                    }
                    a[j][p] = d;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int p = 0; p < NPASS; ++p) { acc[p].x += a[j][p].x; acc[p].y += a[j][p].y; acc[p].z += a[j][p].z; acc[p].w += a[j][p].w; }
        }
    }
}

// ONE launch for the whole gradient (round 3: k_emb_bwd_seg + k_emb_bwd_long, two launches):
//   workgroups [0, gs): tokens with at most NVQA_ES_SHORT occurrences (absent ones included: their row is zero), one wave per token;
//   workgroups [gs, gs + gl): the frequent tokens: wave (slot, chunk) sums its chunk into a partial row; the last of a token's
//     NVQA_ES_CHUNKS waves to arrive adds the partials in chunk order; slots beyond *nlong leave at once;
//   workgroup gs + gl (if present): the err latch of the persistent BPTT launch that ran before (latch.h).
template <int NPASS>
__global__ __launch_bounds__(256) void k_emb_bwd_tok(const int32_t *seg_start, const uint16_t *perm, const int32_t *long_tok, const int32_t *nlong,
                                                     unsigned *done, float *partial /*[slots][CHUNKS][E]*/, unsigned partial_bytes, const float *X,
                                                     const float *dX, const int32_t *sort_idx, int B, int T, int VT, int E, Drop dr,
                                                     float *dWeT /*[VT][E]*/, int plain, int gs, int gl, LatchArgs latch)
{
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= gs + gl) { err_latch_block(latch); return; }
    if ((int)blockIdx.x < gs) {
        const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (v >= VT) return;
        const int s0 = seg_start[v], n = seg_start[v + 1] - s0;
        if (n > NVQA_ES_SHORT) return; // a frequent token: the second block range
        float4 acc[NPASS];
        emb_seg_sum<NPASS>(perm, s0, s0 + n, X, dX, sort_idx, B, T, E, dr, plain, lane, acc);
        float4 *row = reinterpret_cast<float4 *>(dWeT + (size_t)v * E);
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
            if (lane + 64 * p < E / 4) row[lane + 64 * p] = acc[p];
        return;
    }
    const int wv = ((int)blockIdx.x - gs) * 4 + (threadIdx.x >> 6);
    const int slot = wv / NVQA_ES_CHUNKS, ch = wv % NVQA_ES_CHUNKS;
    if (slot >= *nlong) return;
    const int v = long_tok[slot];
    const int s0 = seg_start[v], n = seg_start[v + 1] - s0;
    const int len = (n + NVQA_ES_CHUNKS - 1) / NVQA_ES_CHUNKS;
    const int lo = min(s0 + n, s0 + ch * len), hi = min(s0 + n, lo + len);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r_part = __builtin_amdgcn_make_buffer_rsrc(partial, 0, (int)partial_bytes, 0x00020000);
    float4 acc[NPASS];
    emb_seg_sum<NPASS>(perm, lo, hi, X, dX, sort_idx, B, T, E, dr, plain, lane, acc);
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
        if (lane + 64 * p < E / 4)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(acc[p].x), __float_as_uint(acc[p].y), __float_as_uint(acc[p].z), __float_as_uint(acc[p].w)},
                                                   r_part, (unsigned)(((size_t)(slot * NVQA_ES_CHUNKS + ch) * E + 4 * (lane + 64 * p)) * 4), 0, 16 /* sc1 */);
    // hand-off of the partial rows (MI355X_MICROARCH.md "Valid forms": sc1 write-through stores, drained by the storing
    // wave, then ONE agent-scope add by one of its lanes; the wave whose add came last -- told by the value returned --
    // reads all partials with sc1 buffer loads after that add has returned)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(done + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != NVQA_ES_CHUNKS - 1) return;
    float4 *row = reinterpret_cast<float4 *>(dWeT + (size_t)v * E);
    for (int c4 = lane; c4 < E / 4; c4 += 64) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int cc = 0; cc < NVQA_ES_CHUNKS; ++cc) {
            const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(r_part, (unsigned)(((size_t)(slot * NVQA_ES_CHUNKS + cc) * E + 4 * c4) * 4), 0, 16 /* sc1 */);
            sum.x += __uint_as_float(x[0]); sum.y += __uint_as_float(x[1]); sum.z += __uint_as_float(x[2]); sum.w += __uint_as_float(x[3]);
        }
        row[c4] = sum;
    }
}

// ---------------------------------------------------------------------------------
// arch2 (003_train_vqa_arch2/misc/Encoder_lstm.lua:152-227): the encoder runs T+2 steps;
// step 0 = projected image, step 1 = START token (row V of the lookup table), step t >= 2 =
// token t-2 with null (0) rewritten to token 1 (:197).  It stops at the first all-null time
// row (:185-189): tmax = number of steps run.  One workgroup.
// ---------------------------------------------------------------------------------
#define NVQA_ARCH2_TMAX 256
__global__ void k_arch2_tmax(const int32_t *tok, int B, int T, int32_t *nrows /*[T+2]*/, int32_t *tinfo /*[2]: tmax, tmax-1*/,
                             int32_t *sort_idx, int32_t *sort_inv)
{
    __shared__ int colany[NVQA_ARCH2_TMAX]; // T <= NVQA_ARCH2_TMAX is checked by nvqa_create
    for (int t = threadIdx.x; t < T; t += blockDim.x) colany[t] = 0;
    __syncthreads();
    // (eight loads in flight per thread: the one-at-a-time loop made this one-workgroup kernel a chain of B T / 1024 global
    // round trips -- 8 us for the bench shape)
    const int n = B * T, stride = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * stride) {
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = i0 + j * stride < n ? tok[i0 + j * stride] : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (v[j] != 0) colany[(i0 + j * stride) % T] = 1; // benign race: every writer stores 1
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) { sort_idx[b] = b; sort_inv[b] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tmax = 2;
        for (int t = 0; t < T; ++t) {
            if (!colany[t]) break;
            tmax = t + 3;
        }
        tinfo[0] = tmax;
        tinfo[1] = tmax - 1;
        for (int t = 0; t < T + 2; ++t) nrows[t] = t < tmax ? B : 0;
    }
}

// lookup-table gather for steps >= 1 (Encoder_lstm.lua:177-203); step 0 rows are written by the
// cnn_projection GEMM.  One wave per (step, sample) row.
__global__ void k_arch2_embed(const int32_t *tok, const int32_t *tinfo, const float *Wlk, int B, int T, int V, int E,
                              float *X, int32_t *ptok, unsigned short *Xb /* bf16 image of X, or NULL */, TransposeJob tj)
{
    if (tj.W && (int)blockIdx.x >= tj.first_block) {
        __shared__ float tt[32][33];
        transpose_to_bf16_block(tj, blockIdx.x - tj.first_block, tt);
        return;
    }
    const int row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    const int TS = T + 2;
    if (row >= TS * B) return;
    const int t = row / B, b = row % B;
    float4 *x4 = reinterpret_cast<float4 *>(X + (size_t)row * E);
    unsigned short *xb = Xb ? Xb + (size_t)row * E : nullptr;
    if (t == 0) { // the projected image: written by the projection GEMM in front of this launch; its bf16 image is made here
        if (lane == 0) ptok[row] = -1;
        if (xb)
            for (int i = lane; i < E / 4; i += 64) store_bf16x4(xb + 4 * i, x4[i]);
        return;
    }
    if (t >= tinfo[0]) {
        for (int i = lane; i < E / 4; i += 64) {
            x4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xb) *reinterpret_cast<uint2 *>(xb + 4 * i) = make_uint2(0u, 0u);
        }
        if (lane == 0) ptok[row] = -1;
        return;
    }
    int w = V; // START = token V+1
    if (t >= 2) {
        w = tok[(size_t)b * T + (t - 2)];
        w = (w == 0 ? 1 : w) - 1;
    }
    if (lane == 0) ptok[row] = w;
    const float4 *w4 = reinterpret_cast<const float4 *>(Wlk + (size_t)w * E);
    for (int i = lane; i < E / 4; i += 64) {
        const float4 v = w4[i];
        x4[i] = v;
        if (xb) store_bf16x4(xb + 4 * i, v);
    }
}

// head input: Dropout(h^L at step tmax) (003_.../002_train_baseline.lua:162-164, Encoder_lstm.lua:224)
__global__ void k_arch2_head_prep(const float *Htop /*[(TS+1)*B][R]*/, const int32_t *tinfo, int B, int R, Drop dr, float *hd, LatchArgs latch)
{
    if ((int)blockIdx.x == B) { err_latch_block(latch); return; } // the extra workgroup: the forward launch's err latch (latch.h)
    const int b = blockIdx.x;
    const float *src = Htop + ((size_t)tinfo[0] * B + b) * R;
    for (int j = threadIdx.x; j < R; j += blockDim.x)
        hd[(size_t)b * R + j] = dr.scale(NVQA_SITE_Q, (uint64_t)b * R + j) * src[j];
}

// ---------------------------------------------------------------------------------
// AxB fusion finisher (misc/netdef.lua:10-12 + Dropout of 002_train_baseline.lua:153): the two
// M = B projections W_q Dropout(q), W_v Dropout(v) run as ONE split-K multi-problem launch into
// slabs (each alone has only 128 tiles); here: slab sums + bias, tanh, qc (*) ic, Dropout.
// ---------------------------------------------------------------------------------
__global__ void k_head_fuse(const float *sq, const float *sv, int Z, int Zv, size_t n, int C, const float *bq,
                            const float *bv, Drop dr, float *qc, float *ic, float *zd, int askip)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f, b = 0.f;
    if (Zv == Z) {
        for (int z = 0; z < Z; ++z) { a += sq[(size_t)z * n + i]; b += sv[(size_t)z * n + i]; }
    } else { // the image projection came whole from the forward launch's riding workgroups (Zv = 1)
        for (int z = 0; z < Z; ++z) a += sq[(size_t)z * n + i];
        for (int z = 0; z < Zv; ++z) b += sv[(size_t)z * n + i];
    }
    const int c = (int)(i % C);
    const float q = tanhf_(a + bq[c]), v = tanhf_(b + bv[c]);
    qc[i] = q;
    ic[i] = v;
    if (askip == 2) { // netdef.A_B: zd = Dropout([qc | ic]), 2C wide
        const size_t b = i / C, o = b * 2 * C + c;
        zd[o] = dr.scale(NVQA_SITE_Z, o) * q;
        zd[o + C] = dr.scale(NVQA_SITE_Z, o + C) * v;
        return;
    }
    zd[i] = dr.scale(NVQA_SITE_Z, i) * (askip ? q + q * v : q * v); // netdef.AskipB : netdef.AxB
}

// ---------------------------------------------------------------------------------
// BPTT level finisher.  The recurrent products of one wavefront level are computed as split-K
// GEMMs with 64x64 tiles into partial slabs (a 512x512 output is too small to fill 256 CUs with
// tiles that have a useful arithmetic intensity otherwise: the fused 32x32-tile kernel was bound
// by the ~70 GB/s each CU can pull from L2); this kernel sums the slabs in order and applies the
// fused cell backward of EpiLstmBwd.  blockIdx.y = problem (layer).
// ---------------------------------------------------------------------------------
struct BwdFinish {
    EpiLstmBwd e[NVQA_MAX_LAYERS];
    const float *srec[NVQA_MAX_LAYERS]; // [Z][B][R] partials of dG_{s+1} W_h2h, or NULL
    const float *sup[NVQA_MAX_LAYERS];  // [Z][B][R] partials of dG^{l+1}_s W_i2h^{l+1}, or NULL
    int Z, B, R;
    int xcd2d;  // != 0: element -> block map that follows the products' (row half, column quarter) XCD map
    int zadapt; // != 0: the products chose their number of K slices from nrows (gemm_f32.h zsplit_for(tiles, Z, zadapt))
};
__global__ void k_lstm_bwd_finish(BwdFinish a)
{
    const int p = blockIdx.y;
    const size_t n = (size_t)a.B * a.R;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.xcd2d) {
        // the level products were dealt (row half, column quarter) per XCD (gemm_f32.h, xcd == 3): block b runs on XCD
        // b % 8 and takes its elements from that XCD's region, so the slabs come out of the L2 they were written through
        const unsigned k = blockIdx.x % 8, local = blockIdx.x / 8, qc = a.R / 4, hr = a.B / 2; // region: hr rows x qc columns
        const unsigned e = local * blockDim.x + threadIdx.x;                                    // element inside the region
        if (e >= hr * qc) return;
        i = (size_t)((k / 4) * hr + e / qc) * a.R + (k % 4) * qc + e % qc;
    }
    if (i >= n) return;
    const int m = (int)(i / a.R), u = (int)(i % a.R);
    const EpiLstmBwd &e = a.e[p];
    if (m >= *e.nrows) { // a question that has not started at this step: zeros, without loading its (stale) operands
        const size_t gi = (size_t)m * 4 * e.R + u;
        e.gates[gi] = 0.f; e.gates[gi + e.R] = 0.f; e.gates[gi + 2 * e.R] = 0.f; e.gates[gi + 3 * e.R] = 0.f;
        e.dc[(size_t)m * e.R + u] = 0.f;
        return;
    }
    const EpiLstmBwd::Pre q = e.preload(m, u);
    float v = 0.f, v2 = 0.f;
    if (m < q.nr) {
        const int Z = a.zadapt ? zsplit_for((min(q.nr, a.B) + 63) / 64, (a.B + 63) / 64, a.Z, a.zadapt) : a.Z; // 64 = CfgBwdLevel::BM
        if (a.srec[p])
            for (int z = 0; z < Z; ++z) v += a.srec[p][(size_t)z * n + i];
        if (a.sup[p])
            for (int z = 0; z < Z; ++z) v2 += a.sup[p][(size_t)z * n + i];
    }
    e(0, m, u, v, v2, q);
}

// ---------------------------------------------------------------------------------
// gradients:clamp(-c,c) + optim.rmsprop in one pass (002_train_baseline.lua:329,408;
// misc/rmsprop_lrscale.lua:16-34).  gscale = 1/world for the data-parallel mean.
// 20 B/parameter of HBM traffic (read g,m,x; write m,x).
// ---------------------------------------------------------------------------------
// LSTM bias gradients from the row blocks' partial column sums (fixed order); both bias vectors of a layer receive the sum
__global__ void k_bias_sum(const float *part /*[RB][N]*/, int RB, int N, float *out, float *out2)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += part[(size_t)r * N + n];
    out[n] = s;
    out2[n] = s;
}

// skip (single process): the sticky err records of the persistent LSTM kernels (8 words, persist_fwd.hip: k_err_latch).
// dp_skip (data parallel): the number of ranks whose kernel gave up in THIS step, all-reduced -- the same number on every
// rank.  With a communicator it is the ONLY thing consulted (the launcher passes skip = nullptr): a rank-local sticky record
// stays set until that rank's host looks, which would make the rank that timed out keep skipping steps the other ranks
// apply.  While either is set the step's gradients are invalid and nothing is applied; the host reports the
// failed step at its next synchronisation point -- h_dp[1] (pinned) keeps the count of the LAST failed step until the host
// clears it, so a later good step does not hide the report on the ranks whose own kernels were fine.
__global__ void k_rmsprop(float4 *x, const float4 *g, float4 *m, size_t n4, float lr, float alpha,
                          float eps, float wd, float clamp, float gscale, const unsigned *skip, const float *dp_skip, float *h_dp,
                          float *dp_next)
{
    // (data parallel: the status word the NEXT step will use is cleared here -- the two words alternate -- instead of by a 5 us
    // fill kernel at the head of every step)
    if (dp_next && blockIdx.x == 0 && threadIdx.x == 0) *dp_next = 0.f;
    if (skip && (skip[0] | skip[4]) != 0u) return;
    if (dp_skip && dp_skip[0] != 0.f) {
        if (h_dp && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(h_dp + 1, dp_skip[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const float om = 1.0f - alpha;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        float4 xv = x[i], gv = g[i], mv = m[i];
        float *xp = &xv.x, *gp = &gv.x, *mp = &mv.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float gi = gp[c] * gscale;
            if (clamp > 0.f) gi = fminf(fmaxf(gi, -clamp), clamp);
            if (wd != 0.f) gi += wd * xp[c];
            const float mi = alpha * mp[c] + om * gi * gi;
            mp[c] = mi;
            xp[c] += -lr * gi / (sqrtf(mi) + eps);
        }
        x[i] = xv;
        m[i] = mv;
    }
}

__global__ void k_clamp_copy(const float *g, float *out, size_t n, float clamp, float gscale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = g[i] * gscale;
        if (clamp > 0.f) v = fminf(fmaxf(v, -clamp), clamp);
        out[i] = v;
    }
}


// sum of squares of x[0..n) in double, one partial per workgroup of 256 threads (nvqa_param_norms)
__global__ void k_sumsq(const float *x, size_t n, double *part)
{
    __shared__ double sh[256];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (double)x[i] * (double)x[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void k_fill(float *p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

} // namespace nvqa
