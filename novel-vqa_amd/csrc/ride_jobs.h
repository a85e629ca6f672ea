// ride_jobs.h -- work that rides in the workgroups WITHOUT A ROLE of the persistent BPTT launch (lstm_persist_bwd2.h: the grid
// is 8 XCDs x 32 slots of which 240 (f32, L = 2) or 192 (bf16) carry a role).  Everything here depends only on results that
// exist before the BPTT starts and is needed only after it: off the step's critical path, no launch of its own.
//   * the token-segment index of the embedding gradient (tok_index.h), one workgroup;
//   * the head's bias gradients (column sums over the B rows of dscores, dqc, dic), blocks of 64 columns;
//   * tiles of head weight-gradient products dW = dY^T X (arch1: W_o, W_q; single-GPU runs only -- with a communicator
//     the multimodal segment is all-reduced BEFORE the BPTT and these slots belong to the collective's kernels), dealt
//     round-robin to the idle workgroups of one slot group, 64 x 64 tiles, 4 waves (gemm_f32_body with external LDS).
#pragma once
#include "epilogues.h"
#include "gemm_f32.h"
#include "tok_index.h"

#ifndef NVQA_PF_THREADS
#define NVQA_PF_THREADS 256
#endif

namespace nvqa {

// Several SHORT column sums (M = B rows: the head's bias gradients), one stage: block `blk` owns 64 columns of one problem and
// walks all its rows (4 row lanes x 4 independent chains), fixed order.  256 threads; sm: 4 x 64 floats of LDS.
struct ColsumBatch {
    const float *X[4];
    float *out[4];
    int M[4], N[4], ld[4];
    int first_block[5]; // block range of problem p: [first_block[p], first_block[p+1])
};
__device__ __forceinline__ void colsum_batch_block(const ColsumBatch &a, int blk, float *sm)
{
    int p = 0;
    while (p < 3 && blk >= a.first_block[p + 1]) ++p;
    const float *X = a.X[p];
    const int M = a.M[p], N = a.N[p], ld = a.ld[p];
    const int n = (blk - a.first_block[p]) * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = w;
        for (; m + 12 < M; m += 16) {
            s0 += X[(size_t)m * ld + n];
            s1 += X[(size_t)(m + 4) * ld + n];
            s2 += X[(size_t)(m + 8) * ld + n];
            s3 += X[(size_t)(m + 12) * ld + n];
        }
        for (; m < M; m += 4) s0 += X[(size_t)m * ld + n];
    }
    sm[w * 64 + (threadIdx.x & 63)] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && n < N) a.out[p][n] = (sm[threadIdx.x] + sm[64 + threadIdx.x]) + (sm[128 + threadIdx.x] + sm[192 + threadIdx.x]);
}

#define NVQA_RIDE_MAXGEMM 3
typedef Cfg<16, 64, 64, 64, 2, 2, 1, 1> CfgRide; // 256 threads = the persistent kernels' workgroup

struct RideGemm {
    GemmArgs g;   // A_MC x B_NC: dW[M x N] = A^T B, A stored [K][M], B stored [K][N]
    EpiStore e;
    int tx, ty;   // tiles along N / M
};
struct RideJobs {
    TokIndexArgs tok;
    int has_tok, ngemm, has_colsum;
    RideGemm gm[NVQA_RIDE_MAXGEMM];
    ColsumBatch cs;   // has_colsum: the head's bias gradients (first_block[4] blocks of 64 columns)
};

// called by every thread of an idle workgroup; me / n: its index among the idle workgroups that share the jobs
template <bool BF>
__device__ __forceinline__ void ride_jobs_run(const RideJobs *jobs, int me, int n, float *smem)
{
    if (jobs->has_tok && me == 0) {
        tok_index_body<NVQA_PF_THREADS>(reinterpret_cast<unsigned *>(smem), jobs->tok);
        __syncthreads();
    }
    if (jobs->has_colsum) {
        const ColsumBatch cs = jobs->cs;
        for (int blk = n - 1 - me; blk < cs.first_block[4]; blk += n) { // (from the far end: workgroup 0 is still indexing)
            colsum_batch_block(cs, blk, smem);
            __syncthreads();
        }
    }
    const int ngemm = jobs->ngemm;
    for (int p = 0; p < ngemm; ++p) {
        const GemmArgs g = jobs->gm[p].g;
        const EpiStore e = jobs->gm[p].e;
        const int tx = jobs->gm[p].tx, total = tx * jobs->gm[p].ty;
        // (the workgroup that built the index starts later: it takes the tiles from the far end)
        for (int t = me; t < total; t += n) {
            if constexpr (BF) gemm_f32_body<typename WithBF<CfgRide>::type, A_MC, B_NC, false, EpiStore, 0, true>(g, e, t % tx, t / tx, 0, smem);
            else gemm_f32_body<CfgRide, A_MC, B_NC, false, EpiStore, 0, true>(g, e, t % tx, t / tx, 0, smem);
            __syncthreads();
        }
    }
}

} // namespace nvqa
