// tok_index.h -- the token-segment index of the embedding gradient (kernels.h: k_emb_bwd_seg / k_emb_bwd_long): one workgroup
// sorts the packed positions of a batch by token.  A device function templated on the workgroup size, because it runs in
// two places: as its own one-workgroup kernel (k_tok_index, 1024 threads), and -- round 3 -- inside an IDLE workgroup of the
// persistent BPTT launch (256 threads; lstm_persist_bwd2.h): the BPTT grid is 8 XCDs x 32 slots of which 240 (f32, L = 2)
// or 192 (bf16) have a role, the index is needed only by the embedding gradient behind the BPTT, and a kernel of its own
// costs its 18-23 us on the step's critical path (on a side stream it delayed the persistent launches: measured slower).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NVQA_ES_SHORT 32   // occurrences one wave sums by itself
#define NVQA_ES_CHUNKS 16  // a longer segment is cut into this many chunks
#define NVQA_TI_THREADS 1024
#define NVQA_TI_MAXNP 16384 // packed positions: T*B <= 16384
__host__ __device__ inline size_t tok_index_lds(int VT, int NP) { return (size_t)VT * 4 + (size_t)NP * 4 + NVQA_TI_THREADS * 4 + 16; }

struct TokIndexArgs {
    const int32_t *ptok;   // [NP] packed token per position (-1: none); NULL: no job
    int NP, VT;
    int32_t *seg_start;    // [VT+1]
    uint16_t *perm;        // [NP]
    int32_t *long_tok;     // [NP / SHORT + 1]
    unsigned *done;        // [NP / SHORT + 1]
    int32_t *nlong_out;
};

// all THREADS threads of the workgroup call this; smem: tok_index_lds(VT, NP) bytes
template <int THREADS>
__device__ __forceinline__ void tok_index_body(unsigned *ti_smem, const TokIndexArgs &t)
{
    constexpr int NPT = NVQA_TI_MAXNP / THREADS; // packed positions per thread
    static_assert(THREADS % 64 == 0 && THREADS <= 1024 && NVQA_TI_MAXNP % THREADS == 0, "workgroup size");
    const int32_t *ptok = t.ptok;
    const int NP = t.NP, VT = t.VT;
    int32_t *seg_start = t.seg_start;
    uint16_t *perm = t.perm;
    int32_t *long_tok = t.long_tok;
    unsigned *done = t.done;
    int32_t *nlong_out = t.nlong_out;
    unsigned *cnt = ti_smem;              // [VT] counts -> exclusive prefix sums = placement cursors -> segment ENDS
    unsigned *part = cnt + VT;            // [threads] scan partials
    unsigned *nlong = part + NVQA_TI_THREADS;
    unsigned *tmp = nlong + 4;            // [NP] (token << 16 | position), grouped by token, unordered inside a group
    const int tid = threadIdx.x;
    int w[NPT]; // this thread's tokens: one batch of independent loads
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int k = tid + i * THREADS;
        const int x = k < NP ? ptok[k] : -1;
        w[i] = x >= 0 && x < VT ? x : -1;
    }
    for (int v = tid; v < VT; v += THREADS) cnt[v] = 0;
    if (tid == 0) *nlong = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPT; ++i)
        if (w[i] >= 0) atomicAdd(&cnt[w[i]], 1u);
    __syncthreads();
    // exclusive scan: thread -> a contiguous run of tokens
    const int per = (VT + THREADS - 1) / THREADS, v0 = tid * per, v1 = min(VT, v0 + per);
    unsigned sum = 0;
    for (int v = v0; v < v1; ++v) sum += cnt[v];
    // inclusive scan of the partials: inside each wave by shuffles, then the wave totals
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned x = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += x;
    }
    if ((tid & 63) == 63) part[tid >> 6] = inc;
    __syncthreads();
    if (tid < 64) {
        const unsigned t0 = tid < THREADS / 64 ? part[tid] : 0u;
        unsigned tt = t0;
#pragma unroll
        for (int o = 1; o < THREADS / 64; o <<= 1) {
            const unsigned x = __shfl_up(tt, o, 64);
            if (tid >= o) tt += x;
        }
        if (tid < THREADS / 64) part[64 + tid] = tt - t0; // exclusive prefix of the wave totals
        if (tid == THREADS / 64 - 1) part[128] = tt;      // grand total
    }
    __syncthreads();
    unsigned run = part[64 + (tid >> 6)] + inc - sum;
    for (int v = v0; v < v1; ++v) {
        const unsigned n = cnt[v];
        cnt[v] = run;
        seg_start[v] = (int)run;
        if (n > NVQA_ES_SHORT) { // (which slot a token gets does not matter)
            const unsigned slot = atomicAdd(nlong, 1u);
            long_tok[slot] = v;
            done[slot] = 0;
        }
        run += n;
    }
    const unsigned total = part[128];
    if (tid == THREADS - 1) seg_start[VT] = (int)total;
    __syncthreads();
    if (tid == 0) *nlong_out = (int)*nlong;
#pragma unroll
    for (int i = 0; i < NPT; ++i)
        if (w[i] >= 0) tmp[atomicAdd(&cnt[w[i]], 1u)] = ((unsigned)w[i] << 16) | (unsigned)(tid + i * THREADS);
    __syncthreads();
    // order inside a group: rank of each position among its group (groups are short, or a few long ones); cnt[v] is now
    // the END of token v's group, so its start is the end of the group before it
    for (unsigned p = tid; p < total; p += THREADS) {
        const unsigned x = tmp[p], v = x >> 16;
        const unsigned s = v ? cnt[v - 1] : 0u, e = cnt[v];
        unsigned rank = 0;
        for (unsigned q = s; q < e; ++q) rank += tmp[q] < x ? 1u : 0u;
        perm[s + rank] = (uint16_t)(x & 0xffffu);
    }
}
