// persist_fwd3.hip -- the instances of round 4's direct-operand persistent forward kernel (lstm_persist_fwd3.h), a translation
// unit of their own so that the library builds in parallel (persist_fwd.hip holds the launcher and round 2's LDS-ring kernel).
#include <stdlib.h>

#include "lstm_persist_fwd3.h"
#include "persist_host.h"

#ifndef NVQA_FWD3_RAGGED_DEFAULT
#define NVQA_FWD3_RAGGED_DEFAULT 2
#endif

namespace nvqa {

template <int KA, int KR, int TILES, int PD, bool RAG>
static int launch_persist_fwd3_t(nvqa_ctx *c, const PersistFwdArgs &a, int grid)
{
    const size_t lds = persist_fwd3_lds<TILES>();
    static int resident = -1; // per instantiation: workgroups of this kernel one CU can hold
    if (resident < 0) {
        NVQA_HIP(hipFuncSetAttribute((const void *)k_lstm_fwd_persist3<KA, KR, TILES, PD, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        NVQA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lstm_fwd_persist3<KA, KR, TILES, PD, RAG>, NVQA_PF_THREADS, lds));
        resident = nb;
    }
    if (resident < 1 || grid > c->num_cus) { // the workgroups wait for each other: all of them must be resident at once
        set_error("persistent LSTM kernel cannot be co-resident (%d workgroups, %d CUs, %d per CU)", grid, c->num_cus, resident);
        return -1;
    }
    hipLaunchKernelGGL((k_lstm_fwd_persist3<KA, KR, TILES, PD, RAG>), dim3(grid), dim3(NVQA_PF_THREADS), lds, c->s, a);
    NVQA_HIP(hipGetLastError());
    return 0;
}

// How ragged batches (arch1, lengths not all equal) are run -- NVQA_FWD3_RAGGED:
//   2  (default) this kernel's RAG instance: row tiles without active rows skip their loads and MFMAs.  Correct since the skip
//      became a branch INSIDE the asm statements (lstm_persist_fwd3.h, mfma_pair_rag; DESIGN.md section 4.6 has the story of the
//      C++ `if` it replaces); tests/test_gpu_fwd3.py holds it against the ring kernel row by row, the parity suite against the CPU restatement
//   1  this kernel's instance without skips: the cell masks inactive (row, step) slots in every instance, so it is correct on a
//      ragged batch too -- it multiplies all rows (ragged step 2.35 ms against 2.19 ms)
//   0  lstm_persist.h's ring kernel, RAG instance (2.25 ms)
static int ragged_mode()
{
    static const int m = [] {
        const char *e = getenv("NVQA_FWD3_RAGGED");
        return e ? atoi(e) : NVQA_FWD3_RAGGED_DEFAULT;
    }();
    return m;
}

// shapes the direct-operand kernel has instances for: f32, R = 512, E = 200 or 512, row blocks of 8 row tiles.  Any B: a partly
// filled last row block (the reference's default batch of 500) goes through the `li < nloc` masks of the loads and the `grow < B`
// test of the cell (tests/test_gpu_b500.py holds both against the CPU restatement).
bool persist_fwd3_eligible(const nvqa_ctx *c, int MT, bool rag)
{
    return !c->bf16 && MT == 8 && (!rag || ragged_mode() != 0) && c->d.R == 512 && (c->d.E == 200 || c->d.E == 512);
}

int launch_persist_fwd3(nvqa_ctx *c, const PersistFwdArgs &a, int grid, bool rag)
{
    if (rag && ragged_mode() != 2) rag = false;
    if (c->d.E == 200) {
        if (rag) NVQA_TRY((launch_persist_fwd3_t<200, 512, 8, 16, true>(c, a, grid)));
        else NVQA_TRY((launch_persist_fwd3_t<200, 512, 8, 16, false>(c, a, grid)));
    } else {
        if (rag) NVQA_TRY((launch_persist_fwd3_t<512, 512, 8, 16, true>(c, a, grid)));
        else NVQA_TRY((launch_persist_fwd3_t<512, 512, 8, 16, false>(c, a, grid)));
    }
    return 0;
}

} // namespace nvqa
