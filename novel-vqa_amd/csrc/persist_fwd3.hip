// persist_fwd3.hip -- the instances of round 4's direct-operand persistent forward kernel (lstm_persist_fwd3.h), a translation
// unit of their own so that the library builds in parallel (persist_fwd.hip holds the launcher and round 2's LDS-ring kernel).
#include <stdlib.h>

#include "lstm_persist_fwd3.h"
#include "persist_host.h"

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

// shapes the direct-operand kernel has instances for: f32, R = 512, E = 200 or 512, row blocks of 8 row tiles
bool persist_fwd3_eligible(const nvqa_ctx *c, int MT, bool rag)
{
    // The verified set: equal-length batches whose row blocks are full (B a multiple of 128).  The ragged instances are built but
    // NOT correct with E = 200 (losses 5e-5 .. 2e-4 off, different from run to run: DESIGN.md section 4.6), so ragged batches keep
    // lstm_persist.h's kernel, and so do batches with a partly filled last row block (B = 500) until that case has been through the
    // same checks.  NVQA_FWD3_ALL=1 asks for this kernel wherever it has an instance (debugging).
    static const int all_on = [] { const char *e = getenv("NVQA_FWD3_ALL"); return e ? atoi(e) : 0; }();
    const bool full_blocks = !rag && c->d.B % 128 == 0;
    return !c->bf16 && MT == 8 && (full_blocks || all_on) && c->d.R == 512 && (c->d.E == 200 || c->d.E == 512);
}

int launch_persist_fwd3(nvqa_ctx *c, const PersistFwdArgs &a, int grid, bool rag)
{
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
