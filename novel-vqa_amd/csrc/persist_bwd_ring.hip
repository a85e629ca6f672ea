// persist_bwd_ring.hip -- the instances of round 3's LDS-ring persistent BPTT kernel (lstm_persist_bwd2.h: default in bf16,
// NVQA_BWD_KERNEL=2 in f32), a translation unit of their own so that the library builds in parallel (persist_bwd.hip holds
// the launcher and the direct-operand kernel of round 4).
#include <stdlib.h>
#include <algorithm>

#include "lstm_persist_bwd2.h"
#include "persist_host.h"

namespace nvqa {

template <int GKT, int MTA, int MTB, int NTN, int GPC, bool BF, bool RAG>
static int launch_persist_bwd2(nvqa_ctx *c, const PersistBwd2Args &a, int grid)
{
    size_t lds = PersistBwd2Geom<MTA, MTB, NTN, GPC>::LDS_BYTES;
    if (a.jobs && c->ride.has_tok) lds = std::max(lds, tok_index_lds(c->ride.tok.VT, c->ride.tok.NP));
    static int resident = -1;
    if (resident < 0) {
        // (the whole CU's LDS: a launch that carries the token-index job asks for more than the kernel's own layout)
        NVQA_HIP(hipFuncSetAttribute((const void *)k_lstm_bwd_persist2<GKT, MTA, MTB, NTN, GPC, BF, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nb = 0;
        NVQA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lstm_bwd_persist2<GKT, MTA, MTB, NTN, GPC, BF, RAG>, NVQA_PF_THREADS, lds));
        resident = nb;
    }
    if (resident < 1 || grid > c->num_cus) {
        set_error("persistent BPTT kernel cannot be co-resident (%d workgroups, %d CUs, %d per CU)", grid, c->num_cus, resident);
        return -1;
    }
    hipLaunchKernelGGL((k_lstm_bwd_persist2<GKT, MTA, MTB, NTN, GPC, BF, RAG>), dim3(grid), dim3(NVQA_PF_THREADS), lds, c->s, a);
    NVQA_HIP(hipGetLastError());
    return 0;
}

// MT: row tiles per workgroup chosen by persist_bwd_rows (4: two chains of 2; 7: 4 + 3); rag: the ragged instance
int launch_persist_bwd_ring(nvqa_ctx *c, const PersistBwd2Args &a, int grid, int MT, bool rag)
{
#define NVQA_PB2_GO(GKT, MTA, MTB, NTN, GPC, BFv)                                                                   \
    do {                                                                                                             \
        if (rag) NVQA_TRY((launch_persist_bwd2<GKT, MTA, MTB, NTN, GPC, BFv, true>(c, a, grid)));                    \
        else NVQA_TRY((launch_persist_bwd2<GKT, MTA, MTB, NTN, GPC, BFv, false>(c, a, grid)));                       \
    } while (0)
    if (c->bf16) {
        // (4 K groups per chunk -- half the barriers -- measured slower: 0.52 vs 0.45 ms; the step is a chain of latencies)
        if (a.L == 1) NVQA_PB2_GO(16, 2, 2, 2, 2, true); else NVQA_PB2_GO(16, 2, 2, 4, 2, true);
    } else {
        if (MT == 4) NVQA_PB2_GO(32, 2, 2, 2, 2, false); else NVQA_PB2_GO(32, 4, 3, 2, 2, false);
    }
#undef NVQA_PB2_GO
    return 0;
}

} // namespace nvqa
