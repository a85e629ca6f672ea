// latch.h -- the err latch of the persistent LSTM launches as a workgroup body (persist_fwd.hip explains what it does); carried
// by an extra workgroup of kernels.h' k_head_prep / k_arch2_head_prep / k_emb_bwd_tok, or launched alone (k_err_latch).
#pragma once
#include <hip/hip_runtime.h>
#include "nvqa_ctx.h"

namespace nvqa {

__device__ __forceinline__ void err_latch_block(const LatchArgs &a)
{
    if (!a.cnt) return;
    if (threadIdx.x == 0) {
        const unsigned *err = a.cnt + a.words - 4;
        if (err[0] != 0) {
            if (a.sticky[0] == 0) { a.sticky[1] = err[1]; a.sticky[2] = err[2]; a.sticky[3] = err[3]; a.sticky[0] = err[0]; }
            if (a.dp_status) a.dp_status[0] = 1.0f;
        }
        for (int i = 0; i < 4; ++i) __hip_atomic_store(a.host_copy + i, a.sticky[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < a.words; i += blockDim.x) a.cnt[i] = 0;
}

} // namespace nvqa
