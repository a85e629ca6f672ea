// persist_host.h -- host-side entry points of the persistent LSTM kernels.  The kernels and their launchers live in
// their own translation units (persist_fwd.hip: lstm_persist.h; persist_bwd.hip: lstm_persist_bwd2.h)
// so that the library builds in parallel; nvqa_api.hip only calls these.
#pragma once
#include "nvqa_ctx.h"
#include "prof.h"

namespace nvqa {

// row tiles of 16 per workgroup of the persistent forward kernel (MT), or 0 when the path does not apply
int persist_rows(const nvqa_ctx *c);
// the whole forward unroll as one launch (lstm_persist.h)
int lstm_forward_persist(nvqa_ctx *c, const Drop &dr, int MT);

// the direct-operand forward kernel's instances (persist_fwd3.hip; lstm_persist_fwd3.h)
struct PersistFwdArgs;
bool persist_fwd3_eligible(const nvqa_ctx *c, int MT, bool rag);
int launch_persist_fwd3(nvqa_ctx *c, const PersistFwdArgs &a, int grid, bool rag);

// bf16 + NVQA_QUIRK_H0: refresh the bf16 image of the top layer's step-0 hidden state after arch2_backward rewrote it
int persist_reimage_h0_top(nvqa_ctx *c);
// copies a launch's err record (device) into the sticky record at word `off` (0 forward, 4 BPTT) and that to the host
int persist_latch_err(nvqa_ctx *c, unsigned *cnt, size_t words, int off); // after every persistent launch: its err latch now waits for a carrier kernel
LatchArgs latch_take(nvqa_ctx *c, int which); // a kernel that follows the launch (0 forward, 1 BPTT) carries the latch as an extra workgroup ...
int latch_flush(nvqa_ctx *c);                 // ... or, where none does, k_err_latch runs it alone

// BPTT as one launch: row tiles per workgroup (0: not eligible / switched off), row blocks in *RB
int persist_bwd_rows(const nvqa_ctx *c, int *RB);
int lstm_backward_persist(nvqa_ctx *c, const Drop &dr, int MT, int RB);
// the LDS-ring kernel's instances (persist_bwd_ring.hip)
struct PersistBwd2Args;
int launch_persist_bwd_ring(nvqa_ctx *c, const PersistBwd2Args &a, int grid, int MT, bool rag);
// words of the counter block c->pb_cnt must hold (both kernels), for nvqa_create
size_t persist_bwd_counter_words(const nvqa_dims &d, int TS);

} // namespace nvqa
