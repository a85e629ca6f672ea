// persist_fwd.hip -- launcher of the persistent forward LSTM kernel (lstm_persist.h), its own translation unit.
#include <stdlib.h>
#include <string.h>

#include "lstm_persist.h"
#include "persist_host.h"
#include "latch.h"

namespace nvqa {

// The err record of a launch lives in the last 16 bytes of its counter block.  Each launch is followed by a latch, which
// (1) copies the FIRST failure into a sticky record that only the host clears, once it has reported it (check_persist;
// k_rmsprop refuses to apply gradients while a record is set; dp_status[0] (data parallel) is summed over the ranks at the
// end of the step, so that every rank skips the update if any rank's kernel gave up), (2) writes the sticky record
// straight into the host's pinned copy (a 16-byte D2H copy is a 5 us blit kernel of its own), and (3) ZEROES the counter
// block for the next launch (the memset in front of every launch was another 5 us fill kernel).  The blocks are zeroed
// once at creation.  Round 4: the latch is no longer a launch of its own (2 x 5 us per step at the launch floor) -- it is
// the LAST workgroup of a kernel that follows the persistent launch anyway (kernels.h: err_latch_block, carried by
// k_head_prep / k_arch2_head_prep behind the forward launch and by k_emb_bwd_tok behind the BPTT launch); where no
// carrier comes (forward-only entry points on a fallback route, ...) latch_flush launches k_err_latch.
__global__ void k_err_latch(LatchArgs a) { err_latch_block(a); }

int persist_latch_err(nvqa_ctx *c, unsigned *cnt, size_t words, int off)
{
    LatchArgs &l = c->latch_pending[off / 4];
    if (l.cnt) NVQA_TRY(latch_flush(c)); // (an older record still waiting for a carrier: latch it now)
    l.cnt = cnt; l.words = (unsigned)words; l.sticky = c->pf_sticky + off;
    l.dp_status = c->comm ? c->dp_status + c->dp_slot : nullptr; l.host_copy = c->h_pf_err + off;
    return 0;
}
LatchArgs latch_take(nvqa_ctx *c, int which)
{
    const LatchArgs l = c->latch_pending[which];
    c->latch_pending[which] = LatchArgs{};
    return l;
}
int latch_flush(nvqa_ctx *c)
{
    for (int w = 0; w < 2; ++w) {
        if (!c->latch_pending[w].cnt) continue;
        hipLaunchKernelGGL(k_err_latch, dim3(1), dim3(256), 0, c->s, latch_take(c, w));
        NVQA_HIP(hipGetLastError());
    }
    return 0;
}

// The whole forward unroll as one persistent, weight-stationary launch (lstm_persist.h).  Eligible shapes: the two the
// reference trains (R = 512 with E = 200 [arch1] or E = 512 [arch2]) on a device with one CU per workgroup.
template <int KA, int KR, int MT, bool BF, bool RAG>
static int launch_persist_fwd(nvqa_ctx *c, const PersistFwdArgs &a, int grid)
{
    const size_t lds = persist_fwd_lds<KA, KR, MT, BF>();
    static int resident = -1; // per instantiation: workgroups of this kernel one CU can hold
    if (resident < 0) {
        NVQA_HIP(hipFuncSetAttribute((const void *)k_lstm_fwd_persist<KA, KR, MT, BF, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        NVQA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lstm_fwd_persist<KA, KR, MT, BF, RAG>, NVQA_PF_THREADS, lds));
        resident = nb;
    }
    if (resident < 1 || grid > c->num_cus) { // the workgroups wait for each other: all of them must be resident at once
        set_error("persistent LSTM kernel cannot be co-resident (%d workgroups, %d CUs, %d per CU)", grid, c->num_cus, resident);
        return -1;
    }
    hipLaunchKernelGGL((k_lstm_fwd_persist<KA, KR, MT, BF, RAG>), dim3(grid), dim3(NVQA_PF_THREADS), lds, c->s, a);
    NVQA_HIP(hipGetLastError());
    return 0;
}

int persist_rows(const nvqa_ctx *c) // row tiles of 16 per workgroup (MT), or 0 when the path does not apply
{
    const nvqa_dims &d = c->d;
    if (!c->persist_on || d.R != 512 || !(d.E == 200 || d.E == 512) || d.L > NVQA_PF_MAXL) return 0;
    // (arch2 and equal-length arch1 batches: every row is active whenever any is; ragged arch1 batches: the RAG instance)
    const int NU = d.R / 16;
    for (int MT : {4, 8}) { // the smallest row block that still gives every workgroup its own CU
        const int RB = (d.B + 16 * MT - 1) / (16 * MT);
        if (d.L * RB * NU <= c->num_cus) return MT;
    }
    return 0;
}

// step-0 slices of the bf16 images of Hs (zeros, or the carried h0 of NVQA_QUIRK_H0): layer = blockIdx.y
__global__ void k_h0_image(const float *Hs, unsigned short *Hb, size_t layer_stride, int n)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const float2 v = *reinterpret_cast<const float2 *>(Hs + blockIdx.y * layer_stride + i);
    *reinterpret_cast<unsigned *>(Hb + blockIdx.y * layer_stride + i) = pf_pack_bf16(v.x, v.y);
}

// NVQA_QUIRK_H0 + bf16: arch2_backward copies THIS step's dL/dh into slice 0 of Hs[L-1] (the aliased tensor the step-1
// clone's accGradParameters reads, Encoder_lstm.lua:238-239); the bf16 image of that slice, which the weight-gradient kernel
// stages from, was made at forward time from the previous step's gradient: refresh it.
int persist_reimage_h0_top(nvqa_ctx *c)
{
    if (!c->act_b16 || !c->img_fwd_valid) return 0;
    const nvqa_dims &d = c->d;
    const size_t hs = (size_t)(c->TS + 1) * d.B * d.R;
    hipLaunchKernelGGL(k_h0_image, dim3((d.B * d.R / 2 + 255) / 256, 1), dim3(256), 0, c->s, c->Hs[d.L - 1], c->act_b16 + (size_t)(d.L - 1) * hs, hs, d.B * d.R);
    NVQA_HIP(hipGetLastError());
    return 0;
}

int lstm_forward_persist(nvqa_ctx *c, const Drop &dr, int MT)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, R = d.R, L = d.L, TS = c->TS;
    PersistFwdArgs a = {};
    if (c->bf16) { // bf16 images of Hs and U, [L][(TS+1)*B][R] + [L][TS*B][R] (first use of the bf16 instance)
        const size_t hs = (size_t)(TS + 1) * B * R, us = (size_t)TS * B * R;
        if (!c->act_b16) NVQA_HIP(hipMalloc((void **)&c->act_b16, (size_t)L * (hs + us) * 2));
        for (int l = 0; l < L; ++l) { a.Hb[l] = c->act_b16 + l * hs; a.Ub[l] = c->act_b16 + L * hs + l * us; }
        // layer 0's input segment from the bf16 image of X0 the embedding kernel left (E = R: the same instance as the layers above)
        a.Ub[0] = d.E == R && c->x0_img_valid ? c->x0_b16 : nullptr;
        // step-0 slices of the images: zeros like Hs' own (written once), or the carried h0 of NVQA_QUIRK_H0 (every step)
        const bool carried = d.arch == NVQA_ARCH2 && (c->quirks & NVQA_QUIRK_H0);
        if (carried || !c->h0_img_clean) {
            hipLaunchKernelGGL(k_h0_image, dim3((B * R / 2 + 255) / 256, L), dim3(256), 0, c->s, c->Hs[0], c->act_b16, hs, B * R);
            NVQA_HIP(hipGetLastError());
            c->h0_img_clean = !carried;
        }
    }
    for (int l = 0; l < L; ++l) {
        a.Wi[l] = c->P + c->lo.w_i2h[l]; a.Wh[l] = c->P + c->lo.w_h2h[l];
        a.bi[l] = c->P + c->lo.b_i2h[l]; a.bh[l] = c->P + c->lo.b_h2h[l];
        a.U[l] = c->U[l]; a.Hs[l] = c->Hs[l]; a.Cs[l] = c->Cs[l]; a.Gt[l] = c->Gt[l];
    }
    a.X0 = c->X0; a.nrows = c->nrows; a.sort_idx = c->sort_idx;
    a.B = B; a.R = R; a.E = d.E; a.L = L; a.TS = TS;
    a.RB = (B + 16 * MT - 1) / (16 * MT); a.NU = R / 16;
    a.h0_top = d.arch == NVQA_ARCH2 && (c->quirks & NVQA_QUIRK_H0) ? 1 : 0;
    a.dr = dr;
    { static const int dbg = [] { const char *e = getenv("NVQA_PF_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
    a.spin_limit = c->pf_spin ? c->pf_spin : NVQA_PF_SPIN_LIMIT;
    a.cnt = c->pf_cnt; a.err = c->pf_cnt + c->pf_cnt_words - 4; // the last 16 bytes of the block
    a.ts = c->pf_ts;
    c->fwd_ride_done = false;
    if (c->fwd_ride_pending && !c->bf16) { // (arch1_forward: the head's image projection, for the layer-0 workgroups' idle tail)
        a.fr = c->fwd_ride;
        a.fr_on = 1;
        c->fwd_ride_done = true;
        if (c->prof_on) {
            c->prof[PF_RIDE].flops += 2.0 * c->fwd_ride.g.M * c->fwd_ride.g.N * c->fwd_ride.g.K;
            c->prof[PF_RIDE].launches += 1;
        }
    }
    const int grid = L * a.RB * a.NU;
    double flops = 0;
    for (int l = 0; l < L; ++l) flops += 2.0 * B * 4 * R * ((double)TS * (l == 0 ? d.E : R) + (double)(TS - 1) * R);
    ProfScope ps(c, PF_LSTM_FWD, flops, 0);
    // ragged arch1 batch (or lengths known only on the device: the dataset route of a ragged dataset): the instance that
    // skips the MFMAs of row tiles without active rows
    const bool rag = d.arch == NVQA_ARCH1 && !c->batch_uniform;
    // NVQA_FWD_KERNEL: 1 = round 2's LDS-ring kernel (lstm_persist.h), 3 = round 4's direct-operand kernel (lstm_persist_fwd3.h;
    // default where it has an instance: f32, row blocks of 8 row tiles, equal-length batches)
    static const int fwd_kernel = [] { const char *e = getenv("NVQA_FWD_KERNEL"); return e ? atoi(e) : 0; }();
    if (fwd_kernel != 1 && persist_fwd3_eligible(c, MT, rag)) {
        NVQA_TRY(launch_persist_fwd3(c, a, grid, rag));
        NVQA_TRY(persist_latch_err(c, c->pf_cnt, c->pf_cnt_words, 0));
        return 0;
    }
#define NVQA_PF_GO(KA, MTv, BFv, RAGv) NVQA_TRY((launch_persist_fwd<KA, 512, MTv, BFv, RAGv>(c, a, grid)))
    if (d.E == 200) { // arch1
        if (c->bf16) {
            if (rag) { if (MT == 4) NVQA_PF_GO(200, 4, true, true); else NVQA_PF_GO(200, 8, true, true); }
            else { if (MT == 4) NVQA_PF_GO(200, 4, true, false); else NVQA_PF_GO(200, 8, true, false); }
        } else {
            if (rag) { if (MT == 4) NVQA_PF_GO(200, 4, false, true); else NVQA_PF_GO(200, 8, false, true); }
            else { if (MT == 4) NVQA_PF_GO(200, 4, false, false); else NVQA_PF_GO(200, 8, false, false); }
        }
    } else if (rag) { // (an arch1 model with E = 512)
        if (c->bf16) { if (MT == 4) NVQA_PF_GO(512, 4, true, true); else NVQA_PF_GO(512, 8, true, true); }
        else { if (MT == 4) NVQA_PF_GO(512, 4, false, true); else NVQA_PF_GO(512, 8, false, true); }
    } else {
        if (c->bf16) { if (MT == 4) NVQA_PF_GO(512, 4, true, false); else NVQA_PF_GO(512, 8, true, false); }
        else { if (MT == 4) NVQA_PF_GO(512, 4, false, false); else NVQA_PF_GO(512, 8, false, false); }
    }
#undef NVQA_PF_GO
    NVQA_TRY(persist_latch_err(c, c->pf_cnt, c->pf_cnt_words, 0));
    return 0;
}

} // namespace nvqa
