// nvqa_ctx.h -- library-owned state behind the opaque nvqa_ctx of include/nvqa.h.
#pragma once
#include <hip/hip_runtime.h>
#include "ride_jobs.h"
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/nvqa.h"
#include "../../include/nvqa_layout.h"
#include "epilogues.h"

namespace nvqa {

void set_error(const char *fmt, ...);

#define NVQA_HIP(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            nvqa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                           \
        }                                                                                        \
    } while (0)

#define NVQA_TRY(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != 0) return rc_;      \
    } while (0)

// Kernel groups timed by HIP events (nvqa_profile_*); one entry per launch site family.
enum ProfId {
    PF_ASSEMBLE = 0,
    PF_EMB_FWD,
    PF_GEMM_I2H,   // time-batched input projections (forward)
    PF_LSTM_FWD,   // recurrent step: h2h GEMM + fused cell
    PF_HEAD_PREP,
    PF_GEMM_HEAD_FWD,
    PF_SOFTMAX_CE,
    PF_GEMM_HEAD_BWD,
    PF_LSTM_BWD,   // BPTT level: split-K products dgates x W_h2h (and x W_i2h of the layer above)
    PF_GEMM_DGRAD, // time-batched d(input) products
    PF_GEMM_WGRAD, // time-batched weight gradients (split-K)
    PF_REDUCE,     // split-K slab sums
    PF_COLSUM,     // bias gradients
    PF_EMB_BWD,
    PF_RMSPROP,
    PF_ALLREDUCE,
    PF_GATHER,
    PF_LSTM_BWD_FIN, // slab sum + fused cell backward of one BPTT level
    PF_RIDE,         // FLOPs of the head products that ride in the persistent BPTT launch's idle workgroups (no time of their own: inside PF_LSTM_BWD's launch)
    PF_COUNT
};

// sample ids of nvqa_step_indices as kernel arguments of k_gather_batch (kernels.h)
#define NVQA_QARG_MAX 768
struct QIdxArg { int32_t q[NVQA_QARG_MAX]; };

// The err record of a persistent launch is latched (copied to the sticky record and the host, its counter block zeroed) by an
// extra workgroup of a kernel that runs later in the step anyway, not by a launch of its own (persist_fwd.hip).
struct LatchArgs {
    unsigned *cnt = nullptr; // nullptr: nothing to latch
    unsigned words = 0;
    unsigned *sticky = nullptr;
    float *dp_status = nullptr;
    unsigned *host_copy = nullptr;
};

struct ProfEntry {
    double ms = 0, flops = 0, bytes = 0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Dataset {
    int64_t n_q = 0, n_img = 0;
    int32_t *Q = nullptr, *QL = nullptr, *IP = nullptr, *ANS = nullptr;
    float *F = nullptr;
    bool uniform_len = false; // every question has the same length (host check at load)
    float mean_len_frac = 1.f; // mean question length / T (host, at load)
};

} // namespace nvqa

struct nvqa_ctx {
    nvqa_dims d;
    nvqa_layout lo;
    int device = 0;
    hipStream_t s = nullptr;                     // compute stream
    hipEvent_t evStart = nullptr;                // hand-off from the extractor's stream (nvqa_step_images)
    hipStream_t sx = nullptr;                    // side stream: the token-segment index of the embedding gradient, under the forward pass
    hipEvent_t evTok = nullptr, evIdx = nullptr; // ptok written / index ready
    bool tok_seg = true;                         // NVQA_EMB_SEG=0: the scanning kernel (k_emb_bwd) instead
    // ride-along jobs of this step (ride_jobs.h), waiting for the persistent BPTT launch to carry them in its idle workgroups
    nvqa::RideJobs ride = {}, ride_dev_host = {}; // the list, and what the device copy holds
    nvqa::RideJobs *ride_dev = nullptr;
    bool tok_job_pending = false, ride_gemm_pending = false;
    bool ride_gemm_on = true, tok_in_bptt_on = true; // NVQA_RIDE_GEMM / NVQA_TOK_IN_BPTT (read by nvqa_create)
    // arch1, f32: the head's image projection rides in the layer-0 workgroups of the persistent FORWARD launch (lstm_persist.h)
    bool ride_fwd_on = true;                      // NVQA_RIDE_FWD (read by nvqa_create)
    bool fwd_ride_pending = false, fwd_ride_done = false;
    nvqa::RideGemm fwd_ride = {};
    int32_t *seg_start = nullptr, *pslot = nullptr;
    uint16_t *perm = nullptr;
    unsigned *seg_done = nullptr;
    float *seg_part = nullptr;
    size_t seg_part_bytes = 0;
    hipStream_t sc = nullptr;                    // communication stream (RCCL all-reduce buckets)
    hipEvent_t evSeg[3 + NVQA_MAX_LAYERS] = {}, evComm = nullptr; // gradient range ready (3 segments + per-layer slices) / exchange done
    int TS = 0; // recurrent steps: arch1 T, arch2 T+2

    // parameters / gradients / RMSprop mean-square (internal layout = ABI layout with the
    // arch1 embedding table transposed to [V][E])
    float *P = nullptr, *G = nullptr, *M2 = nullptr;
    bool have_grads = false;
    int quirks = 0;                   // nvqa_set_ref_quirks (arch2)
    int fusion_askip = 0;             // 0 netdef.AxB, 1 netdef.AskipB, 2 netdef.A_B (W_o [A x 2C])
    bool pristine = true;             // no parameters set yet: the layout may still change (nvqa_set_fusion)
    float gscale[3] = {1.f, 1.f, 1.f}; // per-segment gradient scale before the clamp (-lr_scale)

    // current batch.  tok / len / lab / img point at one of two device sets: the host-batch entries (nvqa_step, nvqa_forward,
    // nvqa_evaluate) stage the caller's arrays in pinned memory and copy them on the side stream into the set the running
    // step does NOT read, so the JdJ-shaped entry neither synchronises nor serialises its copies with the compute stream
    struct BatchSet { int32_t *tok = nullptr, *len = nullptr, *lab = nullptr; float *img = nullptr; };
    BatchSet bset[2], hset[2];        // device sets / pinned host staging (set 1 and the staging: first host-batch call)
    int bcur = 0;                     // the set tok / len / lab / img point at
    hipEvent_t evCopied[2] = {}, evBatchFree[2] = {}; // H2D of set p complete / the last step that read set p has run
    bool batch_free_rec[2] = {false, false}, copied_rec[2] = {false, false};
    int32_t *tok = nullptr, *len = nullptr, *lab = nullptr;
    float *img = nullptr;
    int64_t *qinds = nullptr;
    nvqa::QIdxArg qarg;                      // this step's sample ids on their way into the kernarg segment
    nvqa::LatchArgs latch_pending[2];        // err latches of this step's persistent launches (0 forward, 1 BPTT) waiting for a carrier
    int32_t *sort_idx = nullptr, *sort_inv = nullptr, *nrows = nullptr, *ptok = nullptr;
    int32_t *tinfo = nullptr; // arch2: {tmax, tmax-1}

    // activations
    float *X0 = nullptr, *dX0 = nullptr;       // [TS*B][E] layer-0 inputs and their gradient
    float *Gt[NVQA_MAX_LAYERS] = {};           // [TS*B][4R] gates (fwd) / d(pre-activations) (bwd)
    float *Hs[NVQA_MAX_LAYERS] = {};           // [(TS+1)*B][R]
    float *Cs[NVQA_MAX_LAYERS] = {};           // [(TS+1)*B][R]
    float *U[NVQA_MAX_LAYERS] = {};            // [TS*B][R] Dropout(h of layer below), l >= 1
    float *dCT = nullptr, *dHT = nullptr;      // [L][B][R] head -> final state gradients
    bool dct_zero = false;                     // arch2: dCT and dHT[0 .. L-2] are known to be zero (nobody wrote them since the last clear)
    bool h0_img_clean = false;                 // bf16: the step-0 slices of the bf16 images of Hs hold what Hs' step-0 slices hold (zeros)
    float *qd = nullptr, *vd = nullptr, *qc = nullptr, *ic = nullptr, *zd = nullptr;
    float *scores = nullptr, *dscores = nullptr, *rowloss = nullptr;
    float *dqc = nullptr, *dic = nullptr;
    float *colpart = nullptr, *slabs = nullptr;
    float *chain_slabs = nullptr; // [L][2][NVQA_BWD_Z][B][R] split-K partials of the BPTT level products
    bool bf16 = false;            // nvqa_set_precision: GEMM operands rounded to bf16, bf16 MFMA, f32 accumulate
    bool fold_i2h = true;         // layer-0 input projection as a first K segment of the level kernel; NVQA_FOLD_I2H=0: separate time-batched GEMM
    bool batch_uniform = false;   // current batch: all lengths equal (known on the host)
    float batch_len_frac = 1.f;   // current batch: filled share of its B x T (row, step) slots (host batches: exact; dataset route: the dataset's mean)
    bool persist_on = false;      // forward LSTM as one persistent weight-stationary launch (lstm_persist.h)
    int num_cus = 0;
    unsigned *pf_cnt = nullptr;   // its arrival counters + err word (zeroed at creation and by the latch kernel behind every launch)
    size_t pf_cnt_words = 0;
    unsigned long long *pf_ts = nullptr; // debug timestamps of the persistent kernels (NVQA_PF_DBG & 32)
    bool img_fwd_valid = false, img_bwd_valid = false; // this step's persistent bf16 kernels wrote act_b16 / dg_b16
    bool wgrad_tr = true;              // bf16 weight gradients on the transposed-read kernel (wgrad_bf16.h)
    float *pb_bias = nullptr;     // [L][RB][4R] LSTM bias-gradient partial sums left by the persistent BPTT kernel
    unsigned *pb_bias_cnt = nullptr; // [L][32] arrivals per (layer, unit tile): the last row block's workgroup writes the bias gradients (NVQA_BIAS_IN_BPTT=0: k_bias_sum)
    bool pb_bias_done = false;    // this step's persistent BPTT launch wrote b_i2h / b_h2h of every layer itself
    int pb_bias_rb = 0;           // row blocks of this step's partial sums (0: none: lstm_wgrads runs the column-sum kernels)
    bool wi2h0_img_step = false;         // ... made by this step's embedding launch (else lstm_dx0 makes it)
    unsigned short *wi2h0_t16 = nullptr; // bf16 image of W_i2h[0]^T ([E][4R]) for the gfx950-form d(input) product, remade every step
    unsigned short *dg_b16 = nullptr;  // bf16 image of dG for the persistent BPTT kernel's bf16 instance (lstm_persist_bwd2.h)
    unsigned short *x0_b16 = nullptr;  // bf16 image of X0 ([TS*B][E]), written by the embedding kernels in bf16 mode (E % 8 == 0)
    bool x0_img_valid = false;         // ... for the current step
    unsigned short *act_b16 = nullptr; // bf16 images of Hs / U for the persistent kernel's bf16 instance (lstm_persist.h)
    unsigned *h_pf_err = nullptr; // pinned copies of the sticky err records (forward: words 0-3, BPTT: words 4-7)
    unsigned pf_spin = 0;         // NVQA_PF_SPIN at nvqa_create: polls before a persistent-kernel wait gives up (0: NVQA_PF_SPIN_LIMIT)
    int pf_spin_steps = -1;       // NVQA_PF_SPIN_STEPS: training steps the NVQA_PF_SPIN limit still applies to (-1: all; a test knob)
    unsigned *pf_sticky = nullptr; // device: first failure of a persistent kernel since the host last looked (persist_fwd.hip: k_err_latch)
    float *dp_status = nullptr, *h_dp_status = nullptr; // data parallel: [0] = ranks whose persistent kernel gave up in this step (summed by the exchange)
    int persist_bwd_on = -1;       // BPTT as one persistent launch (lstm_persist_bwd2.h)
    unsigned *pb_cnt = nullptr;   // its counters + err record
    size_t pb_cnt_words = 0;
    float *pb_pup = nullptr;      // [L-1][TS*B][R] products handed from the UP role to the cells of the layer below
    size_t slab_floats = 0;
    int32_t *argmax = nullptr;
    int32_t *mc = nullptr;   // multiple-choice candidates of the batch being evaluated (nvqa_evaluate), allocated on first use
    float *h_rowloss = nullptr; // pinned [B]: the row losses of the last step / evaluation, written by k_softmax_ce itself
    int loss_rows = 0;          // rows of h_rowloss the loss is the mean of (B after a training step, n after nvqa_evaluate)
    double *norm_part = nullptr; // nvqa_param_norms: per-workgroup sums of squares, allocated on first use

    nvqa::Dataset ds;

    // data parallel
    void *comm = nullptr;
    const void *rccl = nullptr; // the collective library's entry points this communicator came from (nvqa_api.hip)
    int rank = 0, world = 1;
    int dp_slot = 0;            // which of dp_status[0 .. 1] this step uses (they alternate; k_rmsprop clears the next step's)
    bool dp_clean[2] = {false, false};
    bool dp_overlap_bptt = false; // NVQA_DP_OVERLAP_BPTT=1: the multimodal segment is exchanged UNDER the persistent BPTT launch (round 3's order)
    int comm_cus = 0;           // compute units left to the collective while a persistent kernel runs (nvqa_comm_init)

    // profiling
    bool prof_on = false;
    nvqa::ProfEntry prof[nvqa::PF_COUNT];
    std::vector<hipEvent_t> prof_pool; // events waiting for reuse (prof.h)
};
