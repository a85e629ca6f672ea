// nvqa_api.hip -- the C ABI of include/nvqa.h: context, parameter I/O, and the
// orchestration of the training step (forward, backward, update) as launches of the
// gfx950 kernels in gemm_f32.h / kernels.h on one HIP stream.
//
// Reference call stack being replaced: optim.rmsprop -> JdJ
// (002_train_vqa_arch1/002_train_baseline.lua:272-335,408).
#include <dlfcn.h>
#include <limits.h>
#include <unistd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "gemm_f32.h"
#include "kernels.h"
#include "wgrad_bf16.h"
#include "nvqa_ctx.h"
#include "persist_host.h"

using namespace nvqa;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
namespace nvqa {
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace nvqa
extern "C" const char *nvqa_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------
// profiling (scope: prof.h)
// ------------------------------------------------------------------------------------
static const char *kProfNames[PF_COUNT] = {
    "assemble",      "emb_fwd",       "gemm_i2h_fwd", "lstm_step_fwd", "head_prep", "gemm_head_fwd",
    "softmax_ce",    "gemm_head_bwd", "lstm_step_bwd", "gemm_dgrad",   "gemm_wgrad", "reduce_slabs",
    "colsum",        "emb_bwd",       "rmsprop",       "allreduce",    "gather_batch", "lstm_bwd_finish",
    "ride_gemm"};


static void prof_collect(nvqa_ctx *c)
{
    for (int i = 0; i < PF_COUNT; ++i) {
        for (auto &p : c->prof[i].pending) {
            float ms = 0;
            (void)hipEventSynchronize(p.second);
            (void)hipEventElapsedTime(&ms, p.first, p.second);
            c->prof[i].ms += ms;
            c->prof_pool.push_back(p.first);
            c->prof_pool.push_back(p.second);
        }
        c->prof[i].pending.clear();
    }
}

// ------------------------------------------------------------------------------------
// GEMM configurations (tile shapes; see DESIGN.md "Kernels")
// ------------------------------------------------------------------------------------
// BIG : 128x128x32 block, 4 waves of 64x64 (2x2 MFMA 32x32x2 tiles)  -- time-batched products
// MED : 64x64x32 block, 4 waves of 32x32                              -- M = B products of the head
// LSTM forward step : 32 rows x 32 units x 4 gates, MFMA 16x16x4, fused cell
// BPTT level      : 64x64x32 split-K products into slabs + k_lstm_bwd_finish
//                 MF   BM   BN   BK  WM WN WK PF
typedef Cfg<32, 128, 128, 32, 2, 2, 1, 1> CfgBig;
// tools/kbench10 (sweep over the head / d(input) / i2h shapes): 8 waves of 16 x 32 (64x64) or 32 x 64 (128x128) with
// the 16x16x4 MFMA beat the 4-wave 32x32x2 forms by 5-10 % in every layout except the long-K weight gradients
// Round 3: K-tiles of 64 for the 64 x 64 and 64 x 32 configurations (half the barriers and load rounds of these short,
// latency-bound K loops): head backward 0.137 -> 0.131 ms, d(input) 0.132 -> 0.126, arch2 heads 0.110 -> 0.091, bf16 arch2
// heads 0.108 -> 0.083; the one product that LOSES is the two-problem forward projection launch (0.098 -> 0.105): CfgMedMulti
typedef Cfg<16, 64, 64, 64, 4, 2, 1, 1> CfgMed;
typedef Cfg<16, 64, 64, 32, 4, 2, 1, 1> CfgMedMulti;
// gfx950 form of a bf16 product (gemm_f32.h BF = 2: both operands bf16 in memory, K-contiguous): d(layer-0 input)
typedef Cfg<16, 128, 128, 32, 4, 2, 1, 1, 0, 0, 2, 1> CfgDx0B;
typedef Cfg<16, 128, 64, 32, 4, 2, 1, 1, 0, 0, 2, 1> CfgDx0B64;
typedef Cfg<16, 128, 128, 32, 4, 2, 1, 1> CfgBig16;
typedef Cfg<16, 64, 32, 64, 4, 2, 1, 1> CfgNarrow; // 64 x 32 tiles, 8 waves of 16 x 16: products whose 64 x 64 tiling leaves half the
                                                   // CUs idle (W_o, dzd: 128 tiles; 20 vs 27 us) or whose N wastes wide tiles (d(input), N = 200: 122 vs 140 us)
#define NVQA_BWD_Z 4 // K slices of the BPTT level products
#define NVQA_BWD_ZMAX 16 // ... of a ragged batch's levels with few active row tiles (gemm_f32.h zsplit_for)
typedef Cfg<16, 64, 64, 64, 4, 1, 2, 1> CfgLstmFwd; // 8 waves: 2 K-groups x 4 row tiles of 16 rows x 16 units x 4 gates (tools/kbench4: 37.5 vs 44.3 us per level)
typedef Cfg<16, 64, 64, 32, 4, 2, 1, 1> CfgBwdLevel; // 8 waves of 16x32 (tools/kbench2: 34.2 vs 38.9 us per level for the 32x32x2 form)

// LONGK: the time-batched weight gradients (K = T*B): the 4-wave 32x32x2 form is the faster one there
template <int AM, int BMo, class Epi, bool LONGK = false>
static int gemm_big(nvqa_ctx *c, const GemmArgs &g, const Epi &e, hipStream_t st = nullptr)
{
    typedef typename std::conditional<LONGK, CfgBig, CfgBig16>::type C;
    if (c->bf16) NVQA_HIP((launch_gemm<typename WithBF<C>::type, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
    else NVQA_HIP((launch_gemm<C, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
    return 0;
}
template <int AM, int BMo, class Epi>
static int gemm_med(nvqa_ctx *c, const GemmArgs &g, const Epi &e, hipStream_t st = nullptr)
{
    if (((g.M + 63) / 64) * ((g.N + 63) / 64) <= 160) { // fewer tiles than CUs: halve the tile
        if (c->bf16) NVQA_HIP((launch_gemm<WithBF<CfgNarrow>::type, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
        else NVQA_HIP((launch_gemm<CfgNarrow, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
        return 0;
    }
    if (c->bf16) NVQA_HIP((launch_gemm<WithBF<CfgMed>::type, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
    else NVQA_HIP((launch_gemm<CfgMed, AM, BMo, false, Epi>(st ? st : c->s, g, e)));
    return 0;
}

// NVQA_XCD=0 keeps the natural tile order (for A/B measurements of the XCD-aware order)
static int xcd_order()
{
    static const int v = [] { const char *e = getenv("NVQA_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}

static GemmArgs mkargs(const float *A, int lda, const float *B, int ldb, int M, int N, int K,
                       int kslice = 0, int R = 0, const int *mlimit = nullptr)
{
    GemmArgs g = {};
    g.xcd = xcd_order();
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.kslice = kslice > 0 ? kslice : (K > 0 ? K : 1);
    g.R = R; g.mlimit = mlimit;
    return g;
}

static inline Drop mkdrop(const nvqa_dropout *dr, bool train)
{
    Drop d;
    if (!train || !dr || dr->mode == 0) {
        d.mode = 0; d.p = 0.f; d.inv_keep = 1.f; d.seed = 0; d.step = 0;
    } else {
        d.mode = 1; d.p = dr->p; d.inv_keep = 1.0f / (1.0f - dr->p); d.seed = dr->seed; d.step = dr->step;
    }
    return d;
}

static void comm_destroy(nvqa_ctx *c); // with the RCCL loader, below

// ------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------
static int g_alloc_count = 0, g_alloc_fail = -1;
template <class T> static int dalloc(T **p, size_t n)
{
    if (g_alloc_fail >= 0 && g_alloc_count++ == g_alloc_fail) {
        set_error("hipMalloc of %zu bytes failed: injected by NVQA_FAIL_ALLOC", std::max<size_t>(n, 1) * sizeof(T));
        return -1;
    }
    NVQA_HIP(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

static int check_dims(const nvqa_dims *d)
{
    if (!d) { set_error("dims is NULL"); return -1; }
    if (d->arch != NVQA_ARCH1 && d->arch != NVQA_ARCH2) { set_error("arch must be 1 or 2"); return -1; }
    if (d->B < 1 || d->T < 1 || d->V < 1 || d->L < 1 || d->L > NVQA_MAX_LAYERS) {
        set_error("bad dims (B=%d T=%d V=%d L=%d)", d->B, d->T, d->V, d->L);
        return -1;
    }
    // 16-byte vector accesses: every row width must be a multiple of 4 floats
    if (d->E % 4 || d->R % 4 || d->I % 4 || d->A % 4 || (d->arch == NVQA_ARCH1 && d->C % 4)) {
        set_error("E, R, I, C, A must be multiples of 4 (got E=%d R=%d I=%d C=%d A=%d)", d->E, d->R,
                  d->I, d->C, d->A);
        return -1;
    }
    // single-workgroup batch assembly kernels keep per-step / per-row tables in LDS (kernels.h)
    if (d->arch == NVQA_ARCH2 && d->T > NVQA_ARCH2_TMAX) {
        set_error("arch2: T=%d exceeds the %d question steps k_arch2_tmax supports", d->T, NVQA_ARCH2_TMAX);
        return -1;
    }
    if (d->arch == NVQA_ARCH1 && (3 + 16) * ((size_t)d->T + 1) * sizeof(int) > 64 * 1024) {
        set_error("arch1: T=%d needs %zu bytes of LDS in k_sort_lengths (limit 65536)", d->T, (3 + 16) * ((size_t)d->T + 1) * sizeof(int));
        return -1;
    }
    return 0;
}

static int create_impl(nvqa_ctx *c);

extern "C" int nvqa_create(const nvqa_dims *dims, int device, nvqa_ctx **out)
{
    if (!out) { set_error("out is NULL"); return -1; }
    *out = nullptr;
    NVQA_TRY(check_dims(dims));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available (libnvqa has no CPU fallback)");
        return -2;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (0..%d)", device, ndev - 1); return -1; }
    NVQA_HIP(hipSetDevice(device));
    nvqa_ctx *c = new nvqa_ctx();
    c->d = *dims;
    c->device = device;
    if (nvqa_layout_init(dims, &c->lo)) { delete c; set_error("layout"); return -1; }
    const int rc = create_impl(c);
    if (rc != 0) { // a failure half-way (out of memory, ...) releases whatever was acquired; the message survives
        const std::string keep = nvqa_last_error();
        nvqa_destroy(c);
        set_error("%s", keep.c_str());
        return rc;
    }
    *out = c;
    return 0;
}

// NVQA_FAIL_ALLOC=n makes the n-th device allocation of nvqa_create fail (tests: the failure path frees the context)
static int create_impl(nvqa_ctx *c)
{
    {
        const char *e = getenv("NVQA_FAIL_ALLOC");
        g_alloc_fail = e ? atoi(e) : -1;
        g_alloc_count = 0;
    }
    const nvqa_dims &d = c->d;
    c->TS = d.arch == NVQA_ARCH1 ? d.T : d.T + 2;
    const size_t B = d.B, R = d.R, E = d.E, L = d.L, TS = c->TS, TB = TS * B;
    {   // one compute stream (the whole step is a single dependency chain of chip-filling launches) and
        // one communication stream for the data-parallel all-reduce buckets
        int least = 0, greatest = 0;
        NVQA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        NVQA_HIP(hipStreamCreateWithPriority(&c->s, hipStreamNonBlocking, greatest));
        NVQA_HIP(hipStreamCreateWithPriority(&c->sc, hipStreamNonBlocking, greatest));
        NVQA_HIP(hipStreamCreateWithFlags(&c->sx, hipStreamNonBlocking));
        NVQA_HIP(hipEventCreateWithFlags(&c->evTok, hipEventDisableTiming));
        NVQA_HIP(hipEventCreateWithFlags(&c->evIdx, hipEventDisableTiming));
        for (int i = 0; i < 3 + NVQA_MAX_LAYERS; ++i) NVQA_HIP(hipEventCreateWithFlags(&c->evSeg[i], hipEventDisableTiming));
        NVQA_HIP(hipEventCreateWithFlags(&c->evComm, hipEventDisableTiming));
        NVQA_HIP(hipEventCreateWithFlags(&c->evStart, hipEventDisableTiming));
    }
    // (sized for the widest layout of these dimensions: nvqa_set_fusion(ctx, 2) = netdef.A_B makes W_o [A x 2C])
    size_t total_max = c->lo.total;
    if (d.arch == NVQA_ARCH1) { nvqa_layout lj; if (!nvqa_layout_init_fusion(&c->d, 2, &lj)) total_max = std::max(total_max, lj.total); }
    NVQA_TRY(dalloc(&c->P, total_max));
    NVQA_TRY(dalloc(&c->G, total_max));
    NVQA_TRY(dalloc(&c->M2, total_max));
    NVQA_HIP(hipMemsetAsync(c->P, 0, total_max * 4, c->s));
    NVQA_HIP(hipMemsetAsync(c->G, 0, total_max * 4, c->s));
    NVQA_HIP(hipMemsetAsync(c->M2, 0, total_max * 4, c->s));
    NVQA_TRY(dalloc(&c->tok, B * d.T));
    NVQA_TRY(dalloc(&c->len, B));
    NVQA_TRY(dalloc(&c->lab, B));
    NVQA_TRY(dalloc(&c->img, B * d.I));
    NVQA_TRY(dalloc(&c->qinds, B));
    NVQA_TRY(dalloc(&c->sort_idx, B));
    NVQA_TRY(dalloc(&c->sort_inv, B));
    NVQA_TRY(dalloc(&c->nrows, TS));
    NVQA_TRY(dalloc(&c->ptok, TB));
    NVQA_TRY(dalloc(&c->tinfo, 2));
    { // token-segment index for the embedding gradient (kernels.h: k_tok_index)
        const char *e = getenv("NVQA_EMB_SEG");
        c->tok_seg = !(e && e[0] == '0');
        e = getenv("NVQA_WGRAD_TR"); // 0: bf16 weight gradients through gemm_f32.h's BF mode (A/B runs)
        c->wgrad_tr = !(e && e[0] == '0');
        const size_t VT = d.V + 1, slots = TB / NVQA_ES_SHORT + 2;
        NVQA_TRY(dalloc(&c->seg_start, VT + 1));
        NVQA_HIP(hipMemsetAsync(c->seg_start, 0, (VT + 1) * 4, c->s));
        NVQA_TRY(dalloc(&c->pslot, slots + 4));
        NVQA_TRY(dalloc(&c->perm, TB));
        NVQA_TRY(dalloc(&c->seg_done, slots));
        c->seg_part_bytes = slots * NVQA_ES_CHUNKS * E * sizeof(float);
        NVQA_TRY(dalloc(&c->seg_part, slots * NVQA_ES_CHUNKS * E));
    }
    NVQA_TRY(dalloc(&c->X0, TB * E));
    NVQA_TRY(dalloc(&c->dX0, TB * E));
    for (size_t l = 0; l < L; ++l) {
        NVQA_TRY(dalloc(&c->Gt[l], TB * 4 * R));
        if (l == 0) { // one allocation for all layers: [L][(TS+1)*B][R]
            NVQA_TRY(dalloc(&c->Hs[0], L * (TS + 1) * B * R));
            NVQA_TRY(dalloc(&c->Cs[0], L * (TS + 1) * B * R));
            NVQA_HIP(hipMemsetAsync(c->Hs[0], 0, L * (TS + 1) * B * R * 4, c->s)); // step-0 state stays zero
            NVQA_HIP(hipMemsetAsync(c->Cs[0], 0, L * (TS + 1) * B * R * 4, c->s));
        } else {
            c->Hs[l] = c->Hs[0] + l * (TS + 1) * B * R;
            c->Cs[l] = c->Cs[0] + l * (TS + 1) * B * R;
        }
        if (l > 0) NVQA_TRY(dalloc(&c->U[l], TB * R));
    }
    NVQA_TRY(dalloc(&c->dCT, 2 * L * B * R)); // dCT, then dHT: one allocation, so that arch2's per-step clear is one fill
    c->dHT = c->dCT + L * B * R;
    NVQA_HIP(hipMemsetAsync(c->dCT, 0, 2 * L * B * R * 4, c->s)); // NVQA_QUIRK_H0 reads dHT before the first backward
    c->dct_zero = true;
    const size_t Q = d.arch == NVQA_ARCH1 ? 2 * R * L : R, C = d.arch == NVQA_ARCH1 ? d.C : 0;
    NVQA_TRY(dalloc(&c->qd, B * Q));
    NVQA_TRY(dalloc(&c->vd, B * d.I));
    NVQA_TRY(dalloc(&c->qc, B * C));
    NVQA_TRY(dalloc(&c->ic, B * C));
    NVQA_TRY(dalloc(&c->zd, B * 2 * C)); // 2C: netdef.A_B joins qc and ic
    NVQA_TRY(dalloc(&c->dqc, B * std::max(C, R)));
    NVQA_TRY(dalloc(&c->dic, B * C));
    NVQA_TRY(dalloc(&c->scores, B * d.A));
    NVQA_TRY(dalloc(&c->dscores, B * d.A));
    NVQA_TRY(dalloc(&c->rowloss, B));
    NVQA_TRY(dalloc(&c->argmax, B));
    // scratch: column-sum partials (64 splits x widest matrix) and split-K slabs
    const size_t widest = std::max<size_t>(std::max<size_t>(4 * R, d.I), std::max<size_t>(d.A, std::max<size_t>(E, C)));
    NVQA_TRY(dalloc(&c->colpart, 64 * widest));
    c->slab_floats = 8 * 4 * R * std::max<size_t>(std::max(R, E), 128);
    NVQA_TRY(dalloc(&c->slabs, c->slab_floats));
    NVQA_TRY(dalloc(&c->chain_slabs, (size_t)L * 2 * NVQA_BWD_ZMAX * ((B + 63) / 64 * 64) * ((R + 63) / 64 * 64)));
    {   // NVQA_FOLD_I2H=0: layer 0's input projection as a separate time-batched GEMM (A/B runs of the per-level path)
        const char *ef = getenv("NVQA_FOLD_I2H");
        c->fold_i2h = !(ef && ef[0] == '0');
    }
    {   // persistent forward LSTM (lstm_persist.h) where the shape is eligible; NVQA_PERSIST=0: one launch per wavefront level
        const char *ep = getenv("NVQA_PERSIST");
        c->persist_on = !(ep && ep[0] == '0');
        { const char *e = getenv("NVQA_RIDE_GEMM"); c->ride_gemm_on = !(e && e[0] == '0'); }   // A/B switches of the ride-along jobs,
        { const char *e = getenv("NVQA_TOK_IN_BPTT"); c->tok_in_bptt_on = !(e && e[0] == '0'); } // read here like the other switches
        { const char *e = getenv("NVQA_RIDE_FWD"); c->ride_fwd_on = !(e && e[0] == '0'); }
        hipDeviceProp_t prop;
        NVQA_HIP(hipGetDeviceProperties(&prop, c->device));
        c->num_cus = prop.multiProcessorCount;
        c->pf_cnt_words = ((size_t)L * ((B + 63) / 64) * TS + 4 + 3) / 4 * 4; // counters for the finest row blocking + err word, 16-byte multiple
        NVQA_TRY(dalloc(&c->pf_cnt, c->pf_cnt_words));
        NVQA_HIP(hipMemsetAsync(c->pf_cnt, 0, c->pf_cnt_words * 4, c->s)); // once: every launch's latch kernel leaves the block zeroed
        NVQA_TRY(dalloc(&c->pf_ts, 2048)); // NVQA_PF_DBG / NVQA_PB_DBG & 32: timestamps of 256 + 256 workgroups
        NVQA_HIP(hipMemsetAsync(c->pf_ts, 0, 2048 * 8, c->s));
        NVQA_HIP(hipHostMalloc((void **)&c->h_pf_err, 8 * sizeof(unsigned), hipHostMallocDefault));
        memset(c->h_pf_err, 0, 32);
        // persistent BPTT (lstm_persist_bwd2.h): default where the shape is eligible; NVQA_PERSIST_BWD=0: the
        // per-level kernels
        const char *eb = getenv("NVQA_PERSIST_BWD");
        c->persist_bwd_on = !eb ? -1 : (eb[0] == '1' ? 1 : 0);
        if (d.R == 512 && L <= 2) {
            const size_t rbmax = (B + 63) / 64;
            c->pb_cnt_words = persist_bwd_counter_words(d, (int)TS);
            NVQA_TRY(dalloc(&c->pb_cnt, c->pb_cnt_words));
            NVQA_HIP(hipMemsetAsync(c->pb_cnt, 0, c->pb_cnt_words * 4, c->s));
            NVQA_TRY(dalloc(&c->pb_bias, L * rbmax * 4 * R));
            const char *ebb = getenv("NVQA_BIAS_IN_BPTT"); // 0: the row blocks' sums are added by k_bias_sum launches (A/B runs)
            if (!(ebb && ebb[0] == '0')) {
                NVQA_TRY(dalloc(&c->pb_bias_cnt, L * 32));
                NVQA_HIP(hipMemsetAsync(c->pb_bias_cnt, 0, L * 32 * 4, c->s));
            }
            if (L > 1) NVQA_TRY(dalloc(&c->pb_pup, (L - 1) * TS * B * R));
        }
        { const char *es = getenv("NVQA_PF_SPIN"); c->pf_spin = es ? (unsigned)strtoul(es, nullptr, 0) : 0u; } // 0: the kernels' default
        { const char *es = getenv("NVQA_PF_SPIN_STEPS"); c->pf_spin_steps = es ? atoi(es) : -1; } // tests: the limit above holds for the first n training steps only
        NVQA_TRY(dalloc(&c->pf_sticky, 8));
        NVQA_HIP(hipMemsetAsync(c->pf_sticky, 0, 32, c->s));
        NVQA_TRY(dalloc(&c->dp_status, 4));
        NVQA_HIP(hipMemsetAsync(c->dp_status, 0, 16, c->s));
        NVQA_HIP(hipHostMalloc((void **)&c->h_dp_status, 4 * sizeof(float), hipHostMallocDefault));
        memset(c->h_dp_status, 0, 16);
    }
    NVQA_HIP(hipHostMalloc((void **)&c->h_rowloss, B * sizeof(float), hipHostMallocDefault)); // k_softmax_ce writes the row losses here
    memset(c->h_rowloss, 0, B * sizeof(float));
    c->loss_rows = (int)B;
    NVQA_HIP(hipStreamSynchronize(c->s));
    {   // The persistent LSTM kernels exist for the shapes the reference trains (rnn_size 512; input_encoding_size 200 or 512;
        // at most 2 layers for the BPTT): any other -rnn_size runs the per-level kernels (0.48-0.53 of the f32 MFMA peak
        // instead of 0.60-0.65).  Say so once per process instead of silently being slower (nvqa_persistent_state tells
        // a host program the same).
        int RB = 0;
        const bool fwd = persist_rows(c) > 0, bwd = persist_bwd_rows(c, &RB) > 0;
        static bool said = false;
        if ((!fwd || !bwd) && !said && d.R >= 256 && !(getenv("NVQA_QUIET") && atoi(getenv("NVQA_QUIET")))) {
            fprintf(stderr, "[nvqa] note: R=%d E=%d L=%d B=%d is outside the persistent LSTM kernels' shapes (R = 512, E in {200, 512}%s): "
                            "forward %s, BPTT %s\n", d.R, d.E, d.L, d.B, d.L > 2 ? ", L <= 2 for the BPTT" : "",
                    fwd ? "persistent" : "per-level kernels", bwd ? "persistent" : "per-level kernels");
            said = true;
        }
    }
    return 0;
}

extern "C" int nvqa_destroy(nvqa_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    prof_collect(c);
    for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
    c->prof_pool.clear();
    comm_destroy(c);
    if (c->hset[0].tok) { // two device sets + pinned staging of the host-batch entries
        for (int p = 0; p < 2; ++p) {
            for (void *q : {(void *)c->bset[p].tok, (void *)c->bset[p].len, (void *)c->bset[p].lab, (void *)c->bset[p].img})
                if (q) (void)hipFree(q);
            for (void *q : {(void *)c->hset[p].tok, (void *)c->hset[p].len, (void *)c->hset[p].lab, (void *)c->hset[p].img})
                if (q) (void)hipHostFree(q);
            if (c->evCopied[p]) (void)hipEventDestroy(c->evCopied[p]);
            if (c->evBatchFree[p]) (void)hipEventDestroy(c->evBatchFree[p]);
        }
    } else {
        for (void *q : {(void *)c->tok, (void *)c->len, (void *)c->lab, (void *)c->img})
            if (q) (void)hipFree(q);
    }
    void *ptrs[] = {c->P, c->G, c->M2, c->qinds, c->sort_idx, c->sort_inv,
                    c->nrows, c->ptok, c->tinfo, c->X0, c->dX0, c->dCT, c->qd, c->vd, c->qc, c->ic, c->zd, c->dqc,
                    c->dic, c->scores, c->dscores, c->rowloss, c->argmax, c->colpart, c->slabs, c->chain_slabs, c->mc,
                    c->ds.Q, c->ds.QL, c->ds.IP, c->ds.ANS, c->ds.F, c->seg_start, c->pslot, c->perm, c->seg_done, c->seg_part};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (int l = 0; l < NVQA_MAX_LAYERS; ++l) {
        if (c->Gt[l]) (void)hipFree(c->Gt[l]);
        if (l == 0 && c->Hs[l]) (void)hipFree(c->Hs[l]);
        if (l == 0 && c->Cs[l]) (void)hipFree(c->Cs[l]);
        if (c->U[l]) (void)hipFree(c->U[l]);
    }
    if (c->h_rowloss) (void)hipHostFree(c->h_rowloss);
    if (c->h_pf_err) (void)hipHostFree(c->h_pf_err);
    if (c->pf_cnt) (void)hipFree(c->pf_cnt);
    if (c->pf_ts) (void)hipFree(c->pf_ts);
    if (c->act_b16) (void)hipFree(c->act_b16);
    if (c->dg_b16) (void)hipFree(c->dg_b16);
    if (c->wi2h0_t16) (void)hipFree(c->wi2h0_t16);
    if (c->ride_dev) (void)hipFree(c->ride_dev);
    if (c->pb_cnt) (void)hipFree(c->pb_cnt);
    if (c->pb_bias) (void)hipFree(c->pb_bias);
    if (c->pb_bias_cnt) (void)hipFree(c->pb_bias_cnt);
    if (c->x0_b16) (void)hipFree(c->x0_b16);
    if (c->pb_pup) (void)hipFree(c->pb_pup);
    if (c->pf_sticky) (void)hipFree(c->pf_sticky);
    if (c->dp_status) (void)hipFree(c->dp_status);
    if (c->norm_part) (void)hipFree(c->norm_part);
    if (c->h_dp_status) (void)hipHostFree(c->h_dp_status);
    for (hipEvent_t e : {c->evComm, c->evStart})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->evSeg)
        if (e) (void)hipEventDestroy(e);
    if (c->sc) (void)hipStreamDestroy(c->sc);
    if (c->sx) (void)hipStreamDestroy(c->sx);
    if (c->evTok) (void)hipEventDestroy(c->evTok);
    if (c->evIdx) (void)hipEventDestroy(c->evIdx);
    if (c->s) (void)hipStreamDestroy(c->s);
    delete c;
    return 0;
}

// after the stream has drained: did a persistent-kernel wait give up? (lstm_persist.h: every spin is bounded)
// mean of the first loss_rows row losses (pinned host array written by k_softmax_ce) in the FIXED order of round 3's device
// kernel k_loss_mean -- 256 strided partial sums, then a binary tree -- so the value is bit-identical to what that kernel
// produced and does not depend on the host's vector width (f32 adds only, no contraction possible).  Call after the
// stream has been synchronised.
static float loss_mean_host(const nvqa_ctx *c)
{
    volatile const float *r = c->h_rowloss;
    const int n = c->loss_rows;
    float part[256];
    for (int i = 0; i < 256; ++i) {
        float s = 0.f;
        for (int b = i; b < n; b += 256) s += r[b];
        part[i] = s;
    }
    for (int o = 128; o > 0; o >>= 1)
        for (int i = 0; i < o; ++i) part[i] += part[i + o];
    return part[0] / (float)n;
}

static int check_persist(nvqa_ctx *c)
{
    static const int ts_dbg = [] { const char *a = getenv("NVQA_PF_DBG"), *b = getenv("NVQA_PB_DBG"); return ((a ? atoi(a) : 0) | (b ? atoi(b) : 0)) & 32; }();
    static const int seg_dbg = [] { const char *a = getenv("NVQA_PF_DBG"); return (a ? atoi(a) : 0) & 128; }();
    if (seg_dbg && c->pf_ts) { // measurement only (lstm_persist_fwd3.h): shader cycles per chain-step in four segments, per layer, once
        static int left = 3;
        if (left > 0 && --left == 0) {
            std::vector<unsigned long long> h(1024);
            const int RB = (c->d.B + 127) / 128, groups = c->d.L * RB, n = groups * (c->d.R / 16);
            if (n <= 256 && hipMemcpy(h.data(), c->pf_ts, 1024 * 8, hipMemcpyDeviceToHost) == hipSuccess)
                for (int l = 0; l < c->d.L; ++l) {
                    double s4[4] = {0, 0, 0, 0}; int m = 0;
                    for (int b = 0; b < n; ++b) if ((b % groups) / RB == l) { ++m; for (int i = 0; i < 4; ++i) s4[i] += (double)h[b * 4 + i]; }
                    const double den = (double)m * 2 * c->TS;
                    fprintf(stderr, "[nvqa] forward layer %d, cycles per chain-step (wave 0): stream %.0f, last spill %.0f, barrier %.0f, cell %.0f\n", l, s4[0] / den, s4[1] / den, s4[2] / den, s4[3] / den);
                }
        }
    }
    if (ts_dbg && c->pf_ts) { // measurement only: phase times of the persistent kernels' workgroups (min / max over workgroups), once
        static int left = 3;
        if (left > 0 && --left == 0) {
            std::vector<unsigned long long> h(2048);
            if (hipMemcpy(h.data(), c->pf_ts, 2048 * 8, hipMemcpyDeviceToHost) == hipSuccess)
                for (int k = 0; k < 2; ++k) {
                    unsigned long long t0 = ~0ull, w_lo = ~0ull, w_hi = 0, e_lo = ~0ull, e_hi = 0; int n = 0;
                    for (int b = 0; b < 256; ++b) { const unsigned long long *t = &h[k * 1024 + b * 4]; if (t[2]) { t0 = std::min(t0, t[0]); ++n; } }
                    for (int b = 0; b < 256; ++b) {
                        const unsigned long long *t = &h[k * 1024 + b * 4];
                        if (!t[2]) continue;
                        w_lo = std::min(w_lo, t[1] - t0); w_hi = std::max(w_hi, t[1] - t0); e_lo = std::min(e_lo, t[2] - t0); e_hi = std::max(e_hi, t[2] - t0);
                    }
                    if (n) fprintf(stderr, "[nvqa] persistent %s: %d workgroups; weights resident after %.1f .. %.1f us, done after %.1f .. %.1f us\n",
                                   k ? "BPTT" : "forward", n, w_lo * 0.01, w_hi * 0.01, e_lo * 0.01, e_hi * 0.01);
                    for (int l = 1; l <= 4 && !k; ++l) { // (lstm_persist_fwd3.h tags its workgroups: slot 3 = step loop done << 4 | layer + 1)
                        unsigned long long lo = ~0ull, hi = 0, dlo = ~0ull, dhi = 0; int m = 0;
                        for (int b = 0; b < 256; ++b) {
                            const unsigned long long *t = &h[b * 4];
                            if (!t[2] || (int)(t[3] & 15) != l) continue;
                            ++m; lo = std::min(lo, (t[3] >> 4) - t0); hi = std::max(hi, (t[3] >> 4) - t0); dlo = std::min(dlo, t[2] - t0); dhi = std::max(dhi, t[2] - t0);
                        }
                        if (m) fprintf(stderr, "[nvqa]   layer %d: %d workgroups; step loop done after %.1f .. %.1f us, workgroup done after %.1f .. %.1f us\n", l - 1, m, lo * 0.01, hi * 0.01, dlo * 0.01, dhi * 0.01);
                    }
                }
        }
    }
    if (c->h_pf_err && (c->h_pf_err[0] || c->h_pf_err[4])) {
        // one report for both kernels: a forward time-out leaves garbage that the BPTT kernel may also trip over
        char fwd[256] = "", bwd[200] = "";
        if (c->h_pf_err[0]) {
            // word index relative to the err word: counters end 4 words before it
            const long widx = (long)(int)c->h_pf_err[1] + (long)c->pf_cnt_words - 4;
            snprintf(fwd, sizeof(fwd), "persistent LSTM kernel: workgroup %u timed out waiting (code 0x%x, counter word %ld = (layer,rowblock) %ld step %ld, value seen %u)",
                     c->h_pf_err[3], c->h_pf_err[0], widx, widx / c->TS, widx % c->TS, c->h_pf_err[2]);
            c->persist_on = false; // later steps take the per-level path
        }
        if (c->h_pf_err[4]) {
            snprintf(bwd, sizeof(bwd), "persistent BPTT kernel: workgroup %u timed out waiting (code 0x%x, value seen %u)", c->h_pf_err[7], c->h_pf_err[4], c->h_pf_err[6]);
            c->persist_bwd_on = 0;
        }
        set_error("%s%s%s; results of that step are invalid and were not applied", fwd, fwd[0] && bwd[0] ? "; " : "", bwd);
        memset(c->h_pf_err, 0, 32);
        (void)hipMemsetAsync(c->pf_sticky, 0, 32, c->s); // reported: k_rmsprop may apply gradients again
        if (c->comm) { // the same switches and the same cleared status words as the ranks that only see the exchanged count (below)
            c->persist_on = false; c->persist_bwd_on = 0;
            if (c->h_dp_status) c->h_dp_status[0] = c->h_dp_status[1] = 0.f;
        }
        return -3;
    }
    if (c->comm && c->h_dp_status && (c->h_dp_status[0] != 0.f || c->h_dp_status[1] != 0.f)) {
        // [0]: the exchanged count of the last step; [1]: of the last FAILED step (kept by k_rmsprop until cleared here)
        set_error("data parallel: the persistent LSTM kernel of %d rank(s) timed out in an earlier step; no rank applied that step's gradients",
                  (int)(c->h_dp_status[0] != 0.f ? c->h_dp_status[0] : c->h_dp_status[1]));
        c->h_dp_status[0] = c->h_dp_status[1] = 0.f;
        c->persist_on = false; c->persist_bwd_on = 0; // every rank sees the same sum: all of them leave the persistent path together
        return -3;
    }
    return 0;
}

extern "C" int nvqa_sync(nvqa_ctx *c)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_HIP(hipStreamSynchronize(c->s));
    return check_persist(c);
}

extern "C" int nvqa_persistent_state(const nvqa_ctx *c, int out[2])
{
    if (!c || !out) { set_error("NULL argument"); return -1; }
    int RB = 0;
    out[0] = persist_rows(c) > 0 ? 1 : 0;
    out[1] = persist_bwd_rows(c, &RB) > 0 ? 1 : 0;
    return 0;
}

// ------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------
extern "C" size_t nvqa_param_count(const nvqa_ctx *c) { return c ? c->lo.total : 0; }

extern "C" int nvqa_segments(const nvqa_ctx *c, size_t out[3])
{
    if (!c || !out) { set_error("NULL argument"); return -1; }
    out[0] = c->lo.seg[0]; out[1] = c->lo.seg[1]; out[2] = c->lo.seg[2];
    return 0;
}

// ABI layout <-> internal layout: only the arch1 embedding weight differs
// (Torch nn.Linear [E][V] at the ABI, gather-friendly [V][E] on the device).
static void to_internal(const nvqa_ctx *c, const float *abi, std::vector<float> &in)
{
    in.assign(abi, abi + c->lo.total);
    if (c->d.arch == NVQA_ARCH1) {
        const size_t E = c->d.E, V = c->d.V;
        const float *src = abi + c->lo.w_e;
        float *dst = in.data() + c->lo.w_e;
        for (size_t e = 0; e < E; ++e)
            for (size_t v = 0; v < V; ++v) dst[v * E + e] = src[e * V + v];
    }
}
static void to_abi(const nvqa_ctx *c, const std::vector<float> &in, float *abi)
{
    memcpy(abi, in.data(), c->lo.total * sizeof(float));
    if (c->d.arch == NVQA_ARCH1) {
        const size_t E = c->d.E, V = c->d.V;
        const float *src = in.data() + c->lo.w_e;
        float *dst = abi + c->lo.w_e;
        for (size_t v = 0; v < V; ++v)
            for (size_t e = 0; e < E; ++e) dst[e * V + v] = src[v * E + e];
    }
}

// torch.norm of the three parameter segments (003_train_vqa_arch2/002_train_baseline.lua:400-407 logs them every 100
// iterations): sums of squares in double on the device (TH accumulates a FloatTensor's norm in double), 6 KB back.
extern "C" int nvqa_param_norms(nvqa_ctx *c, float out[3])
{
    if (!c || !out) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    const int NB = 256;
    if (!c->norm_part) NVQA_HIP(hipMalloc((void **)&c->norm_part, 3 * NB * sizeof(double)));
    size_t off = 0;
    for (int sgm = 0; sgm < 3; ++sgm) {
        hipLaunchKernelGGL(k_sumsq, dim3(NB), dim3(256), 0, c->s, c->P + off, c->lo.seg[sgm], c->norm_part + sgm * NB);
        off += c->lo.seg[sgm];
    }
    NVQA_HIP(hipGetLastError());
    std::vector<double> h(3 * NB);
    NVQA_HIP(hipMemcpyAsync(h.data(), c->norm_part, 3 * NB * sizeof(double), hipMemcpyDeviceToHost, c->s));
    NVQA_HIP(hipStreamSynchronize(c->s));
    for (int sgm = 0; sgm < 3; ++sgm) {
        double t = 0;
        for (int b = 0; b < NB; ++b) t += h[sgm * NB + b];
        out[sgm] = (float)sqrt(t);
    }
    return check_persist(c);
}

extern "C" int nvqa_set_params(nvqa_ctx *c, const float *params)
{
    if (c) c->pristine = false;
    if (!c || !params) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    std::vector<float> in;
    to_internal(c, params, in);
    NVQA_HIP(hipStreamSynchronize(c->s));
    NVQA_HIP(hipMemcpy(c->P, in.data(), c->lo.total * 4, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int nvqa_get_params(nvqa_ctx *c, float *out)
{
    if (!c || !out) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    std::vector<float> in(c->lo.total);
    NVQA_HIP(hipStreamSynchronize(c->s));
    NVQA_HIP(hipMemcpy(in.data(), c->P, c->lo.total * 4, hipMemcpyDeviceToHost));
    to_abi(c, in, out);
    return 0;
}

extern "C" int nvqa_init_params(nvqa_ctx *c, uint64_t seed, float lo, float hi)
{
    if (c) c->pristine = false;
    if (!c) { set_error("ctx is NULL"); return -1; }
    // *_w:uniform(-0.08, 0.08) over each flat segment (002_train_baseline.lua:174-181);
    // counter-based so that every rank of a data-parallel job draws the same values.
    std::vector<float> p(c->lo.total);
    for (size_t i = 0; i < p.size(); ++i) {
        const float u = (float)(nvqa_hash32(seed, 0, 99u, i) >> 8) * (1.0f / 16777216.0f);
        p[i] = lo + (hi - lo) * u;
    }
    NVQA_TRY(nvqa_set_params(c, p.data()));
    NVQA_HIP(hipMemset(c->M2, 0, c->lo.total * 4));
    return 0;
}

static int reduce_segment(nvqa_ctx *c, int seg); // below: RCCL sum of one parameter segment, overlapped
static int reduce_join(nvqa_ctx *c);

extern "C" int nvqa_get_grads(nvqa_ctx *c, float *out, float clamp)
{
    if (!c || !out) { set_error("NULL argument"); return -1; }
    if (!c->have_grads) { set_error("nvqa_get_grads before any nvqa_step"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    std::vector<float> in(c->lo.total);
    NVQA_HIP(hipStreamSynchronize(c->s));
    NVQA_HIP(hipMemcpy(in.data(), c->G, c->lo.total * 4, hipMemcpyDeviceToHost));
    // with a communicator the device buffer holds the sum over ranks: return the global-batch mean
    {
        size_t off = 0;
        for (int sgm = 0; sgm < 3; ++sgm) {
            const float sc = c->gscale[sgm] * ((c->comm && c->world > 1) ? 1.0f / (float)c->world : 1.0f);
            if (sc != 1.0f)
                for (size_t i = 0; i < c->lo.seg[sgm]; ++i) in[off + i] *= sc;
            off += c->lo.seg[sgm];
        }
    }
    if (clamp > 0.f)
        for (float &v : in) v = std::min(std::max(v, -clamp), clamp);
    to_abi(c, in, out);
    return 0;
}

// ------------------------------------------------------------------------------------
// helpers: bias gradient = column sum, split-K wgrad
// ------------------------------------------------------------------------------------
static int colsum(nvqa_ctx *c, const float *X, int M, int N, int ld, float *out, float *out2,
                  hipStream_t st = nullptr)
{
    if (!st) st = c->s;
    ProfScope ps(c, PF_COLSUM, 0, (double)M * N * 4, st);
    int S = std::min(64, std::max(1, M / 64));
    const int rps = (M + S - 1) / S;
    S = (M + rps - 1) / rps;
    hipLaunchKernelGGL(k_colsum_part, dim3((N + 63) / 64, S), dim3(256), 0, st, X, M, N, ld, rps, c->colpart);
    hipLaunchKernelGGL(k_colsum_final, dim3((N + 255) / 256), dim3(256), 0, st, c->colpart, S, N, out, out2);
    NVQA_HIP(hipGetLastError());
    return 0;
}

// bf16 operand mode: the gfx950 k-major kernel (wgrad_bf16.h); A16 / B16: bf16 images of the operands where this step's
// persistent kernels wrote them (NULL: rounded from the f32 rows on the way into LDS -- the same values either way)
static int wgrad_bf16(nvqa_ctx *c, const float *A, const unsigned short *A16, int lda, const float *Bm, const unsigned short *B16, int ldb,
                      int M, int N, int K, float *dW, float *slabs, hipStream_t st)
{
    const int tiles = ((M + NVQA_WB_BM - 1) / NVQA_WB_BM) * ((N + NVQA_WB_BN - 1) / NVQA_WB_BN);
    int ks = 1;
    while (ks < 16 && tiles * ks < 512 && K / (ks * 2) >= 256) ks *= 2; // measured (round 2): 2048 x 512: 8 slices, 2048 x 200: 16
    { static const int ks_env = [] { const char *e = getenv("NVQA_WB_KS"); return e ? atoi(e) : 0; }(); if (ks_env > 0) ks = ks_env; } // (sweeps)
    if ((size_t)ks * M * N > c->slab_floats) ks = std::max<int>(1, (int)(c->slab_floats / ((size_t)M * N)));
    int kslice = ((K + ks - 1) / ks + NVQA_WB_BK - 1) / NVQA_WB_BK * NVQA_WB_BK;
    ks = (K + kslice - 1) / kslice;
    if (M % 8 || lda % 8) A16 = nullptr;
    if (N % 8 || ldb % 8) B16 = nullptr;
    {
        ProfScope ps(c, PF_GEMM_WGRAD, 2.0 * M * N * K, ((double)K * (M + N) * (A16 ? 2 : 4) + (double)ks * M * N * 4), st);
        WgradBf16Args g{A, Bm, A16, B16, ks == 1 ? dW : slabs, ks == 1 ? 0 : (size_t)M * N, lda, ldb, N, M, N, K, kslice};
        const dim3 grid((M + NVQA_WB_BM - 1) / NVQA_WB_BM, (N + NVQA_WB_BN - 1) / NVQA_WB_BN, ks);
        static bool attr = false;
        if (!attr) {
            NVQA_HIP(hipFuncSetAttribute((const void *)k_wgrad_bf16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES));
            NVQA_HIP(hipFuncSetAttribute((const void *)k_wgrad_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES));
            NVQA_HIP(hipFuncSetAttribute((const void *)k_wgrad_bf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NVQA_WB_LDS_BYTES));
            attr = true;
        }
        if (A16 && B16) hipLaunchKernelGGL((k_wgrad_bf16<true, true>), grid, dim3(256), NVQA_WB_LDS_BYTES, st, g);
        else if (A16) hipLaunchKernelGGL((k_wgrad_bf16<true, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, st, g);
        else hipLaunchKernelGGL((k_wgrad_bf16<false, false>), grid, dim3(256), NVQA_WB_LDS_BYTES, st, g);
        NVQA_HIP(hipGetLastError());
    }
    if (ks == 1) return 0;
    ProfScope ps(c, PF_REDUCE, 0, (double)(ks + 1) * M * N * 4, st);
    const size_t n4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((n4 + 255) / 256), dim3(256), 0, st, slabs, ks, n4, reinterpret_cast<float4 *>(dW));
    NVQA_HIP(hipGetLastError());
    return 0;
}

// dW[M x N] = A^T B with A stored [K][M], B stored [K][N]; K = TS*B is long, the output
// small: split K over blockIdx.z into slabs, then sum the slabs in order (deterministic).
static int wgrad(nvqa_ctx *c, const float *A, int lda, const float *Bm, int ldb, int M, int N, int K,
                 float *dW, float *slabs, hipStream_t st, const unsigned short *A16 = nullptr, const unsigned short *B16 = nullptr)
{
    if (c->bf16 && c->wgrad_tr && M % 4 == 0 && N % 4 == 0) return wgrad_bf16(c, A, A16, lda, Bm, B16, ldb, M, N, K, dW, slabs, st);
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int ks = 1;
    while (ks < 16 && tiles * ks < 512 && K / (ks * 2) >= 256) ks *= 2; // tools/kbench11: 2048 x 200 x 13312: 16 slices 115 us, 8 slices 129 us
    if ((size_t)ks * M * N > c->slab_floats) ks = std::max<int>(1, (int)(c->slab_floats / ((size_t)M * N)));
    int kslice = (K + ks - 1) / ks;
    kslice = (kslice + 31) / 32 * 32;
    ks = (K + kslice - 1) / kslice;
    {
        ProfScope ps(c, PF_GEMM_WGRAD, 2.0 * M * N * K, ((double)K * (M + N) + (double)ks * M * N) * 4, st);
        GemmArgs g = mkargs(A, lda, Bm, ldb, M, N, K, kslice);
        g.kseg_limits = c->nrows; g.seg_rows = c->d.B; // K rows = (step, sorted row): skip the zero padding
        if (ks == 1) {
            NVQA_TRY((gemm_big<A_MC, B_NC, EpiStore, true>(c, g, EpiStore{dW, N, 0}, st)));
            return 0;
        }
        NVQA_TRY((gemm_big<A_MC, B_NC, EpiStore, true>(c, g, EpiStore{slabs, N, (size_t)M * N}, st)));
    }
    ProfScope ps(c, PF_REDUCE, 0, (double)(ks + 1) * M * N * 4, st);
    const size_t n4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((n4 + 255) / 256), dim3(256), 0, st, slabs, ks, n4,
                       reinterpret_cast<float4 *>(dW));
    NVQA_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// LSTM: shared by arch1 and arch2 (misc/LSTM.lua, misc/LSTM_encoder.lua have the same cell)
// X0 [TS*B][E] holds the layer-0 inputs; nrows[t] rows are active at step t.
// ------------------------------------------------------------------------------------
static int lstm_forward(nvqa_ctx *c, const Drop &dr)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, R = d.R, L = d.L, TS = c->TS, TB = TS * B;
    c->img_fwd_valid = c->img_bwd_valid = false;
    c->pb_bias_rb = 0;
    c->pb_bias_done = false;
    if (const int MT = persist_rows(c)) { c->img_fwd_valid = c->bf16; return lstm_forward_persist(c, dr, MT); }
    // Layer 0 takes W_i2h x_t as a first K segment inside the level kernel, like the layers above it: the
    // time-batched projection (0.128 ms, a 109 MB write and its re-read by the level epilogues) costs more than
    // the 0.074 ms the extra K = E adds to the 27 levels, and the two layers' workgroups become closer in length
    // (12 vs 16 K-tiles instead of 8 vs 16).  NVQA_FOLD_I2H=0 restores the batched projection (A/B runs).
    const bool fold0 = c->fold_i2h; // NVQA_FOLD_I2H, read at nvqa_create
    if (!fold0) {   // layer 0: time-batched input projection + both biases (LSTM.lua:41-43), off the chain
        const int in = d.E;
        ProfScope ps(c, PF_GEMM_I2H, 2.0 * TB * 4 * R * in, ((double)TB * (in + 4 * R) + 4.0 * R * in) * 4);
        GemmArgs g = mkargs(c->X0, in, c->P + c->lo.w_i2h[0], in, TB, 4 * R, in);
        g.mseg_limits = c->nrows; g.seg_rows = B; // rows of not-yet-started questions are skipped
        // K-contiguous x K-contiguous: 64x64 tiles, 16x16x4 MFMA, 2 K-groups measured 119 TF vs 99 TF for
        // the 128x128 32x32x2 form (tools/kbench3)
        typedef Cfg<16, 64, 64, 32, 4, 2, 1, 1> CfgI2h;
        const EpiBias2 e{c->Gt[0], 4 * R, c->P + c->lo.b_i2h[0], c->P + c->lo.b_h2h[0]};
        if (c->bf16) NVQA_HIP((launch_gemm<WithBF<CfgI2h>::type, A_KC, B_KC, false, EpiBias2>(c->s, g, e)));
        else NVQA_HIP((launch_gemm<CfgI2h, A_KC, B_KC, false, EpiBias2>(c->s, g, e)));
    }
    // Wavefront over (layer, step): layer l at step t needs layer l at t-1 and layer l-1 at t, so
    // diagonal dg = t + l holds up to L independent steps; they go out as ONE launch.
    for (int dg = 0; dg < TS + L - 1; ++dg) {
        MultiArgs<EpiLstmFwd> ma;
        int np = 0;
        double flops = 0, bytes = 0;
        for (int l = 0; l < L; ++l) {
            const int t = dg - l;
            if (t < 0 || t >= TS) continue;
            EpiLstmFwd &e = ma.e[np];
            e = EpiLstmFwd{};
            e.gx = c->Gt[l] + (size_t)t * B * 4 * R;
            e.c_prev = c->Cs[l] + (size_t)t * B * R;
            e.c = c->Cs[l] + (size_t)(t + 1) * B * R;
            e.h = c->Hs[l] + (size_t)(t + 1) * B * R;
            e.u_next = l + 1 < L ? c->U[l + 1] + (size_t)t * B * R : nullptr;
            e.bias1 = l == 0 && !fold0 ? nullptr : c->P + c->lo.b_i2h[l];
            e.bias2 = l == 0 && !fold0 ? nullptr : c->P + c->lo.b_h2h[l];
            e.nrows = c->nrows + t;
            e.sort_idx = c->sort_idx;
            e.R = R; e.B = B; e.T = TS; e.t = t; e.lnext_m1 = l;
            e.dr = dr;
            const float *hprev = c->Hs[l] + (size_t)t * B * R;
            // segment 1: Dropout(h^{l-1}_t) x W_i2h^T (layers >= 1; layer 0 reads the batched projection)
            // segment 2: h^l_{t-1} x W_h2h^T (absent at step 0: h_{-1} = 0)
            GemmArgs &g = ma.g[np];
            if (l == 0) g = mkargs(c->X0 + (size_t)t * B * d.E, d.E, c->P + c->lo.w_i2h[0], d.E, B, R, fold0 ? d.E : 0, 0, R, c->nrows + t);
            else g = mkargs(c->U[l] + (size_t)t * B * R, R, c->P + c->lo.w_i2h[l], R, B, R, R, 0, R, c->nrows + t);
            // h_{-1} = 0: no recurrent product at step 0 -- except under NVQA_QUIRK_H0, where the top layer's h0 is live
            const bool h0_live = t > 0 || (d.arch == NVQA_ARCH2 && (c->quirks & NVQA_QUIRK_H0) && l == L - 1);
            g.A2 = hprev; g.lda2 = R; g.B2 = c->P + c->lo.w_h2h[l]; g.ldb2 = R; g.K2 = h0_live ? R : 0;
            const double kk = (l == 0 ? (fold0 ? d.E : 0) : R) + (h0_live ? R : 0);
            flops += 2.0 * B * 4 * R * kk;
            bytes += ((double)B * kk + 4.0 * R * kk + (double)B * 4 * R * 2) * 4;
            ++np;
        }
        ProfScope ps(c, PF_LSTM_FWD, flops, bytes);
        static const int fwd_map = [] { const char *e = getenv("NVQA_FWD_MAP"); return e ? atoi(e) : 0; }();
        ma.xcd = fwd_map;
        if (c->bf16) NVQA_HIP((launch_gemm_multi<WithBF<CfgLstmFwd>::type, A_KC, B_KC, true, EpiLstmFwd, 1>(c->s, ma, np)));
        else NVQA_HIP((launch_gemm_multi<CfgLstmFwd, A_KC, B_KC, true, EpiLstmFwd, 1>(c->s, ma, np)));
    }
    return 0;
}

// BPTT.  On entry dCT/dHT [L][B][R] hold dL/d(final c, h) per layer (sorted row order).
// On exit Gt[l] holds d(pre-activations) for every step, G holds the LSTM weight gradients,
// and (dX0 != NULL) dX0 holds dL/d(layer-0 input).  Same wavefront as the forward pass, top
// layer first; the time-batched weight-gradient GEMMs of a layer start on the low-priority
// bulk stream as soon as that layer's last step is done.
static int lstm_backward(nvqa_ctx *c, const Drop &dr)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, R = d.R, L = d.L, TS = c->TS;
    {
        int RB = 0;
        if (const int MT = persist_bwd_rows(c, &RB)) { c->img_bwd_valid = c->bf16; return lstm_backward_persist(c, dr, MT, RB); }
    }
    // ragged arch1 batches (or lengths known only on the device): the products pick their split-K depth from nrows[s] on
    // the device -- a level with 2 of 8 row tiles active runs 16 short K slices instead of 4 long ones; the launch's workgroups are
    // re-dealt over (active row tile, column tile, slice).  NVQA_BWD_ZADAPT=0 switches it off.
    c->dct_zero = false; // (the per-level finisher carries the cell gradient in dCT, in place)
    static const bool zad_on = [] { const char *e = getenv("NVQA_BWD_ZADAPT"); return !(e && e[0] == '0'); }();
    const bool zad = zad_on && d.arch == NVQA_ARCH1 && !c->batch_uniform && (4 * R) % (NVQA_BWD_ZMAX * 32) == 0;
    const int Zl = zad ? NVQA_BWD_ZMAX : NVQA_BWD_Z;
    for (int dg = 0; dg < TS + L - 1; ++dg) {
        // diagonal dg: layer l (from the top: j = L-1-l) at step s = TS-1 - (dg - j).
        // Products of the level: dG_{s+1} W_h2h (none at the last step) and, below the top layer,
        // dG^{l+1}_s W_i2h^{l+1}; all are [B x 4R] x [4R x R] -> split-K GEMMs into slabs in ONE launch.
        MultiArgs<EpiStore> ma;
        BwdFinish fin;
        fin.Z = NVQA_BWD_Z; fin.B = B; fin.R = R; fin.zadapt = zad ? NVQA_BWD_ZMAX : 0; fin.xcd2d = 0;
        int np = 0, nf = 0;
        double flops = 0, bytes = 0;
        const size_t slab = (size_t)B * R;
        const size_t slab_st = slab; // stride between the K slices' slabs: [B][R] row-major
        for (int l = L - 1; l >= 0; --l) {
            const int s = TS - 1 - (dg - (L - 1 - l));
            if (s < 0 || s >= TS) continue;
            const bool top = l == L - 1, last = s == TS - 1;
            EpiLstmBwd &e = fin.e[nf];
            e = EpiLstmBwd{};
            e.gates = c->Gt[l] + (size_t)s * B * 4 * R;
            e.c_prev = c->Cs[l] + (size_t)s * B * R;
            e.c = c->Cs[l] + (size_t)(s + 1) * B * R;
            e.dc = c->dCT + (size_t)l * B * R;
            e.dh_ext = nullptr;
            e.dh_ext2 = (last || d.arch == NVQA_ARCH2) ? c->dHT + (size_t)l * B * R : nullptr;
            e.tlast = d.arch == NVQA_ARCH2 ? c->tinfo + 1 : nullptr;
            e.nrows = c->nrows + s;
            e.R = R;
            e.sort_idx = c->sort_idx; e.B = B; e.T = TS; e.s = s; e.lm1 = l; e.dr = dr;
            e.has_upper = top ? 0 : 1;
            float *srec = c->chain_slabs + ((size_t)l * 2 + 0) * Zl * slab_st;
            float *sup = c->chain_slabs + ((size_t)l * 2 + 1) * Zl * slab_st;
            fin.srec[nf] = last ? nullptr : srec;
            fin.sup[nf] = top ? nullptr : sup;
            if (!last) {
                ma.g[np] = mkargs(c->Gt[l] + (size_t)(s + 1) * B * 4 * R, 4 * R, c->P + c->lo.w_h2h[l], R, B, R, 4 * R,
                                  4 * R / NVQA_BWD_Z, 0, c->nrows + s);
                ma.e[np] = EpiStore{srec, R, slab_st};
                ++np;
            }
            if (!top) {
                ma.g[np] = mkargs(c->Gt[l + 1] + (size_t)s * B * 4 * R, 4 * R, c->P + c->lo.w_i2h[l + 1], R, B, R, 4 * R,
                                  4 * R / NVQA_BWD_Z, 0, c->nrows + s);
                ma.e[np] = EpiStore{sup, R, slab_st};
                ++np;
            }
            const double nseg = (last ? 0 : 1) + (top ? 0 : 1);
            flops += 2.0 * B * 4 * R * R * nseg;
            bytes += ((double)B * 4 * R * (1 + nseg) + 4.0 * R * R * nseg + (double)B * R * 5) * 4;
            ++nf;
        }
        if (np > 0) {
            ProfScope ps(c, PF_LSTM_BWD, flops, bytes);
            ma.zsplit = NVQA_BWD_Z; ma.zadapt = zad ? NVQA_BWD_ZMAX : 0;
            // tile -> XCD map: (row half, column quarter) per XCD where the tile grid allows it (each L2 then sees half of
            // every dG and a quarter of every W: 0.3 % of the step against the natural order, in which every XCD reads all
            // of dG -- 105 MB of fabric traffic per level, PMC); NVQA_BWD_MAP=0 restores the natural order
            static const int bmap = [] { const char *e = getenv("NVQA_BWD_MAP"); return e ? atoi(e) : 3; }();
            const bool grid2d = ((R + CfgBwdLevel::BN - 1) / CfgBwdLevel::BN) % 4 == 0 && ((B + CfgBwdLevel::BM - 1) / CfgBwdLevel::BM) % 2 == 0;
            ma.xcd = bmap == 3 && grid2d && !zad ? 3 : xcd_order();
            if (c->bf16) NVQA_HIP((launch_gemm_multi<WithBF<CfgBwdLevel>::type, A_KC, B_NC, false, EpiStore, 0>(c->s, ma, np)));
            else NVQA_HIP((launch_gemm_multi<CfgBwdLevel, A_KC, B_NC, false, EpiStore, 0>(c->s, ma, np)));
        }
        {
            // the finisher: slab sums + fused cell backward of every cell problem of the level
            BwdFinish rest = fin;
            rest.xcd2d = (np > 0 && ma.xcd == 3 && R % 4 == 0 && B % 2 == 0 && ((size_t)B * R / 8) % 256 == 0) ? 1 : 0;
            if (nf > 0) {
                ProfScope ps(c, PF_LSTM_BWD_FIN, 0, (double)nf * slab * (2.0 * NVQA_BWD_Z + 14) * 4);
                hipLaunchKernelGGL(k_lstm_bwd_finish, dim3((unsigned)((slab + 255) / 256), nf), dim3(256), 0, c->s, rest);
            }
        }
        NVQA_HIP(hipGetLastError());
    }
    return 0;
}

// Weight gradients of layer l, sums over all steps (002_train_baseline.lua:323-326): two time-batched
// split-K products and the bias column sums.  They run after the chains, one after the other on the main
// stream: each fills the chip by itself (split-K), so extra streams bought nothing (measured), and
// overlapping them with the latency-critical chain kernels only slowed the chain (5.2 vs 4.5 ms per step).
// In data parallel the layer's slice of the flat gradient goes out right behind them (reduce_range), so
// only the last, smallest slice is not hidden under compute.
static int reduce_range(nvqa_ctx *c, size_t off, size_t count, int ev);
static int lstm_wgrads(nvqa_ctx *c, int l)
{
    const nvqa_dims &d = c->d;
    const int R = d.R, TB = c->TS * d.B;
    const int in = l == 0 ? d.E : R;
    const float *Xin = l == 0 ? c->X0 : c->U[l];
    // bf16 images of the operands, where this step's persistent bf16 kernels left them (lstm_persist.h / lstm_persist_bwd2.h)
    const size_t hs = (size_t)(c->TS + 1) * d.B * R, us = (size_t)c->TS * d.B * R;
    const unsigned short *G16 = c->img_bwd_valid ? c->dg_b16 + (size_t)l * TB * 4 * R : nullptr;
    const unsigned short *H16 = c->img_fwd_valid ? c->act_b16 + l * hs : nullptr;
    const unsigned short *X16 = l == 0 ? (c->x0_img_valid ? c->x0_b16 : nullptr)
                                       : (c->img_fwd_valid ? c->act_b16 + d.L * hs + l * us : nullptr);
    NVQA_TRY(wgrad(c, c->Gt[l], 4 * R, c->Hs[l], R, 4 * R, R, TB, c->G + c->lo.w_h2h[l], c->slabs, c->s, G16, H16));
    NVQA_TRY(wgrad(c, c->Gt[l], 4 * R, Xin, in, 4 * R, in, TB, c->G + c->lo.w_i2h[l], c->slabs, c->s, G16, X16));
    if (c->pb_bias_rb > 0 && c->pb_bias_done) {
        // the persistent BPTT kernel of this step wrote both bias gradients of every layer itself (its last workgroup per unit tile)
    } else if (c->pb_bias_rb > 0) { // ... or left the column sums per row block (lstm_persist_bwd2.h; NVQA_BIAS_IN_BPTT=0)
        ProfScope ps(c, PF_COLSUM, 0, (double)c->pb_bias_rb * 4 * R * 4);
        hipLaunchKernelGGL(k_bias_sum, dim3((4 * R + 255) / 256), dim3(256), 0, c->s, c->pb_bias + (size_t)l * c->pb_bias_rb * 4 * R, c->pb_bias_rb,
                           4 * R, c->G + c->lo.b_i2h[l], c->G + c->lo.b_h2h[l]);
        NVQA_HIP(hipGetLastError());
    } else {
        NVQA_TRY(colsum(c, c->Gt[l], TB, 4 * R, 4 * R, c->G + c->lo.b_i2h[l], c->G + c->lo.b_h2h[l], c->s));
    }
    NVQA_TRY(reduce_range(c, c->lo.w_i2h[l], c->lo.b_h2h[l] + 4 * (size_t)R - c->lo.w_i2h[l], 3 + l));
    return 0;
}

// W [rows][cols] f32 -> its transpose [cols][rows] in bf16 (round to nearest even), 32 x 32 tiles through LDS
__global__ void k_transpose_to_bf16(const float *W, int rows, int cols, __bf16 *out)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, cc = c0 + threadIdx.x;
        t[i][threadIdx.x] = r < rows && cc < cols ? W[(size_t)r * cols + cc] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int cc = c0 + i, r = r0 + threadIdx.x;
        if (cc < cols && r < rows) out[(size_t)cc * rows + r] = (__bf16)t[threadIdx.x][i];
    }
}

static bool dx0_b2_enabled() // NVQA_DX0_B2=0: d(layer-0 input) in the BF = 1 form (A/B runs)
{
    static const bool on = [] { const char *e = getenv("NVQA_DX0_B2"); return !(e && e[0] == '0'); }();
    return on;
}

// dL/d(layer-0 input) for all steps at once; runs after the LSTM weight gradients so that their
// all-reduce (data parallel) overlaps it.
static int lstm_dx0(nvqa_ctx *c, float *dX0)
{
    const nvqa_dims &d = c->d;
    const int R = d.R, TB = c->TS * d.B;
    ProfScope ps(c, PF_GEMM_DGRAD, 2.0 * TB * d.E * 4 * R, ((double)TB * (4 * R + d.E) + 4.0 * R * d.E) * 4);
    // bf16 mode behind the persistent BPTT kernel: dG exists as a bf16 image ([TS*B][4R], K-contiguous); with a transposed
    // bf16 image of W_i2h (made here, per step: the weights move) both operands are K-contiguous bf16 and the product runs
    // in the gfx950 form (BF = 2: bf16 LDS images, v_mfma_f32_16x16x32_bf16) instead of rounding f32 LDS images per MFMA.
    // Same operand values, f32 accumulate, f32 result.  NVQA_DX0_B2=0: the BF = 1 form (A/B runs).
    if (c->bf16 && c->img_bwd_valid && dx0_b2_enabled() && (4 * R) % 64 == 0) {
        if (!c->wi2h0_img_step) { // (normally made by the embedding launch of this step: wi2h0_image_job)
            if (!c->wi2h0_t16) NVQA_HIP(hipMalloc((void **)&c->wi2h0_t16, (size_t)d.E * 4 * R * 2));
            hipLaunchKernelGGL(k_transpose_to_bf16, dim3((d.E + 31) / 32, (4 * R + 31) / 32), dim3(32, 8), 0, c->s, c->P + c->lo.w_i2h[0], 4 * R, d.E,
                               reinterpret_cast<__bf16 *>(c->wi2h0_t16));
        }
        GemmArgs g = {};
        g.A = reinterpret_cast<const float *>(c->dg_b16); g.lda = 4 * R / 2; // sizes in storage floats (two bf16)
        g.B = reinterpret_cast<const float *>(c->wi2h0_t16); g.ldb = 4 * R / 2;
        g.M = TB; g.N = d.E; g.K = g.kslice = 4 * R / 2;
        g.mseg_limits = c->nrows; g.seg_rows = d.B;
        g.xcd = d.arch == NVQA_ARCH1 && !c->batch_uniform ? 0 : xcd_order();
        if (d.E > 256) NVQA_HIP((launch_gemm<CfgDx0B, A_KC, B_KC, false, EpiStore>(c->s, g, EpiStore{dX0, d.E, 0})));
        else NVQA_HIP((launch_gemm<CfgDx0B64, A_KC, B_KC, false, EpiStore>(c->s, g, EpiStore{dX0, d.E, 0})));
        return 0;
    }
    GemmArgs g = mkargs(c->Gt[0], 4 * R, c->P + c->lo.w_i2h[0], d.E, TB, d.E, 4 * R);
    g.mseg_limits = c->nrows; g.seg_rows = d.B;
    // ragged arch1 batch: the XCD-contiguous tile order would give the early steps (few active rows: their row tiles
    // leave at once) to the first XCDs and the full late steps to the last ones -- the launch then takes as long as a
    // full-length batch (128 us either way); dealt round-robin the XCDs share the active tiles
    if (d.arch == NVQA_ARCH1 && !c->batch_uniform) g.xcd = 0;
    // N = E: 128-wide tiles waste (256 - 200) / 256 of the MFMA work at the reference's E = 200, 32-wide ones 24 / 224
    const int w128 = (d.E + 127) / 128 * 128, w32 = (d.E + 31) / 32 * 32;
    if (w32 * 10 <= w128 * 9) {
        if (c->bf16) NVQA_HIP((launch_gemm<WithBF<CfgNarrow>::type, A_KC, B_NC, false, EpiStore>(c->s, g, EpiStore{dX0, d.E, 0})));
        else NVQA_HIP((launch_gemm<CfgNarrow, A_KC, B_NC, false, EpiStore>(c->s, g, EpiStore{dX0, d.E, 0})));
        return 0;
    }
    NVQA_TRY((gemm_big<A_KC, B_NC>(c, g, EpiStore{dX0, d.E, 0})));
    return 0;
}

// Embedding / lookup-table gradient by token segments (kernels.h: k_tok_index + k_emb_bwd_seg).  The index of the packed
// token list is built on a side stream as soon as the forward pass has written ptok, under the LSTM unroll.
static bool emb_index_ok(const nvqa_ctx *c, int VT, int NP)
{
    return c->tok_seg && NP <= NVQA_TI_MAXNP && VT <= 65535 && tok_index_lds(VT, NP) <= 160 * 1024 && c->d.E <= 512;
}
static TokIndexArgs tok_index_args(nvqa_ctx *c, int VT, int NP)
{
    return TokIndexArgs{c->ptok, NP, VT, c->seg_start, c->perm, c->pslot, c->seg_done, c->pslot + (NP / NVQA_ES_SHORT + 2)};
}
static int tok_index_launch(nvqa_ctx *c, int VT, int NP)
{
    static bool attr = false;
    if (!attr) { NVQA_HIP(hipFuncSetAttribute((const void *)k_tok_index, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    ProfScope ps(c, PF_ASSEMBLE, 0, 0);
    hipLaunchKernelGGL(k_tok_index, dim3(1), dim3(NVQA_TI_THREADS), tok_index_lds(VT, NP), c->s, tok_index_args(c, VT, NP));
    NVQA_HIP(hipGetLastError());
    return 0;
}
// The index of this step's packed token list.  Where the step's BPTT is the persistent launch, the job rides in one of that
// launch's idle workgroups (ride_jobs.h; c->ride is consumed by lstm_backward_persist; NVQA_TOK_IN_BPTT=0: own kernel);
// otherwise it is a one-workgroup kernel of its own on the main stream.  (Round 3 also tried that kernel on the side stream
// behind the forward kernel, under the head's GEMMs: SLOWER -- 3.175 vs 3.144 ms per step: with its 160 KB of LDS it waits
// for a whole CU to drain and then delays whatever needs that CU next, the persistent BPTT launch in the worst case.)
static int emb_index_begin(nvqa_ctx *c, int VT, int NP)
{
    c->tok_job_pending = false;
    if (!emb_index_ok(c, VT, NP)) return 0;
    int RB = 0;
    if (c->tok_in_bptt_on && persist_bwd_rows(c, &RB)) {
        c->ride.tok = tok_index_args(c, VT, NP);
        c->tok_job_pending = true;
        return 0;
    }
    return tok_index_launch(c, VT, NP);
}
// dWeT [VT][E] = sum over the packed positions of each token; plain = 1: nn.LookupTable (arch2), 0: arch1's Tanh + Dropout
static int emb_backward(nvqa_ctx *c, int VT, int NP, int T, const float *dX, const Drop &dr, float *dWeT, int plain)
{
    const nvqa_dims &d = c->d;
    const int E = d.E, B = d.B;
    if (c->tok_job_pending) { // no persistent BPTT launch took the job after all (fallback path): build the index now
        c->tok_job_pending = false;
        NVQA_TRY(tok_index_launch(c, VT, NP));
    }
    ProfScope ps(c, PF_EMB_BWD, 0, (2.0 * NP * E + (double)VT * E) * 4);
    if (emb_index_ok(c, VT, NP)) {
        const int slots = NP / NVQA_ES_SHORT + 2; // c->pslot: [slots] tokens of the long segments, then their number
        const int32_t *nlong = c->pslot + slots;
        const int gs = (VT + 3) / 4, gl = (slots * NVQA_ES_CHUNKS + 3) / 4;
        const LatchArgs la = latch_take(c, 1); // the BPTT launch's err latch rides as the last workgroup (latch.h)
        const dim3 grid(gs + gl + (la.cnt ? 1 : 0));
#define NVQA_ES_GO(NP_)                                                                                                                        \
    hipLaunchKernelGGL(k_emb_bwd_tok<NP_>, grid, dim3(256), 0, c->s, c->seg_start, c->perm, c->pslot, nlong, c->seg_done, c->seg_part,          \
                       (unsigned)c->seg_part_bytes, c->X0, dX, c->sort_idx, B, T, VT, E, dr, dWeT, plain, gs, gl, la)
        if (E <= 256) { NVQA_ES_GO(1); } else { NVQA_ES_GO(2); }
#undef NVQA_ES_GO
    } else {
        const int waves = plain ? 8 : 4; // arch1 tokens spread over the vocabulary: 4 scanning waves (32 KB of LDS, 4 workgroups per CU) beat 8
        const int blocks = (VT + NVQA_EB_ROWS - 1) / NVQA_EB_ROWS;
        hipLaunchKernelGGL(k_emb_bwd, dim3(blocks, (E + NVQA_EB_COLS - 1) / NVQA_EB_COLS), dim3(64 * waves),
                           (size_t)waves * NVQA_EB_ROWS * NVQA_EB_COLS * sizeof(float), c->s, c->ptok, c->X0, dX, c->sort_idx, NP, B, T, VT, E, dr,
                           dWeT, plain);
    }
    NVQA_HIP(hipGetLastError());
    return 0;
}

// bf16 mode: the embedding kernels also leave a bf16 image of the layer-0 inputs -- the B operand of layer 0's input weight
// gradient (k_wgrad_bf16<true, true>: 47 instead of 58 us) and, where E = R (arch2), the persistent forward kernel's layer-0
// input segment (half the bytes per chunk, no rounding on the way into LDS).  Same values: f32 rounded to nearest even.
static int x0_image_begin(nvqa_ctx *c, int TB)
{
    c->x0_img_valid = false;
    static const bool on = [] { const char *e = getenv("NVQA_X0_B16"); return !(e && e[0] == '0'); }();
    if (!c->bf16 || !on || c->d.E % 8) return 0;
    if (!c->x0_b16) NVQA_HIP(hipMalloc((void **)&c->x0_b16, (size_t)TB * c->d.E * 2));
    c->x0_img_valid = true;
    return 0;
}

// bf16 mode, training step, persistent BPTT ahead: the bf16 image of W_i2h[0]^T that lstm_dx0's gfx950-form product wants is
// made by extra workgroups of the embedding launch (the weights do not move during a step).
static int wi2h0_image_job(nvqa_ctx *c, bool train, int first_block, TransposeJob *tj)
{
    *tj = TransposeJob{};
    c->wi2h0_img_step = false;
    int rb = 0;
    const int R = c->d.R, E = c->d.E;
    if (!train || !c->bf16 || !dx0_b2_enabled() || (4 * R) % 64 || persist_bwd_rows(c, &rb) == 0) return 0;
    if (!c->wi2h0_t16) NVQA_HIP(hipMalloc((void **)&c->wi2h0_t16, (size_t)E * 4 * R * 2));
    tj->W = c->P + c->lo.w_i2h[0]; tj->out = c->wi2h0_t16; tj->rows = 4 * R; tj->cols = E;
    tj->first_block = first_block; tj->nblocks = ((E + 31) / 32) * ((4 * R + 31) / 32);
    c->wi2h0_img_step = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// arch1
// ------------------------------------------------------------------------------------
static int arch1_forward(nvqa_ctx *c, const Drop &dr, bool train, bool want_argmax)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, T = d.T, R = d.R, L = d.L, E = d.E, I = d.I, C = d.C, A = d.A, Q = 2 * R * L;
    const int TB = T * B;
    const int Zh = 4;                 // K slices of the head's projections
    const size_t nBC = (size_t)B * C;
    // f32, persistent forward launch ahead: the image projection W_v Dropout(v) rides in that launch's layer-0 workgroups, which
    // are done ~30 % before it ends (lstm_persist.h: PersistFwdArgs::fr).  NVQA_RIDE_FWD=0: in the head's own launch.
    const bool fr = c->ride_fwd_on && !c->bf16 && persist_rows(c) > 0 && Q % (32 * Zh) == 0 && I % 64 == 0 && (Zh + 1) * nBC <= c->slab_floats;
    {
        ProfScope ps(c, PF_ASSEMBLE);
        hipLaunchKernelGGL(k_sort_lengths, dim3(1), dim3(1024), (3 + 16) * (T + 1) * sizeof(int), c->s, c->len, B, T,
                           c->sort_idx, c->sort_inv, c->nrows);
    }
    {
        ProfScope ps(c, PF_EMB_FWD, 0, 2.0 * TB * E * 4);
        NVQA_TRY(x0_image_begin(c, TB));
        const int rows_blocks = (TB + 3) / 4, vd_blocks = fr ? B : 0;
        TransposeJob tj;
        NVQA_TRY(wi2h0_image_job(c, train, rows_blocks + vd_blocks, &tj));
        hipLaunchKernelGGL(k_emb_fwd, dim3(rows_blocks + vd_blocks + tj.nblocks), dim3(256), 0, c->s, c->tok, c->sort_idx, c->nrows,
                           c->P + c->lo.w_e, c->P + c->lo.b_e, B, T, E, dr, c->X0, c->ptok, c->x0_img_valid ? c->x0_b16 : nullptr, tj,
                           c->img, fr ? c->vd : (float *)nullptr, I, rows_blocks);
    }
    if (fr) { // sv = Dropout(v) W_v^T, whole K per 64 x 64 tile, into the slab the finisher reads (k_head_fuse: Zv = 1)
        RideGemm &r = c->fwd_ride;
        r.g = mkargs(c->vd, I, c->P + c->lo.w_v, I, B, C, I);
        r.e = EpiStore{c->slabs + Zh * nBC, C, nBC};
        r.tx = (C + CfgRide::BN - 1) / CfgRide::BN; r.ty = (B + CfgRide::BM - 1) / CfgRide::BM;
        c->fwd_ride_pending = true;
    }
    NVQA_HIP(hipGetLastError());
    NVQA_TRY(lstm_forward(c, dr));
    c->fwd_ride_pending = false;
    const bool fr_done = fr && c->fwd_ride_done; // (fr without fr_done: the forward took another route after all; Dropout(v) exists either way)
    if (train) NVQA_TRY(emb_index_begin(c, d.V, TB));
    {
        ProfScope ps(c, PF_HEAD_PREP, 0, 2.0 * B * (Q + (fr ? 0 : I)) * 4);
        const LatchArgs la = latch_take(c, 0); // the forward launch's err latch rides as an extra workgroup (latch.h)
        hipLaunchKernelGGL(k_head_prep, dim3(B + (la.cnt ? 1 : 0)), dim3(256), 0, c->s, c->Cs[0] + (size_t)T * B * R,
                           c->Hs[0] + (size_t)T * B * R, (size_t)(T + 1) * B * R, c->sort_inv, c->img, B, R, L, I, dr, c->qd, fr ? (float *)nullptr : c->vd, la);
    }
    NVQA_HIP(hipGetLastError());
    {
        ProfScope ps(c, PF_GEMM_HEAD_FWD, 2.0 * B * ((double)C * Q + (fr_done ? 0.0 : (double)C * I) + (double)A * C),
                     ((double)C * Q + (double)C * I + (double)A * C) * 4);
        // qc = tanh(W_q Dropout(q) + b_q), ic = tanh(W_v Dropout(v) + b_v), zd = Dropout(qc (*) ic):
        // both projections as one split-K multi-problem launch into slabs, then k_head_fuse
        if (fr_done) { // the image projection came from the forward launch (one slab): only the question projection is left
            MultiArgs<EpiStore> ma;
            ma.g[0] = mkargs(c->qd, Q, c->P + c->lo.w_q, Q, B, C, Q, Q / Zh);
            ma.e[0] = EpiStore{c->slabs, C, nBC};
            ma.zsplit = Zh;
            NVQA_HIP((launch_gemm_multi<CfgMedMulti, A_KC, B_KC, false, EpiStore, 0>(c->s, ma, 1)));
            hipLaunchKernelGGL(k_head_fuse, dim3((unsigned)((nBC + 255) / 256)), dim3(256), 0, c->s, c->slabs,
                               c->slabs + Zh * nBC, Zh, 1, nBC, C, c->P + c->lo.b_q, c->P + c->lo.b_v, dr, c->qc, c->ic, c->zd, c->fusion_askip);
        } else
        if (Q % (32 * Zh) == 0 && I % (32 * Zh) == 0 && 2 * Zh * nBC <= c->slab_floats) {
            MultiArgs<EpiStore> ma;
            ma.g[0] = mkargs(c->qd, Q, c->P + c->lo.w_q, Q, B, C, Q, Q / Zh);
            ma.e[0] = EpiStore{c->slabs, C, nBC};
            ma.g[1] = mkargs(c->vd, I, c->P + c->lo.w_v, I, B, C, I, I / Zh);
            ma.e[1] = EpiStore{c->slabs + Zh * nBC, C, nBC};
            ma.zsplit = Zh;
            if (c->bf16) NVQA_HIP((launch_gemm_multi<WithBF<CfgMedMulti>::type, A_KC, B_KC, false, EpiStore, 0>(c->s, ma, 2)));
            else NVQA_HIP((launch_gemm_multi<CfgMedMulti, A_KC, B_KC, false, EpiStore, 0>(c->s, ma, 2)));
            hipLaunchKernelGGL(k_head_fuse, dim3((unsigned)((nBC + 255) / 256)), dim3(256), 0, c->s, c->slabs,
                               c->slabs + Zh * nBC, Zh, Zh, nBC, C, c->P + c->lo.b_q, c->P + c->lo.b_v, dr, c->qc, c->ic, c->zd, c->fusion_askip);
        } else {
            NVQA_TRY((gemm_med<A_KC, B_KC>(c, mkargs(c->qd, Q, c->P + c->lo.w_q, Q, B, C, Q),
                                           EpiBiasTanh{c->qc, C, c->P + c->lo.b_q})));
            NVQA_TRY((gemm_med<A_KC, B_KC>(c, mkargs(c->vd, I, c->P + c->lo.w_v, I, B, C, I),
                                           EpiFuse{c->ic, c->zd, c->qc, c->P + c->lo.b_v, C, dr, c->fusion_askip})));
        }
        // scores = W_o zd + b_o (zd: C wide, or [qc | ic] = 2C wide for netdef.A_B)
        const int ZW = c->fusion_askip == 2 ? 2 * C : C;
        NVQA_TRY((gemm_med<A_KC, B_KC>(c, mkargs(c->zd, ZW, c->P + c->lo.w_o, ZW, B, A, ZW),
                                       EpiBias2{c->scores, A, c->P + c->lo.b_o, nullptr})));
    }
    {
        ProfScope ps(c, PF_SOFTMAX_CE, 0, 2.0 * B * A * 4);
        hipLaunchKernelGGL(k_softmax_ce, dim3((B + 3) / 4), dim3(256), 0, c->s, c->scores,
                           train ? c->lab : (const int32_t *)nullptr, B, A, train ? c->dscores : (float *)nullptr,
                           c->rowloss, want_argmax ? c->argmax : (int32_t *)nullptr, train ? c->h_rowloss : (float *)nullptr);
        if (train) c->loss_rows = B;
    }
    NVQA_HIP(hipGetLastError());
    return 0;
}

// Head weight gradients that only the optimiser needs: in single-GPU runs they ride in the idle workgroups of the persistent
// BPTT launch (ride_jobs.h) instead of taking their time on the critical path in front of it -- with a communicator too, since
// round 4 exchanges their segment BEHIND the BPTT launch (nvqa_comm_init).  Only NVQA_DP_OVERLAP_BPTT=1 (the segment travels under
// the BPTT, and the idle slots belong to the collective's kernels) computes them in place, as NVQA_RIDE_GEMM=0 always does.
static bool ride_begin(nvqa_ctx *c)
{
    int rb = 0;
    c->ride.ngemm = 0;
    c->ride.has_colsum = 0;
    c->ride_gemm_pending = false;
    return c->ride_gemm_on && (!c->comm || !c->dp_overlap_bptt) && persist_bwd_rows(c, &rb) != 0;
}
static void ride_add(nvqa_ctx *c, const GemmArgs &g, const EpiStore &e)
{
    RideGemm &r = c->ride.gm[c->ride.ngemm++];
    r.g = g; r.e = e;
    r.tx = (g.N + CfgRide::BN - 1) / CfgRide::BN; r.ty = (g.M + CfgRide::BM - 1) / CfgRide::BM;
    c->ride_gemm_pending = true;
}
// behind lstm_backward: products the persistent launch did not take (fallback path, no free slot group) are computed now
static int ride_flush(nvqa_ctx *c)
{
    if (!c->ride_gemm_pending) return 0;
    c->ride_gemm_pending = false;
    double fl = 0;
    for (int i = 0; i < c->ride.ngemm; ++i) fl += 2.0 * c->ride.gm[i].g.M * c->ride.gm[i].g.N * c->ride.gm[i].g.K;
    ProfScope ps(c, PF_GEMM_HEAD_BWD, fl, 0);
    for (int i = 0; i < c->ride.ngemm; ++i) NVQA_TRY((gemm_med<A_MC, B_NC>(c, c->ride.gm[i].g, c->ride.gm[i].e)));
    if (c->ride.has_colsum) {
        hipLaunchKernelGGL(k_colsum_batch, dim3(c->ride.cs.first_block[4]), dim3(256), 0, c->s, c->ride.cs);
        NVQA_HIP(hipGetLastError());
    }
    return 0;
}

static int arch1_backward(nvqa_ctx *c, const Drop &dr)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, T = d.T, R = d.R, L = d.L, E = d.E, I = d.I, C = d.C, A = d.A, Q = 2 * R * L;
    const int TB = T * B, V = d.V;
    float *G = c->G;
    const bool ride = ride_begin(c); // (ride_jobs.h)
    {
        // (the two products that ride under the BPTT are booked there, if the launch takes them)
        ProfScope ps(c, PF_GEMM_HEAD_BWD,
                     2.0 * B * ((ride ? 1.0 : 2.0) * A * C + (ride ? 1.0 : 2.0) * C * Q + (double)C * I),
                     (2.0 * A * C + 2.0 * C * Q + 2.0 * C * I) * 4);
        // classifier: dW_o = dscores^T zd ; d(zd) = dscores W_o -> Dropout', CMul', Tanh'
        const int ZW = c->fusion_askip == 2 ? 2 * C : C;
        // dW_o and dW_q are needed only by the optimiser: in single-GPU runs they ride in the idle workgroups of the persistent
        // BPTT launch (ride_jobs.h) instead of taking 40 us of the critical path here (ride_begin says when).
        if (ride) ride_add(c, mkargs(c->dscores, A, c->zd, ZW, A, ZW, B), EpiStore{G + c->lo.w_o, ZW, 0});
        else NVQA_TRY((gemm_med<A_MC, B_NC>(c, mkargs(c->dscores, A, c->zd, ZW, A, ZW, B), EpiStore{G + c->lo.w_o, ZW, 0})));
        NVQA_TRY((gemm_med<A_KC, B_NC>(c, mkargs(c->dscores, A, c->P + c->lo.w_o, ZW, B, ZW, A),
                                       EpiHeadBwd{c->dqc, c->dic, c->qc, c->ic, C, dr, c->fusion_askip})));
        // fusion: dW_q = dqc^T qd ; dW_v = dic^T vd ; d(qd) = dqc W_q (no gradient to the image)
        if (ride) ride_add(c, mkargs(c->dqc, C, c->qd, Q, C, Q, B), EpiStore{G + c->lo.w_q, Q, 0});
        else NVQA_TRY((gemm_med<A_MC, B_NC>(c, mkargs(c->dqc, C, c->qd, Q, C, Q, B), EpiStore{G + c->lo.w_q, Q, 0})));
        NVQA_TRY((gemm_big<A_MC, B_NC>(c, mkargs(c->dic, C, c->vd, I, C, I, B), EpiStore{G + c->lo.w_v, I, 0})));
        NVQA_TRY((gemm_med<A_KC, B_NC>(c, mkargs(c->dqc, C, c->P + c->lo.w_q, Q, B, Q, C),
                                       EpiResort{c->dCT, c->dHT, c->sort_inv, B, R, Q, dr})));
    }
    {   // b_o, b_q, b_v: three B-row column sums in one launch
        ProfScope ps(c, PF_COLSUM, 0, (double)B * (A + 2.0 * C) * 4);
        ColsumBatch cb = {};
        const float *Xs[3] = {c->dscores, c->dqc, c->dic};
        float *outs[3] = {G + c->lo.b_o, G + c->lo.b_q, G + c->lo.b_v};
        const int Ns[3] = {A, C, C};
        int blocks = 0;
        for (int i = 0; i < 3; ++i) {
            cb.X[i] = Xs[i]; cb.out[i] = outs[i]; cb.M[i] = B; cb.N[i] = Ns[i]; cb.ld[i] = Ns[i];
            cb.first_block[i] = blocks;
            blocks += (Ns[i] + 63) / 64;
        }
        cb.first_block[3] = cb.first_block[4] = blocks;
        if (ride) { // with the head weight gradients above: under the BPTT (ride_jobs.h)
            c->ride.cs = cb;
            c->ride.has_colsum = 1;
            c->ride_gemm_pending = true;
        } else {
            hipLaunchKernelGGL(k_colsum_batch, dim3(blocks), dim3(256), 0, c->s, cb);
            NVQA_HIP(hipGetLastError());
        }
    }
    if (c->dp_overlap_bptt) NVQA_TRY(reduce_segment(c, 2)); // NVQA_DP_OVERLAP_BPTT=1: the multimodal segment's all-reduce hides under the BPTT
    float *dX0 = c->dX0;
    NVQA_TRY(lstm_backward(c, dr));
    NVQA_TRY(ride_flush(c));
    if (!c->dp_overlap_bptt) NVQA_TRY(reduce_segment(c, 2)); // default: behind the BPTT launch (no collective beside a persistent kernel), under what follows
    // embedding gradient first, so that its 11.8 MB all-reduce and those of the upper LSTM layers travel
    // under the weight-gradient GEMMs; only layer 0's slice (5.8 MB) is exchanged after the last kernel
    NVQA_TRY(lstm_dx0(c, dX0));
    NVQA_TRY(emb_backward(c, V, TB, T, dX0, dr, G + c->lo.w_e, 0));
    NVQA_TRY(colsum(c, G + c->lo.w_e, V, E, E, G + c->lo.b_e, nullptr));
    NVQA_TRY(reduce_segment(c, 1)); // embedding
    for (int l = L - 1; l >= 0; --l) NVQA_TRY(lstm_wgrads(c, l)); // + the layer's slice of the encoder segment
    return 0;
}

// ------------------------------------------------------------------------------------
// arch2 (003_train_vqa_arch2/002_train_baseline.lua:277-333, misc/Encoder_lstm.lua)
// ------------------------------------------------------------------------------------
static int arch2_forward(nvqa_ctx *c, const Drop &dr, bool train, bool want_argmax)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, T = d.T, R = d.R, L = d.L, E = d.E, I = d.I, A = d.A, TS = c->TS, TB = TS * B;
    {
        ProfScope ps(c, PF_ASSEMBLE);
        hipLaunchKernelGGL(k_arch2_tmax, dim3(1), dim3(1024), 0, c->s, c->tok, B, T, c->nrows, c->tinfo, c->sort_idx, c->sort_inv);
    }
    {   // x_1 = cnn_projection(fv_im): Linear(I, E), no dropout / non-linearity (:166,308) -> step-0 rows of X0
        ProfScope ps(c, PF_GEMM_HEAD_FWD, 2.0 * B * E * I, ((double)B * I + (double)E * I) * 4);
        NVQA_TRY((gemm_med<A_KC, B_KC>(c, mkargs(c->img, I, c->P + c->lo.w_p, I, B, E, I),
                                       EpiBias2{c->X0, E, c->P + c->lo.b_p, nullptr})));
    }
    {
        ProfScope ps(c, PF_EMB_FWD, 0, 2.0 * TB * E * 4);
        NVQA_TRY(x0_image_begin(c, TB));
        TransposeJob tj;
        NVQA_TRY(wi2h0_image_job(c, train, (TB + 3) / 4, &tj));
        hipLaunchKernelGGL(k_arch2_embed, dim3((TB + 3) / 4 + tj.nblocks), dim3(256), 0, c->s, c->tok, c->tinfo, c->P + c->lo.w_lk, B, T, d.V, E, c->X0, c->ptok,
                           c->x0_img_valid ? c->x0_b16 : nullptr, tj);
    }
    NVQA_HIP(hipGetLastError());
    if (c->quirks & NVQA_QUIRK_H0) // top-layer h0 = what the last backward left in the aliased tensor (Encoder_lstm.lua:238-239)
        NVQA_HIP(hipMemcpyAsync(c->Hs[L - 1], c->dHT + (size_t)(L - 1) * B * R, (size_t)B * R * 4, hipMemcpyDeviceToDevice, c->s));
    NVQA_TRY(lstm_forward(c, dr));
    if (train && !(c->quirks & NVQA_QUIRK_LOOKUP)) NVQA_TRY(emb_index_begin(c, d.V + 1, TB));
    {
        ProfScope ps(c, PF_HEAD_PREP, 0, 2.0 * B * R * 4);
        const LatchArgs la = latch_take(c, 0);
        hipLaunchKernelGGL(k_arch2_head_prep, dim3(B + (la.cnt ? 1 : 0)), dim3(256), 0, c->s, c->Hs[L - 1], c->tinfo, B, R, dr, c->qd, la);
    }
    {   // scores = Linear(R, A)(Dropout(h))
        ProfScope ps(c, PF_GEMM_HEAD_FWD, 2.0 * B * A * R, ((double)B * R + (double)A * R) * 4);
        NVQA_TRY((gemm_med<A_KC, B_KC>(c, mkargs(c->qd, R, c->P + c->lo.w_o, R, B, A, R),
                                       EpiBias2{c->scores, A, c->P + c->lo.b_o, nullptr})));
    }
    {
        ProfScope ps(c, PF_SOFTMAX_CE, 0, 2.0 * B * A * 4);
        hipLaunchKernelGGL(k_softmax_ce, dim3((B + 3) / 4), dim3(256), 0, c->s, c->scores,
                           train ? c->lab : (const int32_t *)nullptr, B, A, train ? c->dscores : (float *)nullptr,
                           c->rowloss, want_argmax ? c->argmax : (int32_t *)nullptr, train ? c->h_rowloss : (float *)nullptr);
        if (train) c->loss_rows = B;
    }
    NVQA_HIP(hipGetLastError());
    return 0;
}

static int arch2_backward(nvqa_ctx *c, const Drop &dr)
{
    const bool ride = ride_begin(c);
    const nvqa_dims &d = c->d;
    const int B = d.B, R = d.R, L = d.L, E = d.E, I = d.I, A = d.A, V = d.V, TS = c->TS, TB = TS * B;
    float *G = c->G;
    // only the top layer's h at step tmax receives a gradient from the head (Encoder_lstm.lua:238-239)
    // (dCT and dHT are one allocation.  The head product below rewrites dHT[L-1] in full and the persistent BPTT kernel only READS
    // the two buffers, so on that route they stay as they were zeroed; the per-level finisher carries dc in dCT in place.)
    {
        int rbx = 0;
        if (!(c->dct_zero && persist_bwd_rows(c, &rbx) != 0)) {
            NVQA_HIP(hipMemsetAsync(c->dCT, 0, (size_t)2 * L * B * R * 4, c->s));
            c->dct_zero = true;
        }
    }
    {
        ProfScope ps(c, PF_GEMM_HEAD_BWD, 4.0 * B * A * R, (2.0 * A * R + 2.0 * B * A) * 4);
        if (ride) ride_add(c, mkargs(c->dscores, A, c->qd, R, A, R, B), EpiStore{G + c->lo.w_o, R, 0});
        else NVQA_TRY((gemm_med<A_MC, B_NC>(c, mkargs(c->dscores, A, c->qd, R, A, R, B), EpiStore{G + c->lo.w_o, R, 0})));
        NVQA_TRY((gemm_med<A_KC, B_NC>(c, mkargs(c->dscores, A, c->P + c->lo.w_o, R, B, R, A),
                                       EpiHead2{c->dHT + (size_t)(L - 1) * B * R, R, dr})));
    }
    if (c->quirks & NVQA_QUIRK_H0) { // the aliased h0 tensor now holds THIS step's gradient: the step-1 dW_h2h sees it
        NVQA_HIP(hipMemcpyAsync(c->Hs[L - 1], c->dHT + (size_t)(L - 1) * B * R, (size_t)B * R * 4, hipMemcpyDeviceToDevice, c->s));
        NVQA_TRY(persist_reimage_h0_top(c)); // ... and so does the bf16 image the weight-gradient kernel stages from
    }
    {   // b_o: one B-row column sum (the one-stage form of arch1's head; under the BPTT when the jobs ride)
        ProfScope ps(c, PF_COLSUM, 0, (double)B * A * 4);
        ColsumBatch cb = {};
        cb.X[0] = c->dscores; cb.out[0] = G + c->lo.b_o; cb.M[0] = B; cb.N[0] = A; cb.ld[0] = A;
        cb.first_block[0] = 0;
        cb.first_block[1] = cb.first_block[2] = cb.first_block[3] = cb.first_block[4] = (A + 63) / 64;
        if (ride) {
            c->ride.cs = cb;
            c->ride.has_colsum = 1;
            c->ride_gemm_pending = true;
        } else {
            hipLaunchKernelGGL(k_colsum_batch, dim3(cb.first_block[4]), dim3(256), 0, c->s, cb);
            NVQA_HIP(hipGetLastError());
        }
    }
    if (c->dp_overlap_bptt) NVQA_TRY(reduce_segment(c, 2)); // classifier
    NVQA_TRY(lstm_backward(c, dr));
    NVQA_TRY(ride_flush(c));
    if (!c->dp_overlap_bptt) NVQA_TRY(reduce_segment(c, 2));
    NVQA_TRY(lstm_dx0(c, c->dX0));
    {   // cnn_projection:backward (002_train_baseline.lua:322): dW_p = dx_1^T fv_im, db_p = colsum(dx_1)
        ProfScope ps(c, PF_GEMM_HEAD_BWD, 2.0 * B * E * I, ((double)B * (E + I) + (double)E * I) * 4);
        NVQA_TRY((gemm_med<A_MC, B_NC>(c, mkargs(c->dX0, E, c->img, I, E, I, B), EpiStore{G + c->lo.w_p, I, 0})));
    }
    NVQA_TRY(colsum(c, c->dX0, B, E, E, G + c->lo.b_p, nullptr));
    NVQA_TRY(reduce_segment(c, 0)); // cnn projection
    if (c->quirks & NVQA_QUIRK_LOOKUP) { // the reference's flat gradient never receives the lookup gradient: zeros, nothing to exchange
        NVQA_HIP(hipMemsetAsync(G + c->lo.w_lk, 0, (size_t)(V + 1) * E * 4, c->s));
    } else {   // LookupTable gradient, summed over all steps into the shared gradWeight (Encoder_lstm.lua:53-58,256)
        NVQA_TRY(emb_backward(c, V + 1, TB, TS, c->dX0, dr, G + c->lo.w_lk, 1));
    }
    if (!(c->quirks & NVQA_QUIRK_LOOKUP))
        NVQA_TRY(reduce_range(c, c->lo.w_lk, (size_t)(V + 1) * E, 1)); // lookup table: travels under the weight-gradient GEMMs
    for (int l = L - 1; l >= 0; --l) NVQA_TRY(lstm_wgrads(c, l));  // + each layer's slice of the encoder segment
    return 0;
}

// ------------------------------------------------------------------------------------
// the step
// ------------------------------------------------------------------------------------
static int batch_release(nvqa_ctx *c); // with upload_batch, below
static int run_step(nvqa_ctx *c, const nvqa_dropout *dropout, float *loss_out)
{
    const Drop dr = mkdrop(dropout, true);
    if (c->pf_spin_steps == 0) c->pf_spin = 0; // (NVQA_PF_SPIN_STEPS: back to the kernels' own limit)
    if (c->pf_spin_steps > 0) --c->pf_spin_steps;
    if (c->comm) { // this step's status word: the other one than the last step's; cleared by that step's k_rmsprop (else: here)
        c->dp_slot ^= 1;
        if (!c->dp_clean[c->dp_slot]) NVQA_HIP(hipMemsetAsync(c->dp_status + c->dp_slot, 0, 4, c->s));
        c->dp_clean[c->dp_slot] = false;
    }
    if (c->d.arch == NVQA_ARCH1) {
        NVQA_TRY(arch1_forward(c, dr, true, false));
        NVQA_TRY(arch1_backward(c, dr));
    } else {
        NVQA_TRY(arch2_forward(c, dr, true, false));
        NVQA_TRY(arch2_backward(c, dr));
    }
    NVQA_TRY(batch_release(c)); // (arch2's backward pass reads the image features once more: dW_p)
    NVQA_TRY(latch_flush(c));   // (an err latch no kernel of the step carried: before the status word is exchanged)
    NVQA_TRY(reduce_join(c));
    c->have_grads = true; // (the loss: k_softmax_ce wrote its rows into the pinned h_rowloss; loss_mean_host adds them when asked)
    if (loss_out) {
        NVQA_HIP(hipStreamSynchronize(c->s));
        *loss_out = loss_mean_host(c);
        NVQA_TRY(check_persist(c));
    }
    return 0;
}

static int upload_batch(nvqa_ctx *c, int n, const int32_t *tokens, const int32_t *lengths, const float *img,
                        const int32_t *labels)
{
    const nvqa_dims &d = c->d;
    const size_t B = d.B, T = d.T, I = d.I;
    if (n < 1 || n > d.B) { set_error("batch rows %d outside 1..%d", n, d.B); return -1; }
    if (!tokens || (d.arch == NVQA_ARCH1 && !lengths)) { set_error("NULL batch pointer"); return -1; }
    // validate on the host: a bad index would fault on the device
    std::vector<int32_t> tk(B * T), ln(B), lb(B, 1);
    std::vector<float> im;
    for (size_t b = 0; b < B; ++b) {
        const size_t sb = b < (size_t)n ? b : 0; // short eval batches are padded with row 0
        const int32_t l = lengths ? lengths[sb] : (int32_t)T;
        if (d.arch == NVQA_ARCH1 && (l < 1 || l > (int32_t)T)) { set_error("length[%zu]=%d outside 1..%zu", sb, l, T); return -1; }
        ln[b] = l;
        for (size_t t = 0; t < T; ++t) {
            const int32_t w = tokens[sb * T + t];
            const bool pad = d.arch == NVQA_ARCH1 ? t < T - (size_t)l : false;
            if (pad ? w != 0 : (d.arch == NVQA_ARCH1 ? (w < 1 || w > d.V) : (w < 0 || w > d.V))) {
                set_error("token[%zu][%zu]=%d invalid (V=%d, length=%d)", sb, t, w, d.V, l);
                return -1;
            }
            tk[b * T + t] = w;
        }
        if (labels) {
            if (labels[sb] < 1 || labels[sb] > d.A) { set_error("label[%zu]=%d outside 1..%d", sb, labels[sb], d.A); return -1; }
            lb[b] = labels[sb];
        }
    }
    c->batch_uniform = true;
    for (size_t b = 1; b < B; ++b) c->batch_uniform = c->batch_uniform && ln[b] == ln[0];
    {   // share of the B x T (row, step) slots this batch fills (arch2 runs every row to the longest question): the persistent BPTT
        // launch of a ragged arch1 batch is that much shorter, and so is what may ride in it (persist_bwd.hip)
        double sl = 0;
        for (size_t b = 0; b < B; ++b) sl += ln[b];
        c->batch_len_frac = d.arch == NVQA_ARCH1 ? (float)(sl / ((double)B * T)) : 1.f;
    }
    // ---- pinned staging -> the device set the running step does not read, on the side stream --------------------------------
    if (!c->hset[0].tok) { // first host-batch call: the second device set, two pinned staging sets, the events
        c->bset[0] = {c->tok, c->len, c->lab, c->img};
        NVQA_TRY(dalloc(&c->bset[1].tok, B * T));
        NVQA_TRY(dalloc(&c->bset[1].len, B));
        NVQA_TRY(dalloc(&c->bset[1].lab, B));
        NVQA_TRY(dalloc(&c->bset[1].img, B * I));
        for (int p = 0; p < 2; ++p) {
            NVQA_HIP(hipHostMalloc((void **)&c->hset[p].tok, B * T * 4, hipHostMallocDefault));
            NVQA_HIP(hipHostMalloc((void **)&c->hset[p].len, B * 4, hipHostMallocDefault));
            NVQA_HIP(hipHostMalloc((void **)&c->hset[p].lab, B * 4, hipHostMallocDefault));
            NVQA_HIP(hipHostMalloc((void **)&c->hset[p].img, B * I * 4, hipHostMallocDefault));
            NVQA_HIP(hipEventCreateWithFlags(&c->evCopied[p], hipEventDisableTiming));
            NVQA_HIP(hipEventCreateWithFlags(&c->evBatchFree[p], hipEventDisableTiming));
        }
    }
    const int p = c->bcur ^ 1;
    if (c->copied_rec[p]) NVQA_HIP(hipEventSynchronize(c->evCopied[p])); // the staging set's previous copy (two calls ago) is long done
    memcpy(c->hset[p].tok, tk.data(), B * T * 4);
    memcpy(c->hset[p].len, ln.data(), B * 4);
    memcpy(c->hset[p].lab, lb.data(), B * 4);
    if (img) {
        if ((size_t)n == B) memcpy(c->hset[p].img, img, B * I * 4);
        else for (size_t b = 0; b < B; ++b) memcpy(c->hset[p].img + b * I, img + (b < (size_t)n ? b : 0) * I, I * 4);
    }
    if (c->batch_free_rec[p]) NVQA_HIP(hipStreamWaitEvent(c->sx, c->evBatchFree[p], 0)); // the step that read device set p has run
    NVQA_HIP(hipMemcpyAsync(c->bset[p].tok, c->hset[p].tok, B * T * 4, hipMemcpyHostToDevice, c->sx));
    NVQA_HIP(hipMemcpyAsync(c->bset[p].len, c->hset[p].len, B * 4, hipMemcpyHostToDevice, c->sx));
    NVQA_HIP(hipMemcpyAsync(c->bset[p].lab, c->hset[p].lab, B * 4, hipMemcpyHostToDevice, c->sx));
    if (img) NVQA_HIP(hipMemcpyAsync(c->bset[p].img, c->hset[p].img, B * I * 4, hipMemcpyHostToDevice, c->sx));
    NVQA_HIP(hipEventRecord(c->evCopied[p], c->sx));
    c->copied_rec[p] = true;
    NVQA_HIP(hipStreamWaitEvent(c->s, c->evCopied[p], 0));
    c->bcur = p;
    c->tok = c->bset[p].tok; c->len = c->bset[p].len; c->lab = c->bset[p].lab; c->img = c->bset[p].img;
    return 0;
}

// behind the last kernel of a step that reads the batch buffers: device set bcur may be refilled
static int batch_release(nvqa_ctx *c)
{
    if (!c->hset[0].tok) return 0;
    NVQA_HIP(hipEventRecord(c->evBatchFree[c->bcur], c->s));
    c->batch_free_rec[c->bcur] = true;
    return 0;
}

extern "C" int nvqa_step(nvqa_ctx *c, const int32_t *tokens, const int32_t *lengths, const float *img,
                         const int32_t *labels, const nvqa_dropout *dropout, float *loss_out)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    if (!labels || !img) { set_error("labels / img is NULL"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_TRY(upload_batch(c, c->d.B, tokens, lengths, img, labels));
    return run_step(c, dropout, loss_out);
}

// BASELINE.json configs[4]: the VGG-16 fc7 extractor run on the fly in front of the arch1 step
// (001_prepro_img_vgg.lua:101-113 + 002_train_baseline.lua:117-121 + JdJ).  images [B x 3 x hw x hw] as loadim
// returns them; the extractor's feature width must equal dims.I.
struct nvqa_vgg;
int nvqa_vgg_forward_device(nvqa_vgg *v, const float *images, int n, const float **feats_dev, int *F, hipStream_t *stream);

extern "C" int nvqa_step_images(nvqa_ctx *c, nvqa_vgg *vgg, const float *images, const int32_t *tokens,
                                const int32_t *lengths, const int32_t *labels, const nvqa_dropout *dropout,
                                float *loss_out)
{
    if (!c || !vgg || !images || !labels) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    // tokens / lengths / labels go through the usual validated upload; the image slot is filled below
    NVQA_TRY(upload_batch(c, c->d.B, tokens, lengths, nullptr, labels));
    const float *feats = nullptr;
    int F = 0;
    hipStream_t vs = nullptr;
    NVQA_TRY(nvqa_vgg_forward_device(vgg, images, c->d.B, &feats, &F, &vs));
    if (F != c->d.I) { set_error("extractor feature width %d != dims.I %d", F, c->d.I); return -1; }
    // the step's stream waits for the extractor's stream, then normalises straight into the batch buffer
    NVQA_HIP(hipEventRecord(c->evStart, vs));
    NVQA_HIP(hipStreamWaitEvent(c->s, c->evStart, 0));
    hipLaunchKernelGGL(k_l2norm_copy, dim3((c->d.B + 3) / 4), dim3(256), 0, c->s, feats, c->d.B, c->d.I, c->img);
    NVQA_HIP(hipGetLastError());
    return run_step(c, dropout, loss_out);
}

extern "C" int nvqa_forward(nvqa_ctx *c, int32_t n, const int32_t *tokens, const int32_t *lengths,
                            const float *img, float *scores_out, int32_t *argmax_out)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_TRY(upload_batch(c, n, tokens, lengths, img, nullptr));
    const Drop dr = mkdrop(nullptr, false);
    if (c->d.arch == NVQA_ARCH1) {
        NVQA_TRY(arch1_forward(c, dr, false, true));
    } else {
        NVQA_TRY(arch2_forward(c, dr, false, true));
    }
    NVQA_HIP(hipStreamSynchronize(c->s));
    NVQA_TRY(check_persist(c));
    if (scores_out) NVQA_HIP(hipMemcpy(scores_out, c->scores, (size_t)n * c->d.A * 4, hipMemcpyDeviceToHost));
    if (argmax_out) NVQA_HIP(hipMemcpy(argmax_out, c->argmax, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nvqa_evaluate(nvqa_ctx *c, int32_t n, const int32_t *tokens, const int32_t *lengths, const float *img,
                             const int32_t *labels, const int32_t *mc_ans, int32_t n_mc, float *scores_out,
                             int32_t *argmax_out, int32_t *mc_argmax_out, float *loss_out)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    if (mc_ans && (n_mc < 1 || n_mc > 64)) { set_error("n_mc=%d outside 1..64", n_mc); return -1; }
    if ((loss_out && !labels) || (mc_argmax_out && !mc_ans)) { set_error("loss_out needs labels, mc_argmax_out needs mc_ans"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_TRY(upload_batch(c, n, tokens, lengths, img, labels)); // validates tokens, lengths, labels
    const nvqa_dims &d = c->d;
    if (mc_ans) {
        for (size_t i = 0; i < (size_t)n * n_mc; ++i)
            if (mc_ans[i] < 0 || mc_ans[i] > d.A) { set_error("mc_ans[%zu]=%d outside 0..%d", i, mc_ans[i], d.A); return -1; }
        if (!c->mc) NVQA_TRY(dalloc(&c->mc, (size_t)d.B * 64));
        NVQA_HIP(hipMemcpy(c->mc, mc_ans, (size_t)n * n_mc * 4, hipMemcpyHostToDevice));
    }
    const Drop dr = mkdrop(nullptr, false);
    if (d.arch == NVQA_ARCH1) NVQA_TRY(arch1_forward(c, dr, false, true));
    else NVQA_TRY(arch2_forward(c, dr, false, true));
    if (labels) { // mean cross-entropy of the first n rows (short batches are padded with copies of row 0)
        hipLaunchKernelGGL(k_softmax_ce, dim3((d.B + 3) / 4), dim3(256), 0, c->s, c->scores, c->lab, d.B, d.A, (float *)nullptr,
                           c->rowloss, (int32_t *)nullptr, c->h_rowloss);
        c->loss_rows = n;
    }
    if (mc_ans) // reuses the dscores buffer's first n ints for the answers (training overwrites it every step)
        hipLaunchKernelGGL(k_mc_argmax, dim3((n + 3) / 4), dim3(256), 0, c->s, c->scores, c->mc, n, d.A, n_mc,
                           reinterpret_cast<int32_t *>(c->dscores));
    NVQA_HIP(hipGetLastError());
    NVQA_HIP(hipStreamSynchronize(c->s));
    NVQA_TRY(check_persist(c));
    if (scores_out) NVQA_HIP(hipMemcpy(scores_out, c->scores, (size_t)n * d.A * 4, hipMemcpyDeviceToHost));
    if (argmax_out) NVQA_HIP(hipMemcpy(argmax_out, c->argmax, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (mc_argmax_out) NVQA_HIP(hipMemcpy(mc_argmax_out, c->dscores, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (loss_out) *loss_out = loss_mean_host(c);
    return 0;
}

// model variants of the reference's other training scripts
extern "C" int nvqa_set_fusion(nvqa_ctx *c, int askip)
{
    if (!c || c->d.arch != NVQA_ARCH1 || askip < 0 || askip > 2) { set_error("nvqa_set_fusion: arch1 context and mode 0 (AxB) / 1 (AskipB) / 2 (A_B) expected"); return -1; }
    if ((askip == 2) != (c->fusion_askip == 2)) {
        // netdef.A_B changes the classifier to Linear(2C, A): another parameter vector.  Only on a context that holds no
        // parameters, gradients or optimiser state yet (the buffers are sized for either layout at nvqa_create).
        if (!c->pristine) { set_error("nvqa_set_fusion: switching to / from netdef.A_B changes the parameter layout; call it right after nvqa_create"); return -1; }
        if (nvqa_layout_init_fusion(&c->d, askip, &c->lo)) { set_error("layout"); return -1; }
    }
    c->fusion_askip = askip;
    return 0;
}
extern "C" int nvqa_set_ref_quirks(nvqa_ctx *c, int flags)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    if (c->d.arch != NVQA_ARCH2 || (flags & ~(NVQA_QUIRK_H0 | NVQA_QUIRK_LOOKUP))) {
        set_error("nvqa_set_ref_quirks: arch2 context and flags within %d expected", NVQA_QUIRK_H0 | NVQA_QUIRK_LOOKUP);
        return -1;
    }
    NVQA_HIP(hipSetDevice(c->device));
    const size_t BR = (size_t)c->d.B * c->d.R;
    NVQA_HIP(hipMemsetAsync(c->dHT, 0, (size_t)c->d.L * BR * 4, c->s));  // the carried h0 state
    NVQA_HIP(hipMemsetAsync(c->Hs[c->d.L - 1], 0, BR * 4, c->s));        // and the step-0 rows it may have been copied into
    c->quirks = flags;
    c->h0_img_clean = false;
    return 0;
}
extern "C" int nvqa_set_precision(nvqa_ctx *c, int bf16)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    if (bf16 != 0 && bf16 != 1) { set_error("precision must be 0 (f32) or 1 (bf16 operands)"); return -1; }
    c->bf16 = bf16 != 0;
    return 0;
}

extern "C" int nvqa_set_grad_scales(nvqa_ctx *c, const float scales[3])
{
    if (!c || !scales) { set_error("NULL argument"); return -1; }
    for (int i = 0; i < 3; ++i) c->gscale[i] = scales[i];
    return 0;
}

extern "C" int nvqa_get_loss(nvqa_ctx *c, float *loss_out)
{
    if (!c || !loss_out) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_HIP(hipStreamSynchronize(c->s));
    *loss_out = loss_mean_host(c);
    return check_persist(c);
}

extern "C" int nvqa_rmsprop_update(nvqa_ctx *c, float lr, float alpha, float eps, float wd, float clamp)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    if (!c->have_grads) { set_error("nvqa_rmsprop_update before any nvqa_step"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    ProfScope ps(c, PF_RMSPROP, 0, 20.0 * c->lo.total);
    const float inv_world = 1.0f / (float)c->world;
    if (c->gscale[0] == c->gscale[1] && c->gscale[1] == c->gscale[2]) {
        const size_t n4 = c->lo.total / 4; // every tensor size is a multiple of 4 (check_dims)
        hipLaunchKernelGGL(k_rmsprop, dim3(2048), dim3(256), 0, c->s, reinterpret_cast<float4 *>(c->P),
                           reinterpret_cast<const float4 *>(c->G), reinterpret_cast<float4 *>(c->M2), n4, lr, alpha,
                           eps, wd, clamp, inv_world * c->gscale[0], c->comm ? (const unsigned *)nullptr : c->pf_sticky,
                           c->comm ? c->dp_status + c->dp_slot : (const float *)nullptr, c->comm ? c->h_dp_status : (float *)nullptr,
                           c->comm ? c->dp_status + (c->dp_slot ^ 1) : (float *)nullptr);
    } else { // -lr_scale of 003_train_ae_based_wp.lua:344: encoder / embedding gradients scaled before the clamp
        size_t off = 0;
        for (int sgm = 0; sgm < 3; ++sgm) {
            const size_t n4 = c->lo.seg[sgm] / 4;
            hipLaunchKernelGGL(k_rmsprop, dim3(1024), dim3(256), 0, c->s, reinterpret_cast<float4 *>(c->P + off),
                               reinterpret_cast<const float4 *>(c->G + off), reinterpret_cast<float4 *>(c->M2 + off), n4,
                               lr, alpha, eps, wd, clamp, inv_world * c->gscale[sgm], c->comm ? (const unsigned *)nullptr : c->pf_sticky,
                               c->comm ? c->dp_status + c->dp_slot : (const float *)nullptr, c->comm ? c->h_dp_status : (float *)nullptr,
                           c->comm ? c->dp_status + (c->dp_slot ^ 1) : (float *)nullptr);
            off += c->lo.seg[sgm];
        }
    }
    NVQA_HIP(hipGetLastError());
    if (c->comm) c->dp_clean[c->dp_slot ^ 1] = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// HBM-resident dataset + device-side next_batch
// ------------------------------------------------------------------------------------
extern "C" int nvqa_dataset_load(nvqa_ctx *c, int64_t n_q, const int32_t *questions, const int32_t *lengths,
                                 const int32_t *img_pos, const int32_t *answers, int64_t n_img,
                                 const float *feats, int l2_normalize)
{
    if (!c || !questions || !img_pos || !answers || !feats) { set_error("NULL argument"); return -1; }
    if (n_q < 1 || n_img < 1) { set_error("empty dataset"); return -1; }
    const nvqa_dims &d = c->d;
    if (d.arch == NVQA_ARCH1 && !lengths) { set_error("arch1 needs lengths"); return -1; }
    for (int64_t q = 0; q < n_q; ++q) {
        if (img_pos[q] < 1 || img_pos[q] > n_img) { set_error("img_pos[%lld]=%d outside 1..%lld", (long long)q, img_pos[q], (long long)n_img); return -1; }
        if (answers[q] < 1 || answers[q] > d.A) { set_error("answers[%lld]=%d outside 1..%d", (long long)q, answers[q], d.A); return -1; }
        const int l = lengths ? lengths[q] : d.T;
        if (d.arch == NVQA_ARCH1 && (l < 1 || l > d.T)) { set_error("lengths[%lld]=%d outside 1..%d", (long long)q, l, d.T); return -1; }
        for (int t = 0; t < d.T; ++t) {
            const int w = questions[q * d.T + t];
            const bool pad = d.arch == NVQA_ARCH1 && t < d.T - l;
            if (pad ? w != 0 : (d.arch == NVQA_ARCH1 ? (w < 1 || w > d.V) : (w < 0 || w > d.V))) {
                set_error("questions[%lld][%d]=%d invalid", (long long)q, t, w);
                return -1;
            }
        }
    }
    NVQA_HIP(hipSetDevice(c->device));
    NVQA_HIP(hipStreamSynchronize(c->s));
    Dataset &ds = c->ds;
    for (void *p : {(void *)ds.Q, (void *)ds.QL, (void *)ds.IP, (void *)ds.ANS, (void *)ds.F})
        if (p) (void)hipFree(p);
    ds = Dataset();
    NVQA_TRY(dalloc(&ds.Q, (size_t)n_q * d.T));
    NVQA_TRY(dalloc(&ds.IP, (size_t)n_q));
    NVQA_TRY(dalloc(&ds.ANS, (size_t)n_q));
    NVQA_TRY(dalloc(&ds.F, (size_t)n_img * d.I));
    NVQA_HIP(hipMemcpy(ds.Q, questions, (size_t)n_q * d.T * 4, hipMemcpyHostToDevice));
    NVQA_HIP(hipMemcpy(ds.IP, img_pos, (size_t)n_q * 4, hipMemcpyHostToDevice));
    NVQA_HIP(hipMemcpy(ds.ANS, answers, (size_t)n_q * 4, hipMemcpyHostToDevice));
    NVQA_HIP(hipMemcpy(ds.F, feats, (size_t)n_img * d.I * 4, hipMemcpyHostToDevice));
    if (lengths) {
        NVQA_TRY(dalloc(&ds.QL, (size_t)n_q));
        NVQA_HIP(hipMemcpy(ds.QL, lengths, (size_t)n_q * 4, hipMemcpyHostToDevice));
    }
    ds.n_q = n_q;
    ds.n_img = n_img;
    ds.uniform_len = true; // all questions of one length: every batch drawn from the dataset is full-length
    ds.mean_len_frac = 1.f;
    if (lengths) {
        for (int64_t q = 1; q < n_q; ++q) ds.uniform_len = ds.uniform_len && lengths[q] == lengths[0];
        double sl = 0;
        for (int64_t q = 0; q < n_q; ++q) sl += lengths[q];
        ds.mean_len_frac = (float)(sl / ((double)n_q * d.T)); // what a drawn batch's rows are expected to fill of the T steps
    }
    if (l2_normalize > 1 && (l2_normalize >= d.I || l2_normalize % 4)) { set_error("l2_normalize split %d must be a multiple of 4 below I=%d", l2_normalize, d.I); return -1; }
    if (l2_normalize > 1) {
        hipLaunchKernelGGL(k_l2norm_rows, dim3((unsigned)((n_img + 3) / 4)), dim3(256), 0, c->s, ds.F, n_img, d.I, 0, l2_normalize);
        hipLaunchKernelGGL(k_l2norm_rows, dim3((unsigned)((n_img + 3) / 4)), dim3(256), 0, c->s, ds.F, n_img, d.I, l2_normalize, d.I - l2_normalize);
        NVQA_HIP(hipGetLastError());
        NVQA_HIP(hipStreamSynchronize(c->s));
    } else if (l2_normalize) {
        hipLaunchKernelGGL(k_l2norm_rows, dim3((unsigned)((n_img + 3) / 4)), dim3(256), 0, c->s, ds.F, n_img, d.I, 0, d.I);
        NVQA_HIP(hipGetLastError());
        NVQA_HIP(hipStreamSynchronize(c->s));
    }
    return 0;
}

extern "C" int nvqa_step_indices(nvqa_ctx *c, const int64_t *qinds, const nvqa_dropout *dropout, float *loss_out)
{
    if (!c || !qinds) { set_error("NULL argument"); return -1; }
    if (c->ds.n_q == 0) { set_error("nvqa_step_indices before nvqa_dataset_load"); return -1; }
    const nvqa_dims &d = c->d;
    for (int b = 0; b < d.B; ++b)
        if (qinds[b] < 0 || qinds[b] >= c->ds.n_q) { set_error("qinds[%d]=%lld outside 0..%lld", b, (long long)qinds[b], (long long)c->ds.n_q - 1); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    c->batch_uniform = c->ds.uniform_len;
    c->batch_len_frac = d.arch == NVQA_ARCH1 ? c->ds.mean_len_frac : 1.f;
    {
        ProfScope ps(c, PF_GATHER, 0, 2.0 * d.B * d.I * 4);
        if (d.B <= NVQA_QARG_MAX && c->ds.n_q <= 0x7fffffffLL) { // the ids as kernel arguments (kernels.h: no H2D blit in front of the step)
            for (int b = 0; b < d.B; ++b) c->qarg.q[b] = (int32_t)qinds[b]; // (the launch copies it into the kernarg segment)
            hipLaunchKernelGGL(k_gather_batch<true>, dim3(d.B), dim3(256), 0, c->s, c->qarg, (const int64_t *)nullptr, c->ds.Q, c->ds.QL, c->ds.IP,
                               c->ds.ANS, c->ds.F, d.T, d.I, c->tok, c->len, c->lab, c->img);
        } else {
            NVQA_HIP(hipMemcpyAsync(c->qinds, qinds, (size_t)d.B * 8, hipMemcpyHostToDevice, c->s));
            hipLaunchKernelGGL(k_gather_batch<false>, dim3(d.B), dim3(256), 0, c->s, QIdxArg{}, c->qinds, c->ds.Q, c->ds.QL, c->ds.IP,
                               c->ds.ANS, c->ds.F, d.T, d.I, c->tok, c->len, c->lab, c->img);
        }
    }
    NVQA_HIP(hipGetLastError());
    return run_step(c, dropout, loss_out);
}

// ------------------------------------------------------------------------------------
// data parallel: RCCL (librccl.so resolved lazily -- single-GPU users never load it)
// ------------------------------------------------------------------------------------
namespace {
struct Id128 { char b[128]; }; // ncclUniqueId, passed by value
struct Rccl {
    std::string path;     // what was asked for: NVQA_RCCL_LIB, or "" = the default rule of load_rccl
    std::string resolved; // the file the entry points live in
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
// One table per library path.  NVQA_RCCL_LIB names the collective library to load instead of librccl.so (read at
// every nvqa_comm_* call, so a process can hold contexts on different libraries): any .so that exports the five nccl*
// symbols below.  tests/shim/nccl_shim.hip is one whose all-reduce returns world x send, i.e. what `world` ranks with
// identical gradients produce, so the whole exchange path can be checked on one GPU (tests/test_gpu_dp_shim.py).
//
// WHICH librccl.  A process holds ONE HIP runtime image -- libamdhip64.so.7, whichever copy was
// mapped first: /opt/rocm's when the host loads libnvqa first, the torch wheel's bundled one when the host imported torch
// first -- and the streams and buffers this library hands to ncclAllReduce belong to that image.  The collective library
// must be the one built against it, so the rule is: the librccl that sits NEXT TO the mapped libamdhip64 (absolute path,
// found with dladdr), then the ROCm tree this library was linked against, then the loader's search path.  Always
// RTLD_LOCAL: round 3 opened it RTLD_GLOBAL, and a second librccl image mapped later (import torch brings its own copy)
// then interposed its global C++ objects on the first one's -- both ran their destructors at exit ("double free or
// corruption").  A bare-soname dlopen comes LAST because it silently returns whatever image with that soname is already
// mapped.  The resolved path is kept (nvqa_comm_library) and printed once per process by nvqa_comm_init.
std::vector<Rccl *> g_rccl_libs;
std::string dir_of_mapped_hip()
{
    Dl_info di;
    if (!dladdr((const void *)&hipGetDeviceCount, &di) || !di.dli_fname) return "";
    char real[PATH_MAX];
    const std::string f = realpath(di.dli_fname, real) ? real : di.dli_fname;
    const size_t k = f.rfind('/');
    return k == std::string::npos ? "" : f.substr(0, k);
}
std::string path_of_handle(void *h, const char *sym)
{
    Dl_info di;
    void *p = dlsym(h, sym);
    if (p && dladdr(p, &di) && di.dli_fname) {
        char real[PATH_MAX];
        return realpath(di.dli_fname, real) ? real : di.dli_fname;
    }
    return "?";
}
const Rccl *load_rccl()
{
    const char *env = getenv("NVQA_RCCL_LIB");
    const std::string want = env && env[0] ? env : "";
    for (const Rccl *r : g_rccl_libs)
        if (r->path == want) return r;
    void *h = nullptr;
    if (!want.empty()) {
        h = dlopen(want.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { set_error("cannot load NVQA_RCCL_LIB=%s: %s", want.c_str(), dlerror()); return nullptr; }
    } else {
        std::vector<std::string> cand;
        const std::string hipdir = dir_of_mapped_hip();
        if (!hipdir.empty()) { cand.push_back(hipdir + "/librccl.so.1"); cand.push_back(hipdir + "/librccl.so"); }
        const char *rp = getenv("ROCM_PATH");
        cand.push_back(std::string(rp && rp[0] ? rp : "/opt/rocm") + "/lib/librccl.so.1");
        cand.push_back("librccl.so.1");
        cand.push_back("librccl.so");
        std::string tried;
        for (const std::string &p : cand) {
            if (p[0] == '/' && access(p.c_str(), R_OK) != 0) continue;
            h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
            tried += (tried.empty() ? "" : "; ") + p + ": " + dlerror();
        }
        if (!h) { set_error("cannot load librccl (%s)", tried.c_str()); return nullptr; }
    }
    Rccl *r = new Rccl();
    r->path = want;
    r->GetUniqueId = (decltype(r->GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r->CommInitRank = (decltype(r->CommInitRank))dlsym(h, "ncclCommInitRank");
    r->AllReduce = (decltype(r->AllReduce))dlsym(h, "ncclAllReduce");
    r->CommDestroy = (decltype(r->CommDestroy))dlsym(h, "ncclCommDestroy");
    r->GetErrorString = (decltype(r->GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r->GetUniqueId || !r->CommInitRank || !r->AllReduce) {
        set_error("%s lacks the nccl symbols", want.empty() ? "librccl.so" : want.c_str());
        delete r;
        return nullptr;
    }
    r->h = h;
    r->resolved = path_of_handle(h, "ncclAllReduce");
    g_rccl_libs.push_back(r);
    return r;
}
inline const Rccl *rccl_of(const nvqa_ctx *c) { return static_cast<const Rccl *>(c->rccl); }
} // namespace

extern "C" int nvqa_comm_unique_id(void *id_out)
{
    if (!id_out) { set_error("NULL argument"); return -1; }
    const Rccl *r = load_rccl();
    if (!r) return -1;
    const int rc = r->GetUniqueId(id_out);
    if (rc) { set_error("ncclGetUniqueId: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); return -1; }
    return 0;
}

extern "C" const char *nvqa_comm_library(void)
{
    const Rccl *r = load_rccl();
    return r ? r->resolved.c_str() : nullptr;
}

extern "C" int nvqa_comm_init(nvqa_ctx *c, int rank, int world, const void *id)
{
    if (!c || !id) { set_error("NULL argument"); return -1; }
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad rank/world %d/%d", rank, world); return -1; }
    if (c->comm) { set_error("nvqa_comm_init: the context already has a communicator"); return -1; }
    NVQA_HIP(hipSetDevice(c->device));
    {   // When is the multimodal segment exchanged?
        // Default (round 4): BEHIND the persistent BPTT launch, with everything else.  No collective kernel ever runs beside a persistent
        // launch then -- the forward launch never had one beside it (reduce_join ends the step) -- so there is no co-residency to
        // arrange: RCCL keeps its own channel count (the cap below costs it bandwidth), the BPTT grid needs no reserve, and the head's
        // weight gradients ride in its idle workgroups as in single-GPU runs.  The 55 MB then travel under the 1.0 ms of d(input),
        // embedding gradient and weight-gradient GEMMs that follow the BPTT -- ordinary kernels that share CUs with a collective.
        // NVQA_DP_OVERLAP_BPTT=1 (round 3's order): the segment travels UNDER the BPTT launch, which needs all its workgroups
        // resident at once (240 of the 256 CUs in f32) while a collective kernel holds one CU per channel: the library leaves the
        // collective NVQA_COMM_CUS compute units (default 16), caps RCCL's channels at that number BEFORE the communicator is created
        // unless the caller has set them, and persist_bwd_rows() refuses the persistent path when its grid would not leave them free.
        const char *eo = getenv("NVQA_DP_OVERLAP_BPTT");
        c->dp_overlap_bptt = eo && eo[0] == '1';
        c->comm_cus = 0;
        const char *e = getenv("NVQA_COMM_CUS");
        if (c->dp_overlap_bptt) c->comm_cus = e ? atoi(e) : 16;
        if (c->comm_cus > 0) {
            char buf[16];
            snprintf(buf, sizeof(buf), "%d", c->comm_cus);
            setenv("NCCL_MAX_NCHANNELS", buf, 0);
            // What RCCL will actually read: a value the caller exported earlier wins over ours (setenv(..., 0)), and more
            // channels mean more CUs held by the collective's kernel.  Reserve what is really configured, so that
            // persist_bwd_rows() refuses the persistent BPTT launch instead of letting its spins expire.  (If RCCL read its
            // environment earlier in this process -- another communicator -- the cap has no effect on it at all: stated in
            // INTEGRATION.md; the variable is process-wide and caps every other communicator created after this call.)
            const char *eff = getenv("NCCL_MAX_NCHANNELS");
            const int nch = eff ? atoi(eff) : 0;
            if (nch > c->comm_cus) {
                static bool said = false;
                if (!said) {
                    fprintf(stderr, "[nvqa] NCCL_MAX_NCHANNELS=%d exceeds NVQA_COMM_CUS=%d: reserving %d compute units for the collective\n",
                            nch, c->comm_cus, nch);
                    said = true;
                }
                c->comm_cus = nch;
            }
        }
    }
    const Rccl *r = load_rccl();
    if (!r) return -1;
    {
        static bool said = false; // once per process: which collective library the data-parallel path runs on
        if (!said && !(getenv("NVQA_QUIET") && atoi(getenv("NVQA_QUIET")))) {
            fprintf(stderr, "[nvqa] rank %d/%d: collective library %s\n", rank, world, r->resolved.c_str());
            said = true;
        }
    }
    Id128 idv;
    memcpy(idv.b, id, 128);
    void *comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, idv, rank);
    if (rc) { set_error("ncclCommInitRank: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); return -1; }
    c->comm = comm;
    c->rccl = r;
    c->rank = rank;
    c->world = world;
    return 0;
}

// Gradient exchange: one ncclAllReduce(sum) per parameter segment, issued on the communication
// stream as soon as the segment's gradients are final (backward-completion order: multimodal,
// encoder, embedding), so the largest bucket travels over xGMI while BPTT is still running.
// The 1/world scale and the clamp stay in k_rmsprop: the clamp must act on the mean
// (002_train_baseline.lua:329 is non-linear).  Every rank issues the same calls in the same order.
static int reduce_range(nvqa_ctx *c, size_t off, size_t count, int ev)
{
    if (!c->comm || count == 0) return 0;
    NVQA_HIP(hipEventRecord(c->evSeg[ev], c->s));
    NVQA_HIP(hipStreamWaitEvent(c->sc, c->evSeg[ev], 0));
    ProfScope ps(c, PF_ALLREDUCE, 0, 4.0 * count, c->sc);
    const Rccl *r = rccl_of(c);
    const int rc = r->AllReduce(c->G + off, c->G + off, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, c->sc);
    if (rc) { set_error("ncclAllReduce: %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); return -1; }
    return 0;
}
static int reduce_segment(nvqa_ctx *c, int seg)
{
    size_t off = 0;
    for (int i = 0; i < seg; ++i) off += c->lo.seg[i];
    return reduce_range(c, off, c->lo.seg[seg], seg);
}
static void comm_destroy(nvqa_ctx *c)
{
    if (c->comm && rccl_of(c) && rccl_of(c)->CommDestroy) rccl_of(c)->CommDestroy(c->comm);
    c->comm = nullptr;
    c->rccl = nullptr;
}
static int reduce_join(nvqa_ctx *c)
{
    if (!c->comm) return 0;
    {   // did any rank's persistent kernel give up in this step?  dp_status[0] was set by the err latches behind the launches
        // (persist_fwd.hip); its sum over the ranks makes every rank's k_rmsprop skip the update together.
        NVQA_HIP(hipEventRecord(c->evSeg[0], c->s));
        NVQA_HIP(hipStreamWaitEvent(c->sc, c->evSeg[0], 0));
        const Rccl *r = rccl_of(c);
        const int rc = r->AllReduce(c->dp_status + c->dp_slot, c->dp_status + c->dp_slot, 1, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, c->sc);
        if (rc) { set_error("ncclAllReduce (status): %s", r->GetErrorString ? r->GetErrorString(rc) : "?"); return -1; }
        NVQA_HIP(hipMemcpyAsync(c->h_dp_status, c->dp_status + c->dp_slot, 4, hipMemcpyDeviceToHost, c->sc)); // this step's word only: word 1 of the host copy is k_rmsprop's sticky count
    }
    NVQA_HIP(hipEventRecord(c->evComm, c->sc));
    NVQA_HIP(hipStreamWaitEvent(c->s, c->evComm, 0));
    return 0;
}

// ------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------
extern "C" int nvqa_profile_enable(nvqa_ctx *c, int enable)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    NVQA_HIP(hipStreamSynchronize(c->s));
    prof_collect(c);
    c->prof_on = enable != 0;
    return 0;
}
extern "C" int nvqa_profile_reset(nvqa_ctx *c)
{
    if (!c) { set_error("ctx is NULL"); return -1; }
    NVQA_HIP(hipStreamSynchronize(c->s));
    prof_collect(c);
    for (int i = 0; i < PF_COUNT; ++i) { c->prof[i].ms = c->prof[i].flops = c->prof[i].bytes = 0; c->prof[i].launches = 0; }
    return 0;
}
extern "C" int nvqa_profile_count(const nvqa_ctx *) { return PF_COUNT; }
extern "C" const char *nvqa_profile_name(const nvqa_ctx *, int idx) { return idx >= 0 && idx < PF_COUNT ? kProfNames[idx] : ""; }
extern "C" int nvqa_profile_get(nvqa_ctx *c, int idx, double *total_ms, int64_t *launches, double *flops, double *bytes)
{
    if (!c || idx < 0 || idx >= PF_COUNT) { set_error("bad profile index"); return -1; }
    NVQA_HIP(hipStreamSynchronize(c->s));
    prof_collect(c);
    if (total_ms) *total_ms = c->prof[idx].ms;
    if (launches) *launches = c->prof[idx].launches;
    if (flops) *flops = c->prof[idx].flops;
    if (bytes) *bytes = c->prof[idx].bytes;
    return 0;
}
