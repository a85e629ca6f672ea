// persist_bwd.hip -- launcher of the persistent BPTT kernel (lstm_persist_bwd2.h: two independent row chains per workgroup,
// f32 and bf16 instances), its own translation unit.  Round 2's one-chain kernel (lstm_persist_bwd.h) is gone: the two-chain
// form is faster in both precisions (f32 1.10 -> 0.91 ms, bf16 0.55 -> 0.45 ms) and one BPTT route beside the per-level
// fallback is enough.
#include <stdlib.h>
#include <algorithm>
#include <string.h>

#include "lstm_persist_bwd2.h"
#include "lstm_persist_bwd3.h"
#include "persist_host.h"

namespace nvqa {

// column tiles (of 16 units) per workgroup: the bf16 weights take half the registers
static int persist_bwd_ntn(const nvqa_ctx *c) { return c->bf16 && c->d.L > 1 ? 4 : 2; }

size_t persist_bwd_counter_words(const nvqa_dims &d, int TS)
{
    // finest row blocking: 64 rows; up to four chains (lstm_persist_bwd3.h); REC counters + UP flags per unit tile of 32 units; + the err record
    const size_t rbmax = ((size_t)d.B + 63) / 64;
    return ((size_t)d.L * rbmax * 4 * TS * (1 + (size_t)d.R / 32) + 4 + 3) / 4 * 4;
}

int persist_bwd_rows(const nvqa_ctx *c, int *RB)
{
    const nvqa_dims &d = c->d;
    // NVQA_PERSIST_BWD: 1 on, 0 off (the per-level kernels); unset: on
    const bool on = c->persist_bwd_on < 0 ? true : c->persist_bwd_on > 0;
    if (!persist_rows(c) || !on || d.L > 2 || !c->pb_cnt) return 0;
    const int mtiles = (d.B + 15) / 16, NU = d.R / (16 * persist_bwd_ntn(c)), MT = d.L == 1 || c->bf16 ? 4 : 7;
    *RB = (mtiles + MT - 1) / MT;
    if ((2 * d.L - 1) * *RB * NU > c->num_cus || c->num_cus < 256 || (2 * d.L - 1) * *RB > 8 * (32 / NU)) return 0;
    // data parallel: an all-reduce is in flight during BPTT; its kernel keeps the CUs nvqa_comm_init left it
    if (c->comm && c->dp_overlap_bptt && (2 * d.L - 1) * *RB * NU + c->comm_cus > c->num_cus) return 0;
    return MT;
}

template <class K> static int check_resident(nvqa_ctx *c, K kernel, size_t lds, int grid, int *resident)
{
    if (*resident < 0) {
        // (the whole CU's LDS: a launch that carries the token-index job asks for more than the kernel's own layout)
        NVQA_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int nb = 0;
        NVQA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NVQA_PF_THREADS, lds));
        *resident = nb;
    }
    if (*resident < 1 || grid > c->num_cus) {
        set_error("persistent BPTT kernel cannot be co-resident (%d workgroups, %d CUs, %d per CU)", grid, c->num_cus, *resident);
        return -1;
    }
    return 0;
}

// round 4: the direct-operand form (lstm_persist_bwd3.h); NVQA_BWD_KERNEL=2 keeps round 3's LDS-ring form (A/B runs, fallback tests)
template <int GKT, int TILES, int NH, int NTN, int PD, bool BF, bool RAG>
static int launch_persist_bwd3(nvqa_ctx *c, PersistBwd2Args a, int grid)
{
    {   // the counter block laid out for NH chains (lstm_persist_bwd3.h)
        const size_t n_rec = (size_t)a.L * a.RB * NH * a.TS, n_up = (size_t)a.L * a.RB * NH * a.NU * a.TS;
        if (n_rec + n_up + 4 > c->pb_cnt_words) { set_error("persistent BPTT: counter block too small"); return -1; }
        a.cnt_rec = c->pb_cnt; a.cnt_up = c->pb_cnt + n_rec;
    }
    size_t lds = PersistBwd3Geom<pb3_mt(TILES, NH, 0), NTN, NH>::LDS_BYTES;
    if (a.jobs && c->ride.has_tok) lds = std::max(lds, tok_index_lds(c->ride.tok.VT, c->ride.tok.NP));
    static int resident = -1;
    NVQA_TRY(check_resident(c, k_lstm_bwd_persist3<GKT, TILES, NH, NTN, PD, BF, RAG>, lds, grid, &resident));
    hipLaunchKernelGGL((k_lstm_bwd_persist3<GKT, TILES, NH, NTN, PD, BF, RAG>), dim3(grid), dim3(NVQA_PF_THREADS), lds, c->s, a);
    NVQA_HIP(hipGetLastError());
    return 0;
}

int lstm_backward_persist(nvqa_ctx *c, const Drop &dr, int MT, int RB)
{
    const nvqa_dims &d = c->d;
    const int B = d.B, R = d.R, L = d.L, TS = c->TS;
    const int NU = R / (16 * persist_bwd_ntn(c));
    if (c->bf16) { // bf16 image of dG, [L][TS*B][4R] (first use of the bf16 instance)
        const size_t gs = (size_t)TS * B * 4 * R;
        if (!c->dg_b16) NVQA_HIP(hipMalloc((void **)&c->dg_b16, (size_t)L * gs * 2));
    }
    const unsigned lim = c->pf_spin ? c->pf_spin : NVQA_PF_SPIN_LIMIT;
    static const int dbg = [] { const char *e = getenv("NVQA_PB_DBG"); return e ? atoi(e) : 0; }();
    const size_t n_rec = (size_t)L * RB * 2 * TS, n_up = (size_t)L * RB * 2 * NU * TS; // per half
    if (n_rec + n_up + 4 > c->pb_cnt_words) { set_error("persistent BPTT: counter block too small"); return -1; }
    c->pb_bias_rb = c->pb_bias ? RB : 0;
    const int grid = 256; // 8 XCDs x 32 slots (the kernels map groups to XCDs); (2L-1) * RB * NU of them have work
    double flops = 0;
    for (int l = 0; l < L; ++l) flops += 2.0 * B * 4 * R * R * ((double)(TS - 1) + (l + 1 < L ? TS : 0));
    // Ride-along jobs (ride_jobs.h): do the idle workgroups exist, and do the weight-gradient products fit?  They must not
    // lengthen the launch.  Calibration (f32, L = 2, B = 512, T = 26: a 0.90 ms launch, 16 riding workgroups): 3.2 GFLOP of
    // 64 x 64 tiles (dW_o + dW_q) leave the launch at 0.90 ms, every further 1.07 GFLOP (256 rows of dW_v) lengthened it by
    // 0.15 ms, then 0.24 ms (a rider runs its 64 x 64 x 512 tiles in 15-17 us each = 0.28 TFLOP/s, 46 % of one CU's f32 peak: one
    // workgroup of four waves per CU, nothing to overlap its load / store / barrier phases with; two tiles in flight (PF = 2)
    // changed nothing).  So the load is capped at 3.4 GFLOP per 0.90 ms of launch and 16 riders; a BPTT step takes ~34 us in f32 (L = 2;
    // ~21 for L = 1) and ~17 us in bf16.  What does not fit is computed behind the launch (ride_flush).
    const int n_idle = (8 * std::max(1, 32 / NU) - (2 * L - 1) * RB) * NU;
    const bool take_jobs = (c->tok_job_pending || c->ride_gemm_pending) && n_idle > 0 &&
                           (!c->tok_job_pending || tok_index_lds(c->ride.tok.VT, c->ride.tok.NP) <= 160 * 1024);
    // (the calibration is for 7 row tiles per workgroup in f32 (B = 512: 0.9 ms then, 0.81 ms with the round-4 kernel)
    // and 4 in bf16; a launch with fewer row tiles per workgroup is shorter in proportion, and so is what may ride in it.  Round 4: so is
    // the launch of a RAGGED batch -- its roles are done after 0.62 ms at a mean length of 0.56 T (0.81 ms at full length) while 3.2
    // GFLOP of riders kept the launch alive for 0.80 ms: measured, ragged step 2.39 -> 2.26 ms with the products behind the launch --
    // roles(f) = roles(1) x (0.47 + 0.53 f) with f the filled share of the batch's (row, step) slots.)
    const int tiles_wg = ((B + 15) / 16 + RB - 1) / RB;
    const double len_scale = d.arch == NVQA_ARCH1 && !c->batch_uniform ? 0.47 + 0.53 * std::min(1.f, std::max(0.f, c->batch_len_frac)) : 1.0;
    const double bptt_ms = 1e-3 * TS * (c->bf16 ? 16.0 : 31.0) * (L == 1 ? 0.62 : 1.0) * std::min(1.0, tiles_wg / (c->bf16 ? 4.0 : 7.0)) * len_scale;
    static const double ride_cap = [] { const char *e = getenv("NVQA_RIDE_CAP"); return e ? atof(e) : 3.8; }(); // GFLOP per 0.90 ms and 16 riders
    // (round 4: 3.4 -> 3.8 with the direct-operand kernel: the 3.2 GFLOP of dW_o + dW_q still end before its roles do -- measured: step 2.926 ms with
    // them riding, 2.969 with them behind the launch -- and the launch estimate above shrank with the kernel)
    const double cap_flops = 1e9 * ride_cap * (bptt_ms / 0.90) * (n_idle / 16.0);
    // the products ride in list order as long as they fit; the rest is computed behind the launch (ride_flush)
    int n_ride = 0;
    double ride_flops = 0;
    if (take_jobs && c->ride_gemm_pending)
        for (int i = 0; i < c->ride.ngemm; ++i) {
            const double fl = 2.0 * c->ride.gm[i].g.M * c->ride.gm[i].g.N * c->ride.gm[i].g.K;
            if (ride_flops + fl > cap_flops) break;
            ride_flops += fl;
            n_ride = i + 1;
        }
    const bool keep_gemms = take_jobs && c->ride_gemm_pending && n_ride < c->ride.ngemm; // some (or all) stay behind
    // (the riding products are booked under an entry of their own -- they run on CUs the BPTT roles do not use,
    // and counting their FLOPs in the BPTT phase would flatter its fraction of the peak)
    if (c->prof_on && ride_flops > 0) { c->prof[PF_RIDE].flops += ride_flops; c->prof[PF_RIDE].launches += 1; }
    ProfScope ps(c, PF_LSTM_BWD, flops, 0);
    const bool rag = d.arch == NVQA_ARCH1 && !c->batch_uniform; // as in lstm_forward_persist
    unsigned *err = c->pb_cnt + c->pb_cnt_words - 4;
    {
        PersistBwd2Args a = {};
        for (int l = 0; l < L; ++l) {
            a.Wh[l] = c->P + c->lo.w_h2h[l]; a.Wi[l] = c->P + c->lo.w_i2h[l];
            a.Gt[l] = c->Gt[l]; a.Cs[l] = c->Cs[l];
            a.Pup[l] = l + 1 < L ? c->pb_pup + (size_t)l * TS * B * R : nullptr;
            if (c->bf16) a.Gb[l] = c->dg_b16 + (size_t)l * TS * B * 4 * R;
        }
        a.dCT = c->dCT; a.dHT = c->dHT;
        a.nrows = c->nrows; a.sort_idx = c->sort_idx;
        a.tlast = d.arch == NVQA_ARCH2 ? c->tinfo + 1 : nullptr;
        a.B = B; a.R = R; a.L = L; a.TS = TS; a.RB = RB; a.NU = NU;
        a.dr = dr; a.spin_limit = lim; a.dbg = dbg;
        a.cnt_rec = c->pb_cnt; a.cnt_up = c->pb_cnt + n_rec; a.err = err;
        a.bias_part = c->pb_bias;
        if (c->pb_bias && c->pb_bias_cnt) { // the kernel's last workgroup per (layer, unit tile) writes the bias gradients itself
            a.bias_cnt = c->pb_bias_cnt;
            for (int l = 0; l < L; ++l) { a.bias_i[l] = c->G + c->lo.b_i2h[l]; a.bias_h[l] = c->G + c->lo.b_h2h[l]; }
            c->pb_bias_done = true;
        }
        a.ts = c->pf_ts + 1024;
        if (take_jobs) {
            c->ride.has_tok = c->tok_job_pending ? 1 : 0;
            if (!c->ride_gemm_pending) c->ride.ngemm = c->ride.has_colsum = 0;
            RideJobs kept = c->ride;
            if (keep_gemms) c->ride.ngemm = n_ride;
            if (!c->ride_dev) NVQA_HIP(hipMalloc((void **)&c->ride_dev, sizeof(RideJobs)));
            if (memcmp(&c->ride_dev_host, &c->ride, sizeof(RideJobs)) != 0) { // (the same list every step: uploaded once)
                NVQA_HIP(hipMemcpyAsync(c->ride_dev, &c->ride, sizeof(RideJobs), hipMemcpyHostToDevice, c->s));
                NVQA_HIP(hipStreamSynchronize(c->s));
                memcpy(&c->ride_dev_host, &c->ride, sizeof(RideJobs));
            }
            a.jobs = c->ride_dev;
            c->tok_job_pending = c->ride_gemm_pending = false;
            if (keep_gemms) { // ride_flush computes what did not fit behind the launch (the column sums went along)
                c->ride = kept;
                for (int i = n_ride; i < kept.ngemm; ++i) c->ride.gm[i - n_ride] = kept.gm[i];
                c->ride.ngemm = kept.ngemm - n_ride;
                c->ride.has_colsum = 0;
                c->ride_gemm_pending = true;
            }
        }
    // PD / PDR: fragments in flight per lane in the equal-length / the ragged instance.  A ragged stream skips the MFMAs of row tiles
    // without active rows but still cycles their ring slots, so it is bound by loads in flight x latency, not by MFMAs: as deep a ring
    // as the registers allow (28 quads; 32 spills)
#define NVQA_PB3_GO(GKT, TILES, NH, NTN, PD, PDR, BFv)                                                              \
    do {                                                                                                             \
        if (rag) NVQA_TRY((launch_persist_bwd3<GKT, TILES, NH, NTN, PDR, BFv, true>(c, a, grid)));                   \
        else NVQA_TRY((launch_persist_bwd3<GKT, TILES, NH, NTN, PD, BFv, false>(c, a, grid)));                       \
    } while (0)
        // f32: the direct-operand kernel (0.89 -> 0.81 ms); bf16: round 3's ring kernel is still the faster one (0.46 against 0.54 ms:
        // DESIGN.md section 4.6 says where the direct form loses in that mode).  NVQA_BWD_KERNEL=2 / 3 forces one of them.
        static const int ver_env = [] { const char *e = getenv("NVQA_BWD_KERNEL"); return e ? atoi(e) : 0; }();
        const int ver = ver_env ? ver_env : (c->bf16 ? 2 : 3);
        if (ver == 2) { // (the ring kernel's instances live in a translation unit of their own: persist_bwd_ring.hip)
            NVQA_TRY(launch_persist_bwd_ring(c, a, grid, MT, rag));
        } else if (c->bf16) { // four chains of one row tile (a chain-step is ~2 us: two chains do not hide a 5 us hand-off)
            if (L == 1) NVQA_PB3_GO(16, 4, 4, 2, 32, 32, true); else NVQA_PB3_GO(16, 4, 4, 4, 32, 32, true);
        } else {              // two chains of 2 + 2 or 4 + 3 row tiles
            if (MT == 4) NVQA_PB3_GO(32, 4, 2, 2, 16, 16, false); else NVQA_PB3_GO(32, 7, 2, 2, 16, 28, false);
        }
#undef NVQA_PB3_GO
    }
    NVQA_TRY(persist_latch_err(c, c->pb_cnt, c->pb_cnt_words, 4));
    return 0;
}

} // namespace nvqa
