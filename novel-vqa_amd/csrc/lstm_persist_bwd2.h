// lstm_persist_bwd2.h -- BPTT through the LSTM stack (misc/RNNUtils.lua:182-209 driving the nngraph backward of
// misc/LSTM.lua:12-73) as ONE persistent, weight-stationary launch whose workgroups run TWO independent row chains.
//
// Per (layer l, step s) the chain needs   dh^l_s = dG^l_{s+1} W_h2h^l  +  Dropout'(dG^{l+1}_s W_i2h^{l+1})  + head term
// and then the cell backward, which turns the stored gates of (l, s) into dG^l_s in place.  All of K (= 4R gate
// pre-activations of the step after) depends on the previous step of the chain: unlike the forward kernel there is no
// independent K segment to multiply while a hand-off travels.  Round 2's kernel (one chain per workgroup) therefore paid
// counter wait + first-chunk latency + write-through drain serially in every step: 43 us per step against 24 us of MFMAs.
//
// The recurrence is independent per batch row.  Here every workgroup (one per CU, 4 waves, 512 registers each) still owns
//   (role, row block rb, unit tile ut)      role REC(l) = recurrent product + cell backward, UP(l) = the product that
//                                           carries the gradient from layer l+1 down;  K = 4R, N = 16 NTN units
// with wave w holding the K-quarter of gate w of its weight block in registers for the whole launch, but its row block is
// cut into two HALVES (row tiles of 16 dealt alternately: tile 2m + h belongs to half h) that are separate chains with
// their own counters.  The workgroup alternates  (h = 0, s) -> (h = 1, s) -> (h = 0, s - 1) ...; while it multiplies one
// half, the other half's results travel: the write-through drain + signal of a half-step are deferred under the first
// chunk of the next half-step, the counter of the next half-step was satisfied a whole half-step ago, and its first chunks
// are requested under the tail of the running product.  Per half-step the only serial parts left are the K-quarter
// reduction through LDS and the cell backward itself.
//
// Hand-offs (MI355X_MICROARCH.md "Valid forms" row 1, as lstm_persist.h): every handed-off byte is stored sc1
// (write-through), every storing wave drains vmcnt (a counted wait: the loads issued since are younger), the workgroup
// barriers, ONE lane adds 1 to the (layer, row block, half, step) counter; consumers poll per wave with sc1 loads, one
// chunk ahead of the first use, and read the bytes with sc1 buffer loads.  REC(l, h, s) waits for REC(l, h, s+1) [all
// unit tiles] and, below the top layer, for the UP(l, h, s) tile with its rows and units; UP(l, h, s) waits for
// REC(l+1, h, s).  A signal is issued only after waits on strictly earlier half-steps: no cycle.  Every spin is bounded.
//
// bf16 instance (nvqa_set_precision(1)): v_mfma_f32_16x16x32_bf16, weights as packed bf16 (a workgroup holds 64 units),
// the A operand comes from the bf16 image of dG that the cell backward writes next to the f32 one; the LDS image of a
// chunk has the f32 layout byte for byte (a 16-byte piece is a lane's A fragment: 4 f32 or 8 bf16).
#pragma once
#include "tok_index.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "lstm_persist.h"
#include "ride_jobs.h"

namespace nvqa {

struct PersistBwd2Args {
    const float *Wh[NVQA_PF_MAXL], *Wi[NVQA_PF_MAXL]; // W_h2h^l [4R][R]; W_i2h^l [4R][R] for l >= 1
    float *Gt[NVQA_PF_MAXL];                          // [TS*B][4R] gates in, d(pre-activations) out
    const float *Cs[NVQA_PF_MAXL];                    // [(TS+1)*B][R]
    const float *dCT, *dHT;                           // [L][B][R] head -> final cell / hidden state gradients
    float *Pup[NVQA_PF_MAXL];                         // Pup[l], l < L-1: [TS*B][R] products of the UP(l) role
    unsigned short *Gb[NVQA_PF_MAXL];                 // bf16 instance: [TS*B][4R] bf16 image of dG (written by REC, read as A)
    float *bias_part;                                 // [L][RB][4R]: column sums of dG over the steps and the rows of a row block
    unsigned *bias_cnt;                               // [L][NU] arrivals of the row blocks of a (layer, unit tile): the last one adds the row blocks'
    float *bias_i[NVQA_PF_MAXL], *bias_h[NVQA_PF_MAXL]; //   partial sums in row-block order into both bias gradients of the layer (round 3: k_bias_sum,
                                                      //   one more launch per layer); NULL: the partial sums are all the kernel leaves
    const int *nrows, *sort_idx, *tlast;              // tlast (arch2): dHT enters at step *tlast; NULL (arch1): at TS-1
    unsigned *cnt_rec;                                // [L][RB][2][TS] arrivals of the REC(l) unit tiles, per half
    unsigned *cnt_up;                                 // [L][RB][2][NU][TS] flag of the UP(l) tile, per half
    unsigned *err;
    unsigned spin_limit;
    unsigned long long *ts;  // dbg & 32: per workgroup {start, weights resident, steps done} in 100 MHz ticks
    int dbg;                 // measurement only (NVQA_PB_DBG): 1 no flag waits, 2 no cell math / stores, 8 A loads without memory traffic
    int B, R, L, TS, RB, NU; // NU = R / (16 NTN) unit tiles
    Drop dr;
    const RideJobs *jobs;    // != NULL: work for the workgroups without a role (ride_jobs.h): a device copy of the job list (its
                             // arguments are the context's own buffers: written once)
};

// MTA / MTB: row tiles of 16 rows in half 0 / half 1 (MTA >= MTB); NTN column tiles of 16 units; GPC K groups per gate
// and chunk (a group = one A-fragment read = 16 k in f32, 32 k in bf16; a chunk = one ring stage, one barrier)
template <int MTA, int MTB, int NTN, int GPC> struct PersistBwd2Geom {
    static constexpr int MTH = MTA, ROWSH = 16 * MTH, UNITS = 16 * NTN;
    // ring stages: a stage is read (fragment refills from `cur`) only in front of its iteration's barrier and written only in
    // front of the NEXT iteration's barrier, the reads behind a barrier come from the stage just written: two suffice
    static constexpr int NST = 2;
    static constexpr int PPR = 16 * GPC;             // 16-byte pieces per row and chunk: 4 gates x GPC groups x 4
    static constexpr int ROWW = 4 * PPR;             // floats per row of a ring stage
    static constexpr int STAGE = ROWSH * ROWW;       // floats per ring stage
    static constexpr int BSUM_FLOATS = NVQA_PF_THREADS * 16;
    static constexpr int QPR = UNITS / 4, RPP = NVQA_PF_THREADS / QPR, NE = (ROWSH + RPP - 1) / RPP; // epilogue items per thread and half
    static constexpr int DC_FLOATS = NVQA_PF_THREADS * 2 * NE * 4 * 2; // carried cell gradient + carried cell state
    // row stride of the partial tiles: a lane's 4 accumulator rows are 4 apart per lane group, so with a stride of UNITS
    // (32 or 64 floats) the four lane groups of a ds_write_b32 hit the same 16 banks (r03 PMC: 21 % / 46 % of the kernel's
    // LDS cycles were bank conflicts); UNITS + 4 moves each lane group 16 banks on and keeps rows 16-byte aligned
    static constexpr int SROW = UNITS + 4;
    static constexpr size_t LDS_BYTES = (size_t)(NST * STAGE + 4 * ROWSH * SROW + BSUM_FLOATS + DC_FLOATS) * 4;
};

template <int N> __device__ __forceinline__ void pb_wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// GKT: K groups per gate in all (R / 16 in f32, R / 32 in bf16)
template <int GKT, int MTA, int MTB, int NTN, int GPC, bool BF, bool RAG>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_bwd_persist2(PersistBwd2Args a)
{
    typedef PersistBwd2Geom<MTA, MTB, NTN, GPC> GE;
    static_assert(MTA >= MTB && MTB >= 1, "half 0 is the larger half");
    static_assert(GKT % GPC == 0, "whole chunks");
    constexpr int D = 2;                       // chunks in flight = staging-register sets
    constexpr int NT = GKT / GPC;              // chunks per half-step
    static_assert(NT >= 4 && NT % D == 0, "static staging-register sets across half-steps; polls are requested 3 chunks before the end");
    constexpr int ROWSH = GE::ROWSH, NST = GE::NST, STAGE = GE::STAGE, UNITS = GE::UNITS, PPR = GE::PPR, ROWW = GE::ROWW, SROW = GE::SROW;
    constexpr int ES = BF ? 2 : 4;             // bytes per A element
    constexpr int RPS = NVQA_PF_THREADS / PPR; // rows per staging pass (16 / GPC)
    static_assert(16 % RPS == 0, "a staging pass stays inside one row tile");
    constexpr int NLDA = 16 * MTA / RPS, NLDB = 16 * MTB / RPS, NLDM = NLDA; // loads per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float pb2_smem[];
    float *const ring = pb2_smem;                   // [NST][ROWSH][ROWW]: 16-byte pieces, low 4 bits of the piece index XOR-swizzled by the row
    float *const Sred = pb2_smem + NST * STAGE;     // [4 waves][ROWSH][SROW] partial tiles
    float *const bsum = Sred + 4 * ROWSH * SROW;    // [4 gates][thread][4 units]: sum of the thread's dG over its rows and all steps (thread-minor:
                                                    // consecutive lanes -> consecutive 16-byte slots, conflict-free; thread-major strides of 64 / 128 B were 4- / 8-way conflicts)
    float *const dcs = bsum + GE::BSUM_FLOATS;      // [half][item][2][thread][4 units]: the carried cell gradient, and c_s of the step just done (= c_{s-1}
                                                    // of the next one: the thread owns the same cells at every step) -- thread-private slots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int B = a.B, R = a.R, TS = a.TS, L = a.L, RBn = a.RB;

    // workgroup -> (role, row block, unit tile).  roles: 0 .. L-1 = REC(l = L-1-role), top layer first; L .. 2L-2 = UP(l = 2L-2-role).
    // The NU unit tiles of one (role, row block) group exchange dG among themselves: they sit on ONE XCD (workgroups are dealt
    // round-robin to the 8 XCDs: id % 8), REC(l+1, rb) next to the UP(l, rb) that reads its output.  Speed only: the protocol does
    // not depend on the placement.  Grid = 8 XCDs x 32 slots; slots without a group leave at once.
    const int ngroups = (2 * L - 1) * RBn, gpx = 32 / a.NU > 0 ? 32 / a.NU : 1;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int gslot = xcd * gpx + slot / a.NU, ut = slot % a.NU;
    if (slot / a.NU >= gpx || gslot >= ngroups) {
        // a slot without a role.  The workgroups of the slot groups behind the last role (f32, L = 2: one group of 16; bf16:
        // 8 groups of 8) share the step's ride-along jobs (ride_jobs.h; the host made the dynamic LDS large enough): off the
        // critical path, no launch of their own.  (inlined: as a real call it measured 0.01 ms slower per launch -- the
        // callee's register needs then shape the whole kernel's allocation)
        if (a.jobs && slot / a.NU < gpx) ride_jobs_run<BF>(a.jobs, (gslot - ngroups) * a.NU + ut, (8 * gpx - ngroups) * a.NU, pb2_smem);
        return;
    }
    int role, rb;
    if (L == 1) { role = 0; rb = gslot; }
    else if (gslot < 2 * RBn) { role = (gslot & 1) ? L : 0; rb = gslot >> 1; }
    else { role = 1; rb = gslot - 2 * RBn; }
    const bool is_up = role >= L;
    const int l = is_up ? 2 * L - 2 - role : L - 1 - role; // the layer whose dh this tile belongs to
    const int la = is_up ? l + 1 : l;                      // the layer whose dG is the A operand
    const int u0 = ut * UNITS;
    const bool has_up = !is_up && l + 1 < L;
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4] = wall_clock64();

    // ---- weights: rows k = wave * R + kk (gate `wave`), columns u0 .. u0 + UNITS - 1 of W [4R][R]; resident B fragments ----
    constexpr int KG = BF ? 32 : 16;
    const float *W = is_up ? a.Wi[l + 1] : a.Wh[l];
    pf_u32x4 bw[NTN][GKT]; // f32: 4 k = 16 g + 4 lh + w; bf16: 8 k = 32 g + 8 lh + j (packed pairs)
    if constexpr (!BF) {
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int g = 0; g < GKT; ++g) {
                const float *w0 = W + (size_t)(wave * R + KG * g + (KG / 4) * lh) * R + u0 + 16 * nt + li;
                bw[nt][g] = __builtin_bit_cast(pf_u32x4, pf_f32x4{w0[0], w0[(size_t)R], w0[2 * (size_t)R], w0[3 * (size_t)R]});
            }
    } else {
        // bf16: 8 f32 values per fragment, rounded and packed.  Loaded in BATCHES of WB fragments (8 WB loads in flight), packed
        // afterwards: fragment by fragment -- load 8, pack, pin in an AGPR quad -- the prologue was a chain of GKT x NTN = 64 global
        // round trips, 60-80 us of a 0.45 ms launch (r4 timestamps: "weights resident after 62.7 .. 80.2 us" against 6 .. 12 us in f32).
        constexpr int WB = 8;
        static_assert(GKT % WB == 0, "whole batches");
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int g0 = 0; g0 < GKT; g0 += WB) {
                float raw[WB][8];
#pragma unroll
                for (int gg = 0; gg < WB; ++gg) {
                    const float *w0 = W + (size_t)(wave * R + KG * (g0 + gg) + (KG / 4) * lh) * R + u0 + 16 * nt + li;
#pragma unroll
                    for (int j = 0; j < 8; ++j) raw[gg][j] = w0[(size_t)j * R];
                }
#pragma unroll
                for (int gg = 0; gg < WB; ++gg) {
                    pf_u32x4 q;
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = pf_pack_bf16(raw[gg][2 * j], raw[gg][2 * j + 1]);
                    // the fragment is born as ONE 128-bit value in an aligned AGPR quad and stays there (the MFMA's B operand, "a"
                    // constraint): assembled from four 32-bit values hipcc kept the pieces apart and copied them into a scratch
                    // AGPR quad in front of every MFMA -- and, not knowing that the asm statement is a matrix instruction, wrote
                    // that quad again while the previous MFMA was still reading it (all-NaN gradients)
                    asm volatile("" : "+a"(q));
                    bw[nt][g0 + gg] = q;
                }
            }
    }
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 1] = wall_clock64();

    const size_t gt_bytes = (size_t)TS * B * 4 * R * 4, pup_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_a = BF ? pf_rsrc(a.Gb[la], gt_bytes / 2) : pf_rsrc(a.Gt[la], gt_bytes); // A operand: dG of layer la
    const __amdgpu_buffer_rsrc_t r_gb = BF ? pf_rsrc(a.Gb[l], gt_bytes / 2) : r_a;   // REC, bf16: the image of dG it writes
    const __amdgpu_buffer_rsrc_t r_g = pf_rsrc(a.Gt[l], gt_bytes);                   // REC: gates in / dG out
    const __amdgpu_buffer_rsrc_t r_p = pf_rsrc(l + 1 < L ? a.Pup[l] : a.Gt[l], l + 1 < L ? pup_bytes : gt_bytes);

    // local row i of the block <-> sorted batch row rb + RBn i; rows 0 .. nloc-1 exist.  Row rho of half h is local row
    // rho + 16 (rho / 16) + 16 h (tile 2m + h).
    const int nloc = (B - rb + RBn - 1) / RBn;
    // staging map of a chunk: thread -> (row prow + RPS j of the half, piece pp): gate pp / (4 GPC), 16-byte piece pp % (4 GPC) of
    // the chunk's GPC groups of that gate
    const int prow = tid / PPR, pp = tid % PPR;
    const unsigned row_bytes = 4u * R * ES, stride1 = (unsigned)RBn * row_bytes, step_bytes = (unsigned)B * row_bytes;
    const unsigned toff = (unsigned)(rb + RBn * prow) * row_bytes + ((unsigned)(pp / (4 * GPC)) * R * ES) + 16u * (pp % (4 * GPC));
    const unsigned lds_w = (unsigned)(prow * ROWW + 4 * ((pp & ~15) | ((pp & 15) ^ (prow & 15)))); // (rows prow + RPS j: same low 4 bits iff RPS = 16; else recomputed)

    pf_u32x4 stg[D][NLDM];
    // half-step k = 2 (TS-1-s) + h
    auto kstep = [&](int k) { return TS - 1 - (k >> 1); };
    // A slice of half-step k: REC: dG^l_{s+1}; UP: dG^{l+1}_s
    auto ksa = [&](int k) { return is_up ? kstep(k) : kstep(k) + 1; };

    // issue the loads of chunk c of a half-step (half H, A slice sa) into staging set SET, pieces j0 .. j1-1
    unsigned pf_o0 = PF_OOB;
    int pf_act = 0; // RAG: active tiles of the half whose chunk is being requested
    auto prefetch_begin = [&](auto h_tag, int sa, int c, bool en, int act) {
        constexpr int H = decltype(h_tag)::value;
        const unsigned enm = __builtin_amdgcn_readfirstlane((en && !(a.dbg & 8)) ? 0u : PF_OOB);
        pf_o0 = (toff + (unsigned)sa * step_bytes + 64u * GPC * (unsigned)c + 16u * H * stride1) | enm;
        pf_act = act;
    };
    auto prefetch_piece = [&](auto h_tag, auto set_tag, auto j0_tag, auto j1_tag) {
        constexpr int H = decltype(h_tag)::value, SET = decltype(set_tag)::value, NLD = H ? NLDB : NLDA;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < NLD; ++j) {
            const int m = (RPS * j) / 16;                    // row tile of the half
            const int iloc = prow + RPS * j + 16 * m + 16 * H; // local row
            bool ok = iloc < nloc;
            if constexpr (RAG) ok = ok && m < pf_act;
            const unsigned off = ok ? pf_o0 + (unsigned)(RPS * j + 16 * m) * stride1 : PF_OOB;
            stg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(r_a, off, 0, 16 /* sc1 */);
        }
    };
    auto commit_piece = [&](auto h_tag, auto set_tag, int stage, auto j0_tag, auto j1_tag) {
        constexpr int H = decltype(h_tag)::value, SET = decltype(set_tag)::value, NLD = H ? NLDB : NLDA;
        float *dst = ring + stage * STAGE;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < NLD; ++j) {
            if constexpr (RPS == 16) {
                *reinterpret_cast<pf_u32x4 *>(&dst[lds_w + (unsigned)(RPS * j * ROWW)]) = stg[SET][j];
            } else {
                const int row = prow + RPS * j;
                *reinterpret_cast<pf_u32x4 *>(&dst[row * ROWW + 4 * ((pp & ~15) | ((pp & 15) ^ (row & 15)))]) = stg[SET][j];
            }
        }
    };
    const auto J0 = std::integral_constant<int, 0>{};
    const auto JALL = std::integral_constant<int, NLDM>{};

    pf_f32x4 acc[GE::MTH][NTN];
    pf_u32x4 af[GE::MTH];
    // this wave's fragment of row tile m, group gi of the chunk in `src`: piece 4 GPC wave + 4 gi + lh
    auto frag_at = [&](const float *src, int m, int gi) -> pf_u32x4 {
        const int p = 4 * GPC * wave + 4 * gi + lh;
        return *reinterpret_cast<const pf_u32x4 *>(&src[(m * 16 + li) * ROWW + 4 * ((p & ~15) | ((p & 15) ^ li))]);
    };
    // MFMAs of row tile m for resident group g: f32: the NTN MFMAs of k-quarter w (4 per group); bf16: the NTN MFMAs of the group.
    // Inline asm with the B operand constrained to the accumulation registers ("a"): the resident weights then LIVE in the 256
    // AGPRs and the matrix instruction reads them there.  Through the builtin hipcc treats them as ordinary VGPR values parked in
    // AGPRs: one v_accvgpr_read copy (+ s_nop) in front of every MFMA and, with the VGPR file full, scratch spills of the
    // epilogue's values (reloaded behind vmcnt(0) waits).  What the compiler no longer sees, and the code provides: the wait
    // states between the VALU writes that clear an accumulator and its first MFMA (pb_nop_after_clear) and between the last
    // MFMA and the first read of an accumulator (pb_nop_before_read).
    auto mfma_w = [&](auto g_tag, auto m_tag, auto w_tag) {
        constexpr int g = decltype(g_tag)::value, m = decltype(m_tag)::value, w = decltype(w_tag)::value;
        if constexpr (!BF) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) { // (locals: operands named only inside an asm statement are not captured by the lambda)
                pf_f32x4 &c = acc[m][nt];
                const float av = __builtin_bit_cast(pf_f32x4, af[m])[w], bv = __builtin_bit_cast(pf_f32x4, bw[nt][g])[w];
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                pf_f32x4 &c = acc[m][nt];
                const pf_u32x4 av = af[m], bv = bw[nt][g];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
            }
        }
    };
    // The wait states must sit BETWEEN the instructions they separate, and the compiler may move anything that does not
    // depend on an asm statement across it.  So the accumulators pass THROUGH empty asm statements ("+v"): the v_movs that
    // clear them are ordered in front of the first one, the nops follow, and the first MFMA consumes the asm's outputs;
    // likewise the first read of an accumulator consumes the output of an asm that follows the nops behind the last MFMA.
    // (Without this a ragged bf16 instance came out with its clears scheduled directly in front of the first MFMAs: LSTM
    // gradients 10-30 % off while every other instance happened to pass.)
    auto pb_touch_acc = [&] {
#pragma unroll
        for (int m = 0; m < GE::MTH; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) asm volatile("" : "+v"(acc[m][nt]));
    };
    auto pb_nop_after_clear = [&] { pb_touch_acc(); asm volatile("s_nop 7" ::: "memory"); };
    auto pb_nop_before_read = [&] { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); pb_touch_acc(); };

    // epilogue ownership: thread -> (row rho = erow + RPP e of the half, units u0 + 4 eq .. +3)
    constexpr int QPR = GE::QPR, RPP = GE::RPP, NE = GE::NE;
    static_assert(RPP % 16 == 0, "an epilogue pass covers whole row tiles");
    const int eq = tid % QPR, erow = tid / QPR;
    // local row of item e of half h: rho + 16 (rho / 16) + 16 h with rho = erow + RPP e
    auto eloc = [&](int h, int e) { const int rho = erow + RPP * e; return rho + 16 * (rho >> 4) + 16 * h; };
    int esi[2][NE];       // original batch row of the owned rows (indexes the dropout stream)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iloc = eloc(h, e), grow = rb + RBn * iloc;
            const bool ok = erow + RPP * e < 16 * (h ? MTB : MTA) && iloc < nloc;
            pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!is_up && ok) v = *reinterpret_cast<const pf_f32x4 *>(a.dCT + ((size_t)l * B + grow) * R + u0 + 4 * eq);
            *reinterpret_cast<pf_f32x4 *>(dcs + (((h * NE + e) * 2) * NVQA_PF_THREADS + tid) * 4) = v; // REC: the carried cell gradient of the owned (row, unit)s
            {   // ... and the final cell state c_{TS-1} (slice TS of Cs)
                pf_f32x4 cfin = {0.f, 0.f, 0.f, 0.f};
                if (!is_up && ok) cfin = *reinterpret_cast<const pf_f32x4 *>(a.Cs[l] + ((size_t)TS * B + grow) * R + u0 + 4 * eq);
                *reinterpret_cast<pf_f32x4 *>(dcs + (((h * NE + e) * 2 + 1) * NVQA_PF_THREADS + tid) * 4) = cfin;
            }
            esi[h][e] = a.sort_idx[ok ? grow : 0];
        }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<pf_f32x4 *>(bsum + (g * NVQA_PF_THREADS + tid) * 4) = pf_f32x4{0.f, 0.f, 0.f, 0.f};

    // counters
    const unsigned crec = (unsigned)(((l * RBn + rb) * 2) * TS);            // REC(l) counters of this row block: + h * TS + s
    const unsigned cneed = (unsigned)(((la * RBn + rb) * 2) * TS);          // counters of the producers of the A operand
    const unsigned cup = (unsigned)((((l * RBn + rb) * 2) * a.NU) * TS);    // UP(l) flags of this row block: + (h * NU + ut) * TS + s
    auto need_word = [&](int k) { return a.cnt_rec + cneed + (unsigned)((k & 1) * TS + ksa(k)); };
    auto own_word = [&](int k) {
        return is_up ? a.cnt_up + cup + (unsigned)(((k & 1) * a.NU + ut) * TS + kstep(k)) : a.cnt_rec + crec + (unsigned)((k & 1) * TS + kstep(k));
    };
    auto up_word = [&](int k) { return a.cnt_up + cup + (unsigned)(((k & 1) * a.NU + ut) * TS + kstep(k)); };
    // RAG: active tiles of half h at step s (rows dealt round-robin: the active local rows are a prefix)
    auto act_of = [&](int h, int s) -> int {
        if constexpr (!RAG) return h ? MTB : MTA;
        const int nr = a.nrows[s < 0 ? 0 : (s >= TS ? TS - 1 : s)];
        const int tiles = ((nr > rb ? (nr - rb + RBn - 1) / RBn : 0) + 15) >> 4; // local tiles with active rows
        const int t = (tiles - h + 1) >> 1;
        return __builtin_amdgcn_readfirstlane(min(h ? MTB : MTA, max(t, 0)));
    };

    // cell-backward operands of one item (own gates and cell states of the forward pass: nothing in this launch writes
    // them before this workgroup does: default-policy loads).  Every load of the pipelined loop is issued on EVERY path --
    // the UP role gets out-of-range offsets, which return zeros without touching memory -- because a load under a runtime
    // branch makes hipcc wait vmcnt(0) at the join (lstm_persist.h).
    const __amdgpu_buffer_rsrc_t r_cs = pf_rsrc(a.Cs[l], (size_t)(TS + 1) * B * R * 4);
    const unsigned upm = __builtin_amdgcn_readfirstlane(is_up ? PF_OOB : 0u);
    pf_f32x4 e_ig[NE], e_fg[NE], e_og[NE], e_gg[NE], e_cp[NE], e_v2[NE];
    auto ld = [&](const __amdgpu_buffer_rsrc_t &r, unsigned off) { return __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); };
    auto fetch = [&](int h, int s, int e) {
        const int iloc = eloc(h, e), grow = min(rb + RBn * iloc, B - 1);
        const unsigned srow_g = (unsigned)s * B + grow, q4 = (unsigned)(u0 + 4 * eq);
        const unsigned go = ((srow_g * 4u * R + q4) * 4u) | upm;
        e_ig[e] = ld(r_g, go);
        e_fg[e] = ld(r_g, go + (unsigned)R * 4);
        e_og[e] = ld(r_g, go + 2u * R * 4);
        e_gg[e] = ld(r_g, go + 3u * R * 4);
        e_cp[e] = ld(r_cs, ((srow_g * R + q4) * 4u) | upm);
    };
    const int s_head = __builtin_amdgcn_readfirstlane(a.tlast ? *a.tlast : TS - 1); // the step at which dHT enters
    // v2: what is added to the product before the cell backward.  Below the top layer: the UP(l, h, s) products of the owned cells
    // (another workgroup's bytes: after its flag, sc1).  Top layer: the head term dHT at the one step where it enters, nothing
    // otherwise (out-of-range offset: zeros without memory traffic).
    const __amdgpu_buffer_rsrc_t r_v2 = has_up ? r_p : pf_rsrc(a.dHT, (size_t)L * B * R * 4);
    auto fetch_v2 = [&](int h, int s) {
        const bool live_v2 = has_up || (!is_up && s == s_head);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iloc = eloc(h, e), grow = min(rb + RBn * iloc, B - 1);
            const unsigned off = has_up ? (unsigned)((((size_t)s * B + grow) * R + u0 + 4 * eq) * 4) : (unsigned)((((size_t)l * B + grow) * R + u0 + 4 * eq) * 4);
            e_v2[e] = __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_v2, live_v2 ? off : PF_OOB, 0, 16));
        }
    };

    // ---- reduction of the four K-quarters + cell backward (REC) / tile store (UP) of half-step (h, s) --------------------
    // Cut in two so that it can be DEFERRED: spill_acc ends a half-step's product (accumulators -> LDS); cell_item finishes one
    // of the thread's NE items from there.  In the pipelined loop the items of half-step k run behind the first barriers of
    // half-step k+1 (one item per iteration): done right after the product they cost 2.3 us per half-step of latency -- the
    // 32 wait states, the LDS round trip through Sred with its own barrier, the read-modify-writes of the carried state --
    // with the matrix pipe idle; behind the next product's barrier they are ~150 vector instructions per item.
    auto spill_acc = [&](auto h_tag) {
        constexpr int H = decltype(h_tag)::value, MT = H ? MTB : MTA;
        pb_nop_before_read();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sred[(wave * ROWSH + 16 * m + 4 * lh + r) * SROW + 16 * nt + li] = acc[m][nt][r];
    };
    // prod = false: no product at this half-step (REC at the last step)
    auto cell_item = [&](auto h_tag, auto e_tag, auto imm_tag, int s, bool prod, int nr) {
        constexpr int H = decltype(h_tag)::value, MT = H ? MTB : MTA, e = decltype(e_tag)::value;
        constexpr bool IMM = decltype(imm_tag)::value;
        const int rho = erow + RPP * e, iloc = eloc(H, e), grow = rb + RBn * iloc;
        if (rho >= 16 * MT || iloc >= nloc || (a.dbg & 2)) return;
        // Head term: dL/dh of the final state enters at ONE step (arch1: the last; arch2: tmax).  The TOP layer receives it through
        // the v2 operand (fetch_v2 reads dHT there at that step: a load requested a chunk ahead, on every path).  A layer BELOW
        // the top has a head term only in arch1 (q = c and h of all layers, 002_train_baseline.lua:306), at s = TS-1, which is
        // never a pipelined half-step: loaded here in the immediate form only -- a load consumed at once inside the pipelined
        // loop would drain every prefetch in flight (loads return in order).  (arch2: the host zeroes dHT below the top layer.)
        pf_f32x4 hx = {0.f, 0.f, 0.f, 0.f};
        if constexpr (IMM) {
            if (has_up && s == s_head) hx = *reinterpret_cast<const pf_f32x4 *>(a.dHT + ((size_t)l * B + min(grow, B - 1)) * R + u0 + 4 * eq);
        }
        pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (prod) {
#pragma unroll
            for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const pf_f32x4 *>(&Sred[(w * ROWSH + rho) * SROW + 4 * eq]);
        }
        const size_t srow_g = (size_t)s * B + grow;
        const unsigned uo = (unsigned)((srow_g * R + u0 + 4 * eq) * 4);
        if (is_up) {
            // ship the product, already multiplied by Dropout' of the layer boundary (x 0 or x 1/(1-p): exact, so the cell of
            // layer l adds the same value it would have formed itself).  The mask hash is a third of the cell backward's
            // vector instructions; the UP role's epilogue is otherwise one store, and REC(l) sets the pace of the launch.
            const uint64_t didx = ((((uint64_t)l) * B + esi[H][e]) * TS + s) * R + u0 + 4 * eq;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= a.dr.scale(NVQA_SITE_LSTM, didx + j);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v), r_p, uo, 0, 16);
            return;
        }
        const unsigned go = (unsigned)((srow_g * 4 * R + u0 + 4 * eq) * 4);
        pf_f32x4 dgi = {0.f, 0.f, 0.f, 0.f}, dgf = dgi, dgo = dgi, dgg = dgi, dcn = dgi;
        pf_f32x4 *dcp = reinterpret_cast<pf_f32x4 *>(dcs + (((H * NE + e) * 2) * NVQA_PF_THREADS + tid) * 4); // [0]: dc, [NVQA_PF_THREADS]: c
        if (grow < nr) {
            const pf_f32x4 dc0 = *dcp;
            const pf_f32x4 ig = e_ig[e], fg = e_fg[e], og = e_og[e], gg = e_gg[e], cc = dcp[NVQA_PF_THREADS], cp = e_cp[e], v2 = e_v2[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dh = v[j] + v2[j] + hx[j]; // v2: the UP tile, Dropout' applied by its producer (zeros without one)
                const float tc = pf_tanh(cc[j]);
                const float dcv = dc0[j] + dh * og[j] * (1.0f - tc * tc);
                dgi[j] = dcv * gg[j] * ig[j] * (1.0f - ig[j]);
                dgf[j] = dcv * cp[j] * fg[j] * (1.0f - fg[j]);
                dgo[j] = dh * tc * og[j] * (1.0f - og[j]);
                dgg[j] = dcv * ig[j] * (1.0f - gg[j] * gg[j]);
                dcn[j] = dcv * fg[j];
            }
        }
        *dcp = dcn;
        dcp[NVQA_PF_THREADS] = e_cp[e]; // c_{s-1} (slice s of Cs) is the next step's c_s, active row or not
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgi), r_g, go, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgf), r_g, go + (unsigned)R * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgo), r_g, go + 2u * R * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgg), r_g, go + 3u * R * 4, 0, 16);
        if (grow < nr) { // bias gradient: column sums of dG (own LDS slot: no other thread touches it)
            pf_f32x4 *bs = reinterpret_cast<pf_f32x4 *>(bsum + tid * 4); // gate g at bs[g * NVQA_PF_THREADS]
            bs[0] += dgi; bs[NVQA_PF_THREADS] += dgf; bs[2 * NVQA_PF_THREADS] += dgo; bs[3 * NVQA_PF_THREADS] += dgg;
        }
        if constexpr (BF) { // the image the REC / UP products read (the f32 one stays what the weight gradients read)
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            auto img = [&](const pf_f32x4 &x, unsigned gate) {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pf_pack_bf16(x[0], x[1]), pf_pack_bf16(x[2], x[3])}, r_gb,
                                                      go / 2 + gate * (unsigned)R * 2, 0, 16);
            };
            img(dgi, 0); img(dgf, 1); img(dgo, 2); img(dgg, 3);
        }
    };
    // the immediate form: every item now (the steps without a product; the flush behind the last half-step)
    auto epilogue = [&](auto h_tag, int s, bool prod, int nr) {
        if (prod) spill_acc(h_tag);
        __syncthreads();
        [&]<int... E>(std::integer_sequence<int, E...>) { (cell_item(h_tag, std::integral_constant<int, E>{}, std::true_type{}, s, prod, nr), ...); }(std::make_integer_sequence<int, NE>{});
    };
    auto signal_now = [&](int k) { // not deferred: drain, barrier, one add
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(k)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    const int KN = 2 * TS;              // half-steps
    const int k0 = is_up ? 0 : 2;       // first half-step with a product (REC at the last step has none)
    // ---- REC at the last step: head term (+ the UP tile) and the cell backward only ----------------------------------------
    for (int k = 0; k < k0 && k < KN; ++k) {
        const int s = kstep(k), h = k & 1;
#pragma unroll
        for (int e = 0; e < NE; ++e) fetch(h, s, e);
        if (has_up && !(a.dbg & 1)) (void)pf_wait_ge(up_word(k), 1u, a.err, 0x500u + l, a.spin_limit);
        fetch_v2(h, s);
        const int nr0 = a.nrows[s];
        if (h == 0) epilogue(std::integral_constant<int, 0>{}, s, false, nr0);
        else epilogue(std::integral_constant<int, 1>{}, s, false, nr0);
        signal_now(k);
    }

    // ---- the pipelined half-steps ------------------------------------------------------------------------------------------
    unsigned n = 0;       // running chunk counter: ring stage = n % NST
    unsigned pend = 0, pend_up = 0;
    int pub = -1;         // half-step whose stores are issued but not yet drained and signalled
    int pend_k = -1, pend_nr = 0; // half-step whose product sits in Sred, its items still to run (and its nrows[s])
    if (k0 < KN) {
        // pipeline prologue for half-step k0 (half 0): counter, chunks 0 .. D-1, chunk 0 -> LDS, first fragments
        if (!(a.dbg & 1)) (void)pf_wait_ge(need_word(k0), (unsigned)a.NU, a.err, (is_up ? 0x400u : 0x300u) + l, a.spin_limit);
        const int act0 = act_of(0, kstep(k0));
        prefetch_begin(std::integral_constant<int, 0>{}, ksa(k0), 0, true, act0);
        prefetch_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, J0, JALL);
        prefetch_begin(std::integral_constant<int, 0>{}, ksa(k0), 1, true, act0);
        prefetch_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, J0, JALL);
        commit_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, J0, JALL);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MTA; ++m) af[m] = frag_at(ring, m, 0);
    }

    // one half-step: H = its half (compile time), k its index
    auto half_step = [&](auto h_tag, int k) {
        constexpr int H = decltype(h_tag)::value, HN = 1 - H;
        constexpr int MT = H ? MTB : MTA, MTN = HN ? MTB : MTA, NLD = H ? NLDB : NLDA, NLDN = HN ? NLDB : NLDA;
        const auto HT = std::integral_constant<int, H>{};
        const auto HNT = std::integral_constant<int, HN>{};
        const int s = kstep(k), sa = ksa(k);
        const bool more = k + 1 < KN;
        const int kn = more ? k + 1 : k, sn = kstep(kn), san = ksa(kn);
        const int act = act_of(H, s), actn = act_of(HN, sn);
        const int nr = a.nrows[s]; // (requested here: a load in the epilogue would wait for the prefetches in flight)
#pragma unroll
        for (int m = 0; m < GE::MTH; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[m][nt] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        pb_nop_after_clear();

        auto iter = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value;
            const float *cur = ring + (n % NST) * STAGE, *nxt = ring + ((n + 1) % NST) * STAGE;
            const int nst = (int)((n + 1) % NST);
            constexpr bool ld_own = q + D < NT;           // the chunk requested in this iteration belongs to this half-step
            constexpr int QL = ld_own ? q + D : q + D - NT; // ... its index there
            constexpr bool cm_own = q + 1 < NT;           // the chunk committed in this iteration belongs to this half-step
            constexpr int SETL = q % D, SETC = (q + 1) % D;
            // counters of the next half-step / of this half-step's UP tile: requested a chunk ahead of their check
            if constexpr (q == NT - 3) {
                pend = __hip_atomic_load(more ? need_word(kn) : a.cnt_rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend_up = __hip_atomic_load(has_up ? up_word(k) : a.cnt_rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (q == NT - 2) { // the first chunk of the next half-step is requested below: its producers must be done
                if (more && !(a.dbg & 1) && pend < (unsigned)a.NU)
                    (void)pf_wait_ge(need_word(kn), (unsigned)a.NU, a.err, (is_up ? 0x400u : 0x300u) + l, a.spin_limit);
            }
            if constexpr (q == NT - 1) { // the UP tile of this half-step (normally long since there: UP runs ahead)
                if (has_up && !(a.dbg & 1) && pend_up < 1u) (void)pf_wait_ge(up_word(k), 1u, a.err, 0x500u + l, a.spin_limit);
                fetch_v2(H, s);
            }
            if constexpr (ld_own) prefetch_begin(HT, sa, QL, true, act);
            else prefetch_begin(HNT, san, QL, more, actn);
            constexpr int NLDL = ld_own ? NLD : NLDN, NLDC = cm_own ? NLD : NLDN;
            // The chunk as a sequence of TILE-GROUPS (K group gi of the chunk, row tile m): NW x NTN MFMAs each, cut into NW
            // "gaps" (after the NTN MFMAs of one k-quarter).  Every gap carries at most a few housekeeping micro-ops -- one
            // load of chunk q+D, one LDS write of chunk q+1 -- pinned there by sched_barrier: left to itself hipcc gathers a
            // slot's loads, LDS writes and fragment reads behind its last MFMA, where only that one MFMA covers them (the
            // matrix pipe then idles 20 % of the chunk; measured: 34 us per step of MFMAs + LDS alone against a 24 us floor).
            // A tile's fragments are refilled right after its tile-group (from the next group of the chunk, or -- last group --
            // from chunk q+1, which the barrier behind tile-group TGB has published); the other tiles' MFMAs cover the read.
            constexpr int NW = BF ? 1 : 4, NTG = GPC * MT, G = NTG * NW;     // gaps per chunk
            constexpr int QS = NE;                                           // iteration at whose barrier the previous half-step is signalled
            constexpr int TGB = (GPC - 1) * MT;                              // the barrier follows this tile-group
            constexpr int GB = (TGB + 1) * NW;                               // gaps in front of the barrier
            constexpr int GC = GB > NW ? GB - NW : GB;                       // ... that carry LDS writes: the last tile-group in front of the
                                                                             // barrier carries none, so that lgkmcnt(0) there is met on arrival
            auto gap = [&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                // loads of chunk q+D: spread over all gaps; LDS writes of chunk q+1: over the gaps in front of the barrier
                auto ld1 = [&](auto j_tag) {
                    constexpr int J = decltype(j_tag)::value;
                    if constexpr ((J * G) / NLDL == g) {
                        if constexpr (ld_own) prefetch_piece(HT, std::integral_constant<int, SETL>{}, j_tag, std::integral_constant<int, J + 1>{});
                        else prefetch_piece(HNT, std::integral_constant<int, SETL>{}, j_tag, std::integral_constant<int, J + 1>{});
                    }
                };
                [&]<int... J>(std::integer_sequence<int, J...>) { (ld1(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, NLDL>{});
                auto cm1 = [&](auto j_tag) {
                    constexpr int J = decltype(j_tag)::value;
                    if constexpr ((J * GC) / NLDC == g) {
                        if constexpr (cm_own) commit_piece(HT, std::integral_constant<int, SETC>{}, nst, j_tag, std::integral_constant<int, J + 1>{});
                        else commit_piece(HNT, std::integral_constant<int, SETC>{}, nst, j_tag, std::integral_constant<int, J + 1>{});
                    }
                };
                [&]<int... J>(std::integer_sequence<int, J...>) { (cm1(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, NLDC>{});
            };
            auto tile_group = [&](auto tg_tag) {
                    constexpr int TG = decltype(tg_tag)::value;
                    constexpr int gi = TG / MT, m = TG % MT;
                    const bool on = !RAG || m < act;
                    auto quarter = [&](auto w_tag) {
                        constexpr int W = decltype(w_tag)::value;
                        if (on) mfma_w(std::integral_constant<int, GPC * q + gi>{}, std::integral_constant<int, m>{}, w_tag);
                        gap(std::integral_constant<int, TG * NW + W>{});
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    [&]<int... W>(std::integer_sequence<int, W...>) { (quarter(std::integral_constant<int, W>{}), ...); }(std::make_integer_sequence<int, NW>{});
                    if constexpr (TG == TGB) {
                        // Behind the barriers of iterations 0 .. NE-1: one deferred item each of the PREVIOUS half-step (its
                        // product reached Sred in front of this iteration's barrier at the latest); behind the last of them this
                        // half-step's cell operands are requested (one register set: the previous items have just consumed
                        // theirs).  The previous half-step's write-through stores are then drained and signalled at the barrier
                        // of iteration QS = NE (their round trip to memory is longer than a chunk; the consumers have a whole
                        // half-step of slack).  Younger than those stores are exactly: the loads of iteration NE-1 issued behind
                        // its barrier, the 5 NE cell operands, and the loads of iteration NE issued so far -- NLD + 5 NE in all
                        // (every iteration issues the same loads in the same gaps): a counted wait leaves them in flight.
                        if constexpr (q == QS) {
                            static_assert(QS + D < NT, "the iterations up to QS request chunks of this half-step");
                            if (pub >= 0) pb_wait_vmcnt<(NLD + 5 * NE < 63 ? NLD + 5 * NE : 63)>();
                        }
                        __syncthreads();
                        if constexpr (q == QS) {
                            if (pub >= 0) {
                                if (tid == 0) __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(pub)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                pub = -1;
                            }
                        }
                        if constexpr (q < NE) {
                            if (pend_k >= 0) cell_item(HNT, std::integral_constant<int, (q < NE ? q : 0)>{}, std::false_type{}, kstep(pend_k), true, pend_nr);
                        }
                        if constexpr (q == NE - 1) {
                            if (pend_k >= 0) { pub = pend_k; pend_k = -1; }
#pragma unroll
                            for (int e = 0; e < NE; ++e) fetch(H, s, e); // consumed behind the first barriers of the NEXT half-step
                        }
                    }
                    // refill the tile's fragments
                    if constexpr (gi + 1 < GPC) af[m] = frag_at(cur, m, gi + 1);
                    else if constexpr (cm_own || m < MTN) af[m] = frag_at(nxt, m, 0);
                    if constexpr (!cm_own && TG == NTG - 1 && MTN > MT) { // the next half-step has more tiles than this one
#pragma unroll
                        for (int mm = MT; mm < MTN; ++mm) af[mm] = frag_at(nxt, mm, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... TG>(std::integer_sequence<int, TG...>) { (tile_group(std::integral_constant<int, TG>{}), ...); }(std::make_integer_sequence<int, NTG>{});
            ++n;
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (iter(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NT>{});
        spill_acc(HT); // its items run behind the first barriers of the next half-step (or in the flush below)
        pend_k = k; pend_nr = nr;
    };

    for (int k = k0; k < KN; k += 2) {
        half_step(std::integral_constant<int, 0>{}, k);
        half_step(std::integral_constant<int, 1>{}, k + 1);
    }
    if (pub >= 0) signal_now(pub); // (only when the loop did not run its drain iteration)
    if (pend_k >= 0) { // the last half-step's items: nothing left to hide them behind
        __syncthreads();
        const int sp = kstep(pend_k);
        if (pend_k & 1) [&]<int... E>(std::integer_sequence<int, E...>) { (cell_item(std::integral_constant<int, 1>{}, std::integral_constant<int, E>{}, std::true_type{}, sp, true, pend_nr), ...); }(std::make_integer_sequence<int, NE>{});
        else [&]<int... E>(std::integer_sequence<int, E...>) { (cell_item(std::integral_constant<int, 0>{}, std::integral_constant<int, E>{}, std::true_type{}, sp, true, pend_nr), ...); }(std::make_integer_sequence<int, NE>{});
        signal_now(pend_k);
    }

    // bias gradients of this (layer, row block, unit tile): the row groups' partial sums added in a fixed order
    if (!is_up && a.bias_part) {
        __syncthreads();
        const __amdgpu_buffer_rsrc_t r_bp = pf_rsrc(a.bias_part, (size_t)L * RBn * 4 * R * 4);
        if (tid < QPR) { // thread eq: the RPP threads (erow = 0 .. RPP-1) that own the same unit quad
            pf_f32x4 s4[4] = {pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int r = 0; r < RPP; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) s4[g] += *reinterpret_cast<const pf_f32x4 *>(bsum + (g * NVQA_PF_THREADS + r * QPR + tid) * 4);
            // (write-through stores: the workgroup that adds the row blocks' sums below may sit on another XCD)
            const unsigned doff = (unsigned)((((size_t)l * RBn + rb) * 4 * R + u0 + 4 * tid) * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, s4[g]), r_bp, doff + (unsigned)g * R * 4, 0, 16 /* sc1 */);
        }
        if (a.bias_cnt) {
            // b_i2h and b_h2h of this layer's units u0 .. u0 + UNITS - 1: the row blocks' sums added in row-block order (the order of
            // round 3's k_bias_sum launches: bit-identical) by whichever of the RBn workgroups of this (layer, unit tile) finishes
            // last.  Hand-off as everywhere in this kernel (MI355X_MICROARCH.md "Valid forms", first row of the table): sc1 stores,
            // drained by the storing waves, the workgroup's barrier, ONE agent-scope add by one lane; the workgroup whose add came
            // last -- told by the value returned -- reads all partial sums with sc1 loads behind a barrier its adding wave joins;
            // it re-arms the counter.  (An agent-scope fence pair instead -- __threadfence() -- is an L2 write-back and an invalidate
            // per workgroup on a chip whose L2s hold this launch's dirty dG lines: measured slower than the launches it saves.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *const flag = reinterpret_cast<unsigned *>(ring); // (the ring is dead by now)
            if (tid == 0) *flag = __hip_atomic_fetch_add(a.bias_cnt + l * a.NU + ut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag == (unsigned)RBn - 1) {
                if (tid < QPR) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        pf_f32x4 s4 = pf_f32x4{0.f, 0.f, 0.f, 0.f};
                        for (int r = 0; r < RBn; ++r)
                            s4 += __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bp, (unsigned)((((size_t)l * RBn + r) * 4 * R + (size_t)g * R + u0 + 4 * tid) * 4), 0, 16 /* sc1 */));
                        *reinterpret_cast<pf_f32x4 *>(a.bias_i[l] + (size_t)g * R + u0 + 4 * tid) = s4;
                        *reinterpret_cast<pf_f32x4 *>(a.bias_h[l] + (size_t)g * R + u0 + 4 * tid) = s4;
                    }
                }
                if (tid == 0) __hip_atomic_store(a.bias_cnt + l * a.NU + ut, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 2] = wall_clock64();
}

} // namespace nvqa
