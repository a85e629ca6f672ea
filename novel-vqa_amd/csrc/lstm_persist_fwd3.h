// lstm_persist_fwd3.h -- the persistent forward LSTM kernel with its A operand loaded STRAIGHT INTO THE MFMA FRAGMENT
// REGISTERS (round 4; f32 instances).  Same decomposition into workgroups, buffers, cell arithmetic and hand-off protocol as
// lstm_persist.h (misc/RNNUtils.lua:128-154 driving misc/LSTM.lua:12-73): workgroup = (layer l, row block rb of 16 TILES sorted
// batch rows, 16 hidden units = 64 gate columns), one per CU, weights resident in registers for the whole launch.
//
// What changes is the split of the product over the four waves, and with it how [x_t | h_{t-1}] reaches the matrix cores.  In
// lstm_persist.h wave w owns GATE w over all of K, so every wave needs every A fragment: a 64-deep K chunk travels global ->
// staging registers -> LDS ring -> fragment registers, one workgroup barrier per chunk (16 per step), and the fragment reads,
// LDS writes and barriers are cut into pieces that sit between the MFMAs.  Here wave w owns K-QUARTER w of both segments for
// ALL FOUR gates (the same 256 resident B fragments: 4 gates x 16 groups of 16 k).  No A fragment is shared between waves any
// more, and the fragment a lane feeds to v_mfma_f32_16x16x4_f32 -- 4 consecutive k of row li, k = 16 g + 4 lh + w, the k order
// the resident B fragments already have -- is 16 contiguous bytes of the row: ONE sc1 buffer_load_dwordx4 per lane feeds 16
// MFMAs (4 k x 4 gates).  Each wave streams its own fragments through a ring of PD register sets; there is no LDS ring, no
// commit, no fragment read and no barrier inside a chain-step.  What it costs: the four K-quarter partial tiles are summed
// through LDS (in wave order: deterministic) before the cell -- ONE barrier per chain-step, between the spill and the cell;
// the partial-tile buffer is doubled so that no second one is needed.  (lstm_persist_bwd3.h is the same idea for the BPTT.)
//
// NH = 2 independent row CHAINS per workgroup (the row tiles of a block dealt round-robin, as in the BPTT kernels): a step is
// two chain-steps, counters are per (layer, row block, chain, step), and while one chain's hand-off travels (stores drained,
// counter, the consumers' poll, their loads) the workgroup multiplies the other.  The hand-off signal is deferred: every wave
// drains ITS stores of chain-step k with a counted s_waitcnt PSIG pairs into chain-step k + 1 and counts itself in in LDS; the
// wave that arrives last adds to the counter the consumers poll (MI355X_MICROARCH.md "Valid forms": sc1 stores, counted
// drain, LDS arrival counter, one agent-scope add; consumers poll per wave with sc1 loads and load the bytes with sc1).
//
// The K order per output differs from lstm_persist.h (four partial sums per gate instead of one chain): f32 rounding only.
#pragma once
#include "lstm_persist.h"
#include "lstm_persist_bwd2.h" // pb_wait_vmcnt

namespace nvqa {

constexpr int pf3_mt(int tiles, int nh, int h) { return (tiles - h + nh - 1) / nh; }

template <int MTA, int NH> struct PersistFwd3Geom {
    static constexpr int ROWSH = 16 * MTA;                    // rows of the largest chain
    static constexpr int SROW = 64 + 4;                       // row stride of a partial tile [rows][4 gates x 16 units] (conflict-free spill)
    static constexpr int SRED = 4 * ROWSH * SROW;             // floats of one buffer of partial tiles [4 waves][ROWSH][SROW]
    static constexpr int NE = (ROWSH * 4 + NVQA_PF_THREADS - 1) / NVQA_PF_THREADS; // (row, unit quad) items per thread and chain
    static constexpr int CS_FLOATS = NVQA_PF_THREADS * NH * NE * 4; // carried cell state of the owned (row, unit)s
    static constexpr int BIAS_FLOATS = 64;                    // [4 gates][16 units] b_i2h + b_h2h
    static constexpr size_t LDS_BYTES = (size_t)(2 * SRED + CS_FLOATS + BIAS_FLOATS + 4) * 4;
};

// G0Q / G1Q: K groups (16 k) of the input / recurrent segment PER WAVE (a quarter of the segment, rounded up to whole groups; k
// beyond the quarter carry zero weights); TILES row tiles of 16 rows per workgroup, dealt round-robin to NH chains; PD fragment
// loads in flight per lane.  RAG: ragged arch1 batches (lstm_persist.h): row tiles without active rows skip their MFMAs and loads.
template <int G0Q, int G1Q, int TILES, int NH, int PD, bool RAG>
__device__ __forceinline__ void persist_fwd3_layer(const PersistFwdArgs &a, const int l, const int rb, const int ut, float *smem)
{
    constexpr int MTA = pf3_mt(TILES, NH, 0);
    typedef PersistFwd3Geom<MTA, NH> GE;
    constexpr int ROWSH = GE::ROWSH, SROW = GE::SROW, SRED = GE::SRED, NE = GE::NE, GQ = G0Q + G1Q;
    static_assert(NH == 2 && TILES % NH == 0, "two chains of equal length");
    static_assert(PD <= G0Q * MTA, "the next chain-step's first PD fragments (requested in this one's tail) are input-segment fragments");
    static_assert((GQ * MTA) % PD == 0, "the fragment ring keeps its phase across chain-steps");
    float *const Sred = smem;                                  // [2][4 waves][ROWSH][SROW]: partial tiles of chain-step k in buffer k & 1
    float *const cs = smem + 2 * SRED;                         // [chain][item][thread][4]: carried cell state
    float *const biasL = cs + GE::CS_FLOATS;                   // [4 gates][16 units]
    unsigned *const sigcnt = reinterpret_cast<unsigned *>(biasL + GE::BIAS_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    // every field the pipelined loop touches is copied into a local first (lstm_persist_bwd3.h: the layer-indexed arrays keep the
    // argument block in scratch memory, and a scratch load inside the loop drains the fragment ring)
    const int B = a.B, R = a.R, TS = a.TS, RBn = a.RB, NU = a.NU, dbg = a.dbg;
    const int Kin = l == 0 ? a.E : R;
    unsigned *const cnt = a.cnt, *const errw = a.err;
    const unsigned spin_limit = a.spin_limit;
    const int *const nrows_p = a.nrows;
    const Drop dr = a.dr;
    const bool has_next = l + 1 < a.L;
    const bool top_h0 = a.h0_top && l == a.L - 1; // arch2 NVQA_QUIRK_H0: the top layer's h_{-1} is live at step 0
    const int u0 = ut * 16;
    float *const Gt_l = a.Gt[l], *const Cs_l = a.Cs[l];
    if (tid == 0) *sigcnt = 0u;

    // ---- weights: gate gt, units u0 .. u0+15, K-quarter `wave` of [W_i2h | W_h2h]: resident B fragments -----------------------
    // lane (li, lh) supplies B[k = lh][n = li] of MFMA w of group g: k = quarter base + 16 g + 4 lh + w
    const int q0 = ((Kin / 4 + 3) / 4) * 4;                 // the input segment's quarter (a multiple of 4 floats: 16-byte fragments)
    const int q0n = max(0, min(q0, Kin - wave * q0));        // ... of which this wave's range holds q0n k
    pf_u32x4 bw[4][GQ];
    // loaded in BATCHES of 4 fragments that are pinned in their AGPR quads at once (the MFMAs' B operand, "a" constraint): with all
    // 256 fragments in flight through the VGPRs the allocator spills every value that lives across this prologue and reloads it from
    // scratch at each use -- a vector memory load in the cell section, i.e. a wait that drains the fragment ring once per chain-step
    static_assert(G0Q % 4 == 0 && G1Q % 4 == 0, "whole batches of 4 fragments");
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        const float *wi = a.Wi[l] + (size_t)(gt * R + u0 + li) * Kin + wave * q0;
        const float *wh = a.Wh[l] + (size_t)(gt * R + u0 + li) * R + wave * (R / 4);
#pragma unroll
        for (int g0 = 0; g0 < GQ; g0 += 4) {
            pf_u32x4 raw[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int g = g0 + gg;
                pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (g < G0Q) { if (16 * g + 4 * lh < q0n) v = *reinterpret_cast<const pf_f32x4 *>(wi + 16 * g + 4 * lh); }
                else v = *reinterpret_cast<const pf_f32x4 *>(wh + 16 * (g - G0Q) + 4 * lh);
                raw[gg] = __builtin_bit_cast(pf_u32x4, v);
            }
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                pf_u32x4 q = raw[gg];
                asm volatile("" : "+a"(q));
                bw[gt][g0 + gg] = q;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0); // (nothing that follows is computed above the prologue and kept live across it)
    if (tid < 64) biasL[tid] = a.bi[l][(tid >> 4) * R + u0 + (tid & 15)] + a.bh[l][(tid >> 4) * R + u0 + (tid & 15)];
    if ((dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 1] = wall_clock64();

    // ---- buffers other workgroups write during the launch: sc1 buffer accesses only --------------------------------------------
    const size_t hs_bytes = (size_t)(TS + 1) * B * R * 4, u_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_in = l == 0 ? pf_rsrc(a.X0, (size_t)TS * B * a.E * 4) : pf_rsrc(a.U[l], u_bytes);
    const __amdgpu_buffer_rsrc_t r_h = pf_rsrc(a.Hs[l], hs_bytes);
    const __amdgpu_buffer_rsrc_t r_un = has_next ? pf_rsrc(a.U[l + 1], u_bytes) : r_h;

    // local row i of the block <-> sorted batch row rb + RBn i; rows 0 .. nloc-1 exist.  Row rho of chain h is local row
    // (rho % 16) + 16 (NH (rho / 16) + h): its tile m is tile NH m + h of the block.
    const int nloc = (B - rb + RBn - 1) / RBn;
    // this lane's fragment of (chain h, tile m, group g): 16 bytes at row (rb + RBn (li + 16 (NH m + h))) of the step's slice,
    // element (quarter base) + 16 g + 4 lh: voff + step * step_bytes + 64 g (the 64 g in the load's immediate offset)
    const unsigned step_bytes0 = (unsigned)B * Kin * 4, step_bytes1 = (unsigned)B * R * 4;
    // (one lane base per segment + SCALAR tile and step offsets: the bases of a chain-step's row tiles are recomputed where they
    // are needed, 3 vector instructions each, instead of living in registers -- the layer >= 1 instances have no spare register)
    const unsigned lbase0 = ((unsigned)(rb + RBn * li) * Kin + wave * q0 + 4 * lh) * 4u;
    const unsigned lbase1 = ((unsigned)(rb + RBn * li) * R + wave * (R / 4) + 4 * lh) * 4u;
    const unsigned tstride0 = 16u * RBn * Kin * 4u, tstride1 = 16u * RBn * R * 4u; // one row tile further

    // active steps of this row block: one contiguous range [t_lo, t_hi) (lstm_persist.h)
    int t_lo = 0, t_hi = 0;
    {
        int t = 0;
        while (t < TS && rb >= nrows_p[t]) ++t;
        t_lo = t;
        while (t < TS && rb < nrows_p[t]) ++t;
        t_hi = t;
    }
    t_lo = __builtin_amdgcn_readfirstlane(t_lo);
    t_hi = __builtin_amdgcn_readfirstlane(t_hi);

    // counters: [layer][row block][chain][step]
    const unsigned cown = (unsigned)(((l * RBn + rb) * NH) * TS), cbelow = (unsigned)((((l - 1) * RBn + rb) * NH) * TS);
    auto own_word = [&](int k) __attribute__((always_inline)) { return cnt + cown + (unsigned)((k % NH) * TS + k / NH); };
    auto rec_word = [&](int k) __attribute__((always_inline)) { return cnt + cown + (unsigned)((k % NH) * TS + k / NH - 1); };
    auto in_word = [&](int k) __attribute__((always_inline)) { return cnt + cbelow + (unsigned)((k % NH) * TS + k / NH); };
    // RAG: active tiles of chain h at step t (rows dealt round-robin: the active local rows are a prefix)
    auto act_of = [&](int h, int t) __attribute__((always_inline)) -> int {
        if constexpr (!RAG) return MTA;
        if (dbg & 1024) return MTA; // debugging: nothing skipped, nothing out of range
        const int nr = nrows_p[t < 0 ? 0 : (t >= TS ? TS - 1 : t)];
        const int tiles = ((nr > rb ? (nr - rb + RBn - 1) / RBn : 0) + 15) >> 4;
        return __builtin_amdgcn_readfirstlane(min(MTA, max((tiles - h + NH - 1) / NH, 0)));
    };

    pf_f32x4 acc[MTA][4];
    pf_u32x4 af[PD];
    // hook(i): called behind MFMA i = 4 w + gate of the pair: an LDS or vector-memory instruction to issue in the MFMA's shadow.
    // (Vector ARITHMETIC does not hide there: the f32 MFMA runs at the vector f32 rate and the cell's instructions cost the same
    // cycles between the MFMAs as behind them -- measured in round 4 with the cell cut into 33 and into 59 micro-steps behind the
    // MFMAs of the next chain-step: 2 340 and 2 750 exposed cycles against 2 320 as one block; the LDS writes of the spill do hide.)
    auto mfma_pair = [&](auto g_tag, auto m_tag, const pf_u32x4 &frag, auto &&hook) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, m = decltype(m_tag)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
            ([&] __attribute__((always_inline)) { // (locals: operands named only inside an asm statement are not captured by the lambda)
                constexpr int w = I / 4, gt = I % 4;
                pf_f32x4 &c = acc[m][gt];
                const float av = __builtin_bit_cast(pf_f32x4, frag)[w], bv = __builtin_bit_cast(pf_f32x4, bw[gt][g])[w];
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
                hook(std::integral_constant<int, I>{});
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
    };
    // RAG: the same 16 MFMAs, skipped when row tile m of the chain has no active row (m >= act) -- with the branch INSIDE the asm
    // statement (one per K step w: 4 MFMAs).  A C++ `if (m < act)` around mfma_pair puts a control-flow join behind every pair, and at
    // those joins hipcc moves the 64 accumulator registers between two homes with v_mov_b64 copies it schedules without knowing that
    // the asm statements are MFMAs (the one instruction at fault was not identified): the instance came out wrong in the lower register pair of every accumulator of the row tiles m >= 1
    // (rows = 0, 1 mod 4 of their tile) even when nothing was skipped at run time (tests/dbg_rag_rows.py, NVQA_PF_DBG=1024).  Hidden
    // from the compiler the stream stays the straight-line code of the instance without skips.  Same order of the sums: bit-identical.
    auto mfma_pair_rag = [&](auto g_tag, auto m_tag, const pf_u32x4 &frag, int act) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, m = decltype(m_tag)::value;
        [&]<int... W>(std::integer_sequence<int, W...>) __attribute__((always_inline)) {
            ([&] __attribute__((always_inline)) {
                pf_f32x4 &c0 = acc[m][0], &c1 = acc[m][1], &c2 = acc[m][2], &c3 = acc[m][3]; // (locals: see mfma_pair)
                const int act_ = act;
                const float av = __builtin_bit_cast(pf_f32x4, frag)[W];
                const float b0 = __builtin_bit_cast(pf_f32x4, bw[0][g])[W], b1 = __builtin_bit_cast(pf_f32x4, bw[1][g])[W];
                const float b2 = __builtin_bit_cast(pf_f32x4, bw[2][g])[W], b3 = __builtin_bit_cast(pf_f32x4, bw[3][g])[W];
                asm volatile("s_cmp_gt_i32 %9, %10\n\t"
                             "s_cbranch_scc0 1f\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\t"
                             "v_mfma_f32_16x16x4_f32 %1, %4, %6, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %4, %7, %2\n\t"
                             "v_mfma_f32_16x16x4_f32 %3, %4, %8, %3\n"
                             "1:"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                             : "v"(av), "a"(b0), "a"(b1), "a"(b2), "a"(b3), "s"(act_), "n"(m)
                             : "scc");
            }(), ...);
        }(std::make_integer_sequence<int, 4>{});
    };
    auto touch_acc = [&] {
#pragma unroll
        for (int m = 0; m < MTA; ++m)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) asm volatile("" : "+v"(acc[m][gt]));
    };
    auto nop_after_clear = [&] { touch_acc(); asm volatile("s_nop 7" ::: "memory"); };
    auto nop_before_read = [&] { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); touch_acc(); };

    // epilogue ownership: thread -> (row rho = tid / 4 + 64 e of the chain, units u0 + 4 eq .. +3)
    // The 16 quads of a wave are PERMUTED over its 16 rows: a ds_read_b128 is serviced in four fixed groups of 16 lanes (quads {0,3,5,6},
    // {1,2,4,7}, {8,11,13,14}, {9,10,12,15}: MI355X_MICROARCH.md "LDS"), a row of the partial tile starts 4 banks after the row
    // before it (SROW = 68), and a quad's four lanes read 16 consecutive banks -- so the quads of a group must sit on rows 4 apart
    // ({a, a+4, a+8, a+12}: banks 0-15, 16-31, 32-47, 48-63) for the cell's 16 reads of partial tiles to be conflict-free (in lane
    // order, rows 0, 3, 5, 6 of a group overlap 2- and 3-fold).  Which row a thread owns is free: stores stay 64 contiguous bytes per quad.
    const int eq = tid & 3, erow = (tid >> 6) * 16 + (int)((0xFEAB6732DC894510ull >> (4 * ((tid >> 2) & 15))) & 15);
    auto eloc = [&](int h, int e) __attribute__((always_inline)) { const int rho = erow + 64 * e; return (rho & 15) + 16 * (NH * (rho >> 4) + h); };
    int esi[NH][NE]; // original batch row of the owned rows (the dropout stream is indexed by it)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iloc = eloc(h, e), grow = rb + RBn * iloc;
            esi[h][e] = (erow + 64 * e < ROWSH && grow < B) ? a.sort_idx[grow] : 0;
            *reinterpret_cast<pf_f32x4 *>(cs + ((h * NE + e) * NVQA_PF_THREADS + tid) * 4) = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const __amdgpu_buffer_rsrc_t r_gt = pf_rsrc(Gt_l, (size_t)TS * B * 4 * R * 4), r_cs = pf_rsrc(Cs_l, hs_bytes);
    // fused cell of chain h at step t (act = false: the row block has not started / has stopped: zeros), stores
    auto cell_item = [&](auto h_tag, auto e_tag, int t, bool act, int nr, int buf) __attribute__((always_inline)) {
        constexpr int h = decltype(h_tag)::value, e = decltype(e_tag)::value;
        const int rho = erow + 64 * e, iloc = eloc(h, e), grow = rb + RBn * iloc;
        const float *const Sr = Sred + buf * SRED + rho * SROW + 4 * eq; // (one base + compile-time offsets, as for the spill)
        if (rho >= ROWSH || grow >= B || (dbg & 2)) return;
        const bool on = act && grow < nr;
        pf_f32x4 gi = {0.f, 0.f, 0.f, 0.f}, gf = gi, go = gi, gg = gi, cn = gi, hn = gi, un = gi;
        pf_f32x4 *cst = reinterpret_cast<pf_f32x4 *>(cs + ((h * NE + e) * NVQA_PF_THREADS + tid) * 4);
        if (on) {
            // ALL 21 LDS reads first, then the sums: written as p[g] = S[..]; p[g] += S[..]; ... hipcc issues one read at a time and
            // waits for it (lgkmcnt(0)) before the add -- 16 exposed LDS round trips, 1 900 of the cell's 2 250 cycles (r4 stamps)
            pf_f32x4 q[4][4], bq[4], p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int w = 0; w < 4; ++w) q[g][w] = *reinterpret_cast<const pf_f32x4 *>(&Sr[w * ROWSH * SROW + 16 * g]);
                bq[g] = *reinterpret_cast<const pf_f32x4 *>(&biasL[4 * eq + 16 * g]);
            }
            const pf_f32x4 cp = *cst;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) p[g] = ((q[g][0] + q[g][1]) + q[g][2]) + q[g][3] + bq[g]; // K-quarters in wave order, then the bias
            const uint64_t didx = ((((uint64_t)l) * B + esi[h][e]) * TS + t) * R + u0 + 4 * eq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gi[j] = pf_sigmoid(p[0][j]);
                gf[j] = pf_sigmoid(p[1][j]);
                go[j] = pf_sigmoid(p[2][j]);
                gg[j] = pf_tanh(p[3][j]);
                cn[j] = gf[j] * cp[j] + gi[j] * gg[j];
                hn[j] = go[j] * pf_tanh(cn[j]);
            }
            if (has_next) {
#pragma unroll
                for (int j = 0; j < 4; ++j) un[j] = dr.scale(NVQA_SITE_LSTM, didx + j) * hn[j];
            }
        }
        *cst = cn;
        // 32-bit byte offsets (every buffer is < 4 GB): the 64-bit pointer arithmetic of plain stores was a fifth of the cell's instructions
        const unsigned srow_g = (unsigned)t * B + grow, q4 = (unsigned)(u0 + 4 * eq);
        const unsigned go_ = (srow_g * 4u * R + q4) * 4u, uo = (srow_g * R + q4) * 4u, so = uo + (unsigned)B * R * 4u; // so: slice t + 1 of Hs / Cs
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, gi), r_gt, go_, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, gf), r_gt, go_ + (unsigned)R * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, go), r_gt, go_ + 2u * R * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, gg), r_gt, go_ + 3u * R * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, cn), r_cs, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, hn), r_h, so, 0, 16 /* sc1 */);
        if (has_next) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, un), r_un, uo, 0, 16 /* sc1 */);
    };
    auto cell_all = [&](auto h_tag, int t, bool act, int nr, int buf) __attribute__((always_inline)) {
        [&]<int... E>(std::integer_sequence<int, E...>) __attribute__((always_inline)) { (cell_item(h_tag, std::integral_constant<int, E>{}, t, act, nr, buf), ...); }(std::make_integer_sequence<int, NE>{});
    };
    auto signal_now = [&](int k) __attribute__((always_inline)) { // not deferred: drain, barrier, one add
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(k)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto signal_wave = [&](int k) __attribute__((always_inline)) { // deferred: the wave has drained its stores; the last of the four adds
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(sigcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && (old & 3u) == 3u)
            __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(k)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    auto idle_step = [&](int t) __attribute__((always_inline)) { // a step outside the row block's active range: zeros, signalled at once
        [&]<int... Hh>(std::integer_sequence<int, Hh...>) __attribute__((always_inline)) {
            ((cell_all(std::integral_constant<int, Hh>{}, t, false, 0, 0), signal_now(NH * t + Hh)), ...);
        }(std::make_integer_sequence<int, NH>{});
    };

    __syncthreads(); // sigcnt, bias, the carried state
    // ---- steps before the row block starts: zeros, signalled at once ---------------------------------------------------------------
    for (int t = 0; t < t_lo; ++t) idle_step(t);

    if (t_lo < t_hi) {
        const int K0 = NH * t_lo, KN = NH * t_hi; // chain-steps k = NH t + h
        const unsigned dbg_oob = __builtin_amdgcn_readfirstlane((dbg & 8) ? PF_OOB : 0u);
        // base offsets of the lane's rows for chain-step (h, t): input segment / recurrent segment (slice t of Hs = h_{t-1});
        // tile m of chain h is tile NH m + h of the block; rows beyond the batch, disabled loads and (RAG) inactive tiles go out of range
        auto bases0 = [&](auto h_tag, int t, int act, bool en, unsigned (&b)[MTA]) __attribute__((always_inline)) {
            constexpr int H = decltype(h_tag)::value;
            const unsigned enm = __builtin_amdgcn_readfirstlane(en ? 0u : PF_OOB) | dbg_oob;
#pragma unroll
            for (int m = 0; m < MTA; ++m) {
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)t * step_bytes0 + (unsigned)(NH * m + H) * tstride0));
                unsigned o = li < nloc - 16 * (NH * m + H) ? (lbase0 + so) | enm : PF_OOB;
                if constexpr (RAG) o = m < act ? o : PF_OOB;
                b[m] = o;
            }
        };
        auto bases1 = [&](auto h_tag, int t, int act, unsigned (&b)[MTA]) __attribute__((always_inline)) {
            constexpr int H = decltype(h_tag)::value;
#pragma unroll
            for (int m = 0; m < MTA; ++m) {
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)t * step_bytes1 + (unsigned)(NH * m + H) * tstride1));
                unsigned o = li < nloc - 16 * (NH * m + H) ? (lbase1 + so) | dbg_oob : PF_OOB;
                if constexpr (RAG) o = m < act ? o : PF_OOB;
                b[m] = o;
            }
        };
        auto ld0 = [&](unsigned base, auto g_tag) __attribute__((always_inline)) -> pf_u32x4 {
            return __builtin_amdgcn_raw_buffer_load_b128(r_in, base + 64u * decltype(g_tag)::value, 0, 16 /* sc1 */);
        };
        auto ld1 = [&](unsigned base, auto g_tag) __attribute__((always_inline)) -> pf_u32x4 {
            return __builtin_amdgcn_raw_buffer_load_b128(r_h, base + 64u * decltype(g_tag)::value, 0, 16 /* sc1 */);
        };
        unsigned bb[MTA]; // row bases of the fragments being requested: this chain-step's input rows, then its recurrent rows, then the next chain-step's input rows
        unsigned pend_in = 0, pend_rec = 0;
        int pub = -1; // chain-step whose stores are issued but not yet drained and signalled
        // pipeline prologue for chain-step K0 (chain 0): its input rows' producers, the first PD fragments
        if (l > 0 && !(dbg & 1)) (void)pf_wait_ge(in_word(K0), (unsigned)NU, errw, 0x100u + l, spin_limit);
        bases0(std::integral_constant<int, 0>{}, t_lo, act_of(0, t_lo), true, bb);
        [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) {
            ((af[Pp] = ld0(bb[Pp % MTA], std::integral_constant<int, Pp / MTA>{})), ...);
        }(std::make_integer_sequence<int, PD>{});

        // the layer-0 instance of a model whose layers differ does not multiply the recurrent groups of its first active step
        // (h_{-1} = 0; lstm_persist.h SKIP0: every layer above starts that much earlier)
        constexpr bool SKIP0 = G0Q != G1Q;

        unsigned long long tm_st = 0, tm_sp = 0, tm_ba = 0, tm_ce = 0; // NVQA_PF_DBG & 128: shader cycles in the stream / last spill / barrier / cell
        auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long { return (dbg & 128) ? __builtin_amdgcn_s_memtime() : 0ull; };
        auto chain_step = [&](auto h_tag, int k) __attribute__((always_inline)) {
            constexpr int H = decltype(h_tag)::value, HN = (H + 1) % NH;
            constexpr int MT = MTA, P = GQ * MT, PR = G0Q * MT; // pairs of the chain-step; first recurrent pair
            // where the housekeeping sits in the stream of P pairs (one fragment load per pair, all unconditional)
            constexpr int PSIG = 16;                       // drain + signal of the PREVIOUS chain-step's stores
            constexpr int PRC = PR - PD;                   // the first recurrent fragment is requested by this pair: its producers must be done
            constexpr int PRQ = PRC >= 6 ? PRC - 6 : -1;   // ... their counter is requested here (-1: in the previous chain-step's tail)
            constexpr int PNRQ = P - 3;                    // (PRQ < 0) the next chain-step's recurrent counter is requested
            constexpr int PNIQ = P - PD - 6;               // the counter of the next chain-step's input rows is requested (looked at 6 pairs on)
            static_assert(PNIQ >= 0 && PSIG < P, "housekeeping points lie in the chain-step");
            // memory operations YOUNGER than the previous chain-step's stores at the signal point: one fragment per pair, the counters
            constexpr int NYOUNG = PSIG + (PRQ >= 0 && PRQ < PSIG ? 1 : 0) + (PNIQ < PSIG ? 1 : 0) + (PRQ < 0 && PNRQ < PSIG ? 1 : 0);
            static_assert(NYOUNG <= 63, "vmcnt is a 6-bit field");
            const int t = k / NH;
            const bool morel = k + 1 < KN;
            const int kl = morel ? k + 1 : k, tl = kl / NH;
            const int act = act_of(H, t), actn = act_of(HN, tl);
            const int nr = nrows_p[t];
            // h_{t-1} is other workgroups' bytes at every step but 0 (slice 0 is the constant initial state): at the row block's FIRST
            // active step t_lo > 0 the slice was written -- as zeros -- by the idle steps of all NU unit-tile workgroups, and the layers
            // without SKIP0 multiply it: they must have seen those zeros, not the previous batch's h (lstm_persist.h waits there too)
            const bool rec_dep = t > 0;
            const bool skip_rec = SKIP0 && __builtin_amdgcn_readfirstlane((t == t_lo && !top_h0) ? 1 : 0) != 0;
#pragma unroll
            for (int m = 0; m < MTA; ++m)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) acc[m][gt] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            nop_after_clear();
            // this lane's corner of the partial tile of chain-step k.  ONE lane base + compile-time offsets (they fit the ds_write offset
            // field): written as Sred[(... + 16 m + r) * SROW + ...] hipcc keeps all 64 addresses in registers across the stream -- and
            // spills them in the layer >= 1 instances, which have no spare AGPR
            float *const Sw = Sred + (k & 1) * SRED + (wave * ROWSH + 4 * lh) * SROW + li;
            const unsigned long long tm0 = stamp();
            if ((dbg & 256) && l == 0) { // measurement / tests: a slow layer 0, so that the layers above really wait at their counters
                for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(127);
            }
            __builtin_amdgcn_sched_barrier(0);

            auto pair = [&](auto p_tag) __attribute__((always_inline)) {
                constexpr int p = decltype(p_tag)::value, g = p / MT, m = p % MT, slot = p % PD;
                if constexpr (p == PSIG) {
                    if (pub >= 0) {
                        if (dbg & 2048) pb_wait_vmcnt<0>(); // debugging: full drain instead of the counted one
                        else pb_wait_vmcnt<NYOUNG>();
                        signal_wave(pub);
                        pub = -1;
                    }
                }
                if constexpr (p == PRQ) {
                    pend_rec = __hip_atomic_load(rec_dep ? rec_word(k) : cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::: "memory"); // pinned here (lstm_persist_bwd3.h: left alone hipcc sinks it to its use and drains the ring there)
                }
                if constexpr (p == (PRC > 0 ? PRC : 0)) { // the recurrent fragments are requested from here on
                    if (rec_dep && !(dbg & 1) && pend_rec < (unsigned)NU) (void)pf_wait_ge(rec_word(k), (unsigned)NU, errw, 0x200u + l, spin_limit);
                    bases1(h_tag, t, act, bb);
                }
                if constexpr (p == PNIQ) {
                    pend_in = __hip_atomic_load(morel && l > 0 ? in_word(kl) : cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::: "memory");
                }
                if constexpr (p == P - PD) { // the next chain-step's first fragments are requested below
                    if (morel && l > 0 && !(dbg & 1) && pend_in < (unsigned)NU) (void)pf_wait_ge(in_word(kl), (unsigned)NU, errw, 0x100u + l, spin_limit);
                    bases0(std::integral_constant<int, HN>{}, tl, actn, morel, bb);
                }
                if constexpr (PRQ < 0 && p == PNRQ) { // the next chain-step looks at it before its first pair
                    pend_rec = __hip_atomic_load(morel && tl > 0 ? rec_word(kl) : cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::: "memory");
                }
                auto hook = [&](auto i_tag) __attribute__((always_inline)) {
                    if constexpr (p >= P - (MT - 1)) {
                        // the accumulators of row tile mq = p - (P - MT) - 1 took their last MFMA a pair ago: their 16 registers go to the
                        // partial-tile buffer now, one ds_write behind each MFMA (as one block of 64 behind the stream: 850 cycles, now 290)
                        constexpr int i = decltype(i_tag)::value, mq = p - (P - MT) - 1, gt = i / 4, r = i % 4;
                        Sw[(16 * mq + r) * SROW + 16 * gt] = acc[mq][gt][r];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (!(SKIP0 && g >= G0Q && skip_rec)) {
                    // RAG: a pair of a row tile without active rows is skipped inside mfma_pair_rag; the last MT - 1 pairs, whose MFMAs
                    // carry the spill of the tile before them, always multiply (an inactive tile's fragments are zeros: out-of-range loads)
                    if constexpr (RAG && p < P - (MT - 1)) mfma_pair_rag(std::integral_constant<int, g>{}, std::integral_constant<int, m>{}, af[slot], act);
                    else mfma_pair(std::integral_constant<int, g>{}, std::integral_constant<int, m>{}, af[slot], hook);
                    // SKIP0: this `if` is a control-flow join, and behind the pairs that carry a spill hipcc parks the accumulators in a
                    // second register set there -- v_mov_b64 copies that start 8 wait states behind the pair's last MFMA and read its
                    // destination (tools/mfma_hazard_scan.py; an 8-pass MFMA needs 11): wait the difference out
                    if constexpr (SKIP0 && p >= P - (MT - 1)) { asm volatile("s_nop 4" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                } else { // (no MFMAs at this pair: the hooks alone)
                    // ... and then nothing separates the spill of row tile mq from that tile's last MFMAs, one pair back: an MFMA result
                    // needs its wait states before a ds_write reads it as data
                    if constexpr (p >= P - (MT - 1)) nop_before_read();
                    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (hook(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, 16>{});
                }
                // the fragment PD pairs ahead takes the slot just consumed
                if constexpr (p + PD < P) {
                    constexpr int gq = (p + PD) / MT, mq = (p + PD) % MT;
                    if constexpr (gq < G0Q) af[slot] = ld0(bb[mq], std::integral_constant<int, gq>{});
                    else af[slot] = ld1(bb[mq], std::integral_constant<int, gq - G0Q>{});
                } else {
                    constexpr int qn = p + PD - P;
                    af[slot] = ld0(bb[qn % MT], std::integral_constant<int, qn / MT>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) { (pair(std::integral_constant<int, Pp>{}), ...); }(std::make_integer_sequence<int, P>{});

            const unsigned long long tm1 = stamp();
            nop_before_read();
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) // the last row tile's accumulators (the others went out under the last pairs' MFMAs)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sw[(16 * (MT - 1) + r) * SROW + 16 * gt] = acc[MT - 1][gt][r];
            const unsigned long long tm2 = stamp();
            __syncthreads();
            const unsigned long long tm3 = stamp();
            cell_all(h_tag, t, true, nr, k & 1);
            const unsigned long long tm4 = stamp();
            tm_st += tm1 - tm0; tm_sp += tm2 - tm1; tm_ba += tm3 - tm2; tm_ce += tm4 - tm3;
            pub = k; // drained and signalled PSIG pairs into the next chain-step
            if (dbg & 512) { signal_now(k); pub = -1; } // debugging: no deferred signal
        };

        for (int k = K0; k < KN; k += NH) {
            [&]<int... Hh>(std::integer_sequence<int, Hh...>) __attribute__((always_inline)) {
                (chain_step(std::integral_constant<int, Hh>{}, k + Hh), ...);
            }(std::make_integer_sequence<int, NH>{});
        }
        if (pub >= 0) signal_now(pub);
        if ((dbg & 128) && tid == 0) { a.ts[blockIdx.x * 4] = tm_st; a.ts[blockIdx.x * 4 + 1] = tm_sp; a.ts[blockIdx.x * 4 + 2] = tm_ba; a.ts[blockIdx.x * 4 + 3] = tm_ce; }
    }
    // ---- steps after the row block has stopped (arch2: t >= tmax) -----------------------------------------------------------------
    for (int t = t_hi; t < TS; ++t) idle_step(t);
}

// KA: K of layer 0's input; workgroup id -> (layer, row block, unit tile) as in k_lstm_fwd_persist
template <int KA, int KR, int TILES, int PD, bool RAG>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_fwd_persist3(PersistFwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pf_smem[];
    constexpr int Q0 = ((KA / 4 + 3) / 4) * 4, G0A = (Q0 + 15) / 16, GR = KR / 64;
    const int groups = a.L * a.RB;
    const int grp = blockIdx.x % groups, ut = blockIdx.x / groups;
    const int l = grp / a.RB, rb = grp % a.RB;
    if ((a.dbg & 32) && threadIdx.x == 0) a.ts[blockIdx.x * 4] = wall_clock64();
    if (l == 0) persist_fwd3_layer<G0A, GR, TILES, 2, PD, RAG>(a, l, rb, ut, pf_smem);
    else persist_fwd3_layer<GR, GR, TILES, 2, PD, RAG>(a, l, rb, ut, pf_smem);
    if ((a.dbg & 32) && threadIdx.x == 0) a.ts[blockIdx.x * 4 + 3] = (wall_clock64() << 4) | (unsigned long long)(l + 1); // step loop done; layer
    if (l == 0 && a.fr_on) { // the riding product (PersistFwdArgs::fr): tiles dealt round-robin to the layer-0 workgroups
        __syncthreads();
        const int me = ut * a.RB + rb, n = a.RB * a.NU, tx = a.fr.tx, total = tx * a.fr.ty;
        for (int t = me; t < total; t += n) {
            gemm_f32_body<CfgRide, A_KC, B_KC, false, EpiStore, 0, true>(a.fr.g, a.fr.e, t % tx, t / tx, 0, pf_smem);
            __syncthreads();
        }
    }
    if ((a.dbg & 32) && threadIdx.x == 0) a.ts[blockIdx.x * 4 + 2] = wall_clock64();
}
template <int TILES> constexpr size_t persist_fwd3_lds() { return PersistFwd3Geom<pf3_mt(TILES, 2, 0), 2>::LDS_BYTES; }

} // namespace nvqa
