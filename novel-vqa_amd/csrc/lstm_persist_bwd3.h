// lstm_persist_bwd3.h -- the persistent BPTT kernel with its A operand loaded STRAIGHT INTO THE MFMA FRAGMENT REGISTERS
// (round 4).  Same decomposition, roles, counters, hand-off protocol, cell backward and arithmetic as lstm_persist_bwd2.h
// (misc/RNNUtils.lua:182-209 + the nngraph backward of misc/LSTM.lua:41-59): workgroup = (role REC(l) | UP(l), row block,
// 16 NTN hidden units), the K-quarter of gate w on wave w, weights resident in the AGPRs, two independent row chains
// (halves) per workgroup, the four partial tiles summed through LDS in wave order -- every accumulator sees the same MFMAs
// in the same order as in that kernel, so the gradients are BIT-IDENTICAL to it.
//
// What changes is how dG reaches the matrix cores.  In lstm_persist_bwd2.h a chunk of 64 k travels global -> staging
// registers -> LDS ring -> fragment registers, one workgroup barrier per chunk: 8 (bf16) or 16 (f32) barriers and LDS round
// trips per half-step.  But wave w only ever multiplies K-quarter w: NOTHING in that ring is shared between waves -- it is a
// transposition buffer for coalescing.  The fragment a lane feeds to v_mfma_f32_16x16x4_f32 (4 consecutive k of row li:
// k = 16 g + 4 lh + w, the k order the resident B fragments already have) or to v_mfma_f32_16x16x32_bf16 (8 consecutive
// bf16 k of row li) is 16 contiguous bytes of dG's row: ONE buffer_load_dwordx4 per lane, 64 contiguous bytes per row and
// instruction.  So each wave streams its own fragments through a ring of PD registers sets (PD loads in flight per lane, the
// group offset in the instruction's immediate field), there is no LDS ring, no commit, no fragment read and no barrier inside
// a half-step: ONE barrier per half-step remains (between the partial tiles' spill and the cell backward), the partial-tile
// buffer is doubled so that no second one is needed, and the hand-off signal (every storing wave drains with a counted
// s_waitcnt, then the workgroup's ONE agent-scope add) goes through an LDS arrival counter instead of a barrier
// (MI355X_MICROARCH.md "Valid forms", Consumer bullet condition (3), second alternative).
//
// NH independent row CHAINS per workgroup (lstm_persist_bwd2.h has two "halves"): the row tiles of a block are dealt to the
// chains round-robin, a step is NH chain-steps, and while one chain's hand-off travels (stores drained, counter, the consumers'
// poll, their loads: ~5 us) the workgroup multiplies the others.  f32 is MFMA-bound -- a chain-step of 3 or 4 row tiles is
// 13-17 us of MFMAs -- and two chains hide the hand-off; in bf16 a chain-step is ~2 us, so the 4 row tiles of a block run as
// FOUR chains of one tile.
// Measured (B = 512, T = 26, R = 512): DESIGN.md section 4.6.
#pragma once
#include "lstm_persist_bwd2.h"

namespace nvqa {

template <int MTA, int NTN, int NH = 2> struct PersistBwd3Geom {
    static constexpr int ROWSH = 16 * MTA, UNITS = 16 * NTN;
    static constexpr int QPR = UNITS / 4, RPP = NVQA_PF_THREADS / QPR, NE = (ROWSH + RPP - 1) / RPP; // epilogue items per thread and half
    static constexpr int SROW = UNITS + 4;                     // row stride of a partial tile (lstm_persist_bwd2.h: conflict-free spill)
    static constexpr int SRED = 4 * ROWSH * SROW;              // floats of one buffer of partial tiles [4 waves][ROWSH][SROW]
    static constexpr int BSUM_FLOATS = NVQA_PF_THREADS * 16;
    static constexpr int DC_FLOATS = NVQA_PF_THREADS * NH * NE * 4 * 2;
    static constexpr size_t LDS_BYTES = (size_t)(2 * SRED + BSUM_FLOATS + DC_FLOATS + 4) * 4;
};

// row tiles of chain h when TILES tiles are dealt round-robin to NH chains
constexpr int pb3_mt(int tiles, int nh, int h) { return (tiles - h + nh - 1) / nh; }
constexpr int pb3_pairs_before(int gkt, int tiles, int nh, int h) { int n = 0; for (int j = 0; j < h; ++j) n += gkt * pb3_mt(tiles, nh, j); return n; }

// GKT: K groups per gate in all (R / 16 in f32, R / 32 in bf16); TILES: row tiles of 16 rows per row block, dealt round-robin to NH
// chains (chain h: pb3_mt(TILES, NH, h) of them); NTN column tiles of 16 units; PD: fragment loads in flight per lane (ring of PD x 4 registers)
template <int GKT, int TILES, int NH, int NTN, int PD, bool BF, bool RAG>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_bwd_persist3(PersistBwd2Args a)
{
    constexpr int MTA = pb3_mt(TILES, NH, 0);       // the largest chain
    typedef PersistBwd3Geom<MTA, NTN, NH> GE;
    static_assert(NH >= 2 && TILES >= NH, "every chain has a row tile; one chain alone cannot hide its own hand-off");
    constexpr int ROWSH = GE::ROWSH, UNITS = GE::UNITS, SROW = GE::SROW, SRED = GE::SRED;
    constexpr int ES = BF ? 2 : 4;                  // bytes per A element
    // (K group, row tile) pairs of a chain-step of chain h: GKT * pb3_mt(h); the fragment ring keeps its phase across a whole step
    static_assert(pb3_pairs_before(GKT, TILES, NH, NH) % PD == 0, "the fragment ring keeps its phase across a whole step");
    // f32: PD fragments in flight, the next chain-step's first PD ride in the current one's tail (PD <= pairs of the smallest chain-step).
    // bf16: the ring holds TWO whole chain-steps (all chains alike: PD = 2 P) and chain-step k + 2's fragments are requested behind
    // chain-step k's stream, a whole chain-step ahead of their use.
    static_assert(BF ? (TILES % NH == 0 && PD == 2 * GKT * (TILES / NH)) : PD <= GKT * pb3_mt(TILES, NH, NH - 1), "ring depth");
    extern __shared__ __attribute__((aligned(16))) float pb3_smem[];
    float *const Sred = pb3_smem;                   // [2][4 waves][ROWSH][SROW] partial tiles of half-step k in buffer k & 1
    float *const bsum = pb3_smem + 2 * SRED;        // [4 gates][thread][4 units] (lstm_persist_bwd2.h)
    float *const dcs = bsum + GE::BSUM_FLOATS;      // [chain][item][2][thread][4 units]: carried cell gradient, carried cell state
    unsigned *const sigcnt = reinterpret_cast<unsigned *>(dcs + GE::DC_FLOATS); // arrivals of the waves at a hand-off signal
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int B = a.B, R = a.R, TS = a.TS, L = a.L, RBn = a.RB;
    // Every field of the argument block that the pipelined loop touches is copied into a local FIRST: the layer-indexed arrays
    // (a.Gt[la], a.Wi[l + 1], ...) make hipcc keep the whole block in scratch memory, and a scratch load inside the loop is a
    // vector memory operation like any other -- loads return in order, so waiting for it drains the fragment ring.
    const int NU = a.NU, dbg = a.dbg;
    unsigned *const cnt_rec = a.cnt_rec, *const cnt_up = a.cnt_up, *const errw = a.err;
    const unsigned spin_limit = a.spin_limit;
    const int *const nrows_p = a.nrows;
    const Drop dr = a.dr;

    // workgroup -> (role, row block, unit tile): as in lstm_persist_bwd2.h
    const int ngroups = (2 * L - 1) * RBn, gpx = 32 / a.NU > 0 ? 32 / a.NU : 1;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int gslot = xcd * gpx + slot / a.NU, ut = slot % a.NU;
    if (slot / a.NU >= gpx || gslot >= ngroups) {
        if (a.jobs && slot / a.NU < gpx) ride_jobs_run<BF>(a.jobs, (gslot - ngroups) * a.NU + ut, (8 * gpx - ngroups) * a.NU, pb3_smem);
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
    if (tid == 0) *sigcnt = 0u;

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
    if ((dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 1] = wall_clock64();

    const size_t gt_bytes = (size_t)TS * B * 4 * R * 4, pup_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_a = BF ? pf_rsrc(a.Gb[la], gt_bytes / 2) : pf_rsrc(a.Gt[la], gt_bytes); // A operand: dG of layer la
    const __amdgpu_buffer_rsrc_t r_gb = BF ? pf_rsrc(a.Gb[l], gt_bytes / 2) : r_a;   // REC, bf16: the image of dG it writes
    const __amdgpu_buffer_rsrc_t r_g = pf_rsrc(a.Gt[l], gt_bytes);                   // REC: gates in / dG out
    const __amdgpu_buffer_rsrc_t r_p = pf_rsrc(l + 1 < L ? a.Pup[l] : a.Gt[l], l + 1 < L ? pup_bytes : gt_bytes);

    // local row i of the block <-> sorted batch row rb + RBn i; rows 0 .. nloc-1 exist.  Row rho of chain h is local row
    // (rho % 16) + 16 (NH (rho / 16) + h): its tile m is tile NH m + h of the block.
    const int nloc = (B - rb + RBn - 1) / RBn;
    // this lane's fragment of (chain h, row tile m, K group g): 16 bytes at
    //   row (rb + RBn (li + 16 (NH m + h))) of slice sa, element wave * R + KG g + (KG / 4) lh
    // = voff[h][m] + sa * step_bytes + 64 g  (the 64 g goes into the load's immediate offset field)
    const unsigned row_bytes = 4u * R * ES, step_bytes = (unsigned)B * row_bytes;
    unsigned voff[NH][MTA];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int m = 0; m < MTA; ++m) {
            const int iloc = li + 16 * (NH * m + h);
            const bool ok = m < pb3_mt(TILES, NH, h) && iloc < nloc;
            voff[h][m] = ok ? (unsigned)(rb + RBn * iloc) * row_bytes + (unsigned)(wave * R + (KG / 4) * lh) * ES : PF_OOB;
        }
    // chain-step k = NH (TS-1-s) + h
    auto kstep = [&](int k) __attribute__((always_inline)) { return TS - 1 - k / NH; };
    // A slice of half-step k: REC: dG^l_{s+1}; UP: dG^{l+1}_s
    auto ksa = [&](int k) __attribute__((always_inline)) { return is_up ? kstep(k) : kstep(k) + 1; };

    pf_f32x4 acc[MTA][NTN];
    pf_u32x4 af[PD];
    auto mfma_pair = [&](auto g_tag, auto m_tag, const pf_u32x4 &frag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value, m = decltype(m_tag)::value;
        if constexpr (!BF) {
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) { // (locals: operands named only inside an asm statement are not captured by the lambda)
                    pf_f32x4 &c = acc[m][nt];
                    const float av = __builtin_bit_cast(pf_f32x4, frag)[w], bv = __builtin_bit_cast(pf_f32x4, bw[nt][g])[w];
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                pf_f32x4 &c = acc[m][nt];
                const pf_u32x4 av = frag, bv = bw[nt][g];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(bv));
            }
        }
    };
    // wait states between the VALU writes that clear an accumulator and its first MFMA, and between the last MFMA and the first
    // read of an accumulator: the accumulators pass THROUGH empty asm statements (lstm_persist_bwd2.h)
    auto pb_touch_acc = [&] {
#pragma unroll
        for (int m = 0; m < MTA; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) asm volatile("" : "+v"(acc[m][nt]));
    };
    auto pb_nop_after_clear = [&] { pb_touch_acc(); asm volatile("s_nop 7" ::: "memory"); };
    auto pb_nop_before_read = [&] { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); pb_touch_acc(); };

    // epilogue ownership: thread -> (row rho = erow + RPP e of the half, units u0 + 4 eq .. +3)
    constexpr int QPR = GE::QPR, RPP = GE::RPP, NE = GE::NE;
    static_assert(RPP % 16 == 0, "an epilogue pass covers whole row tiles");
    const int eq = tid % QPR, erow = tid / QPR;
    auto eloc = [&](int h, int e) __attribute__((always_inline)) { const int rho = erow + RPP * e; return (rho & 15) + 16 * (NH * (rho >> 4) + h); };
    int esi[NH][NE];      // original batch row of the owned rows (indexes the dropout stream)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iloc = eloc(h, e), grow = rb + RBn * iloc;
            const bool ok = erow + RPP * e < 16 * pb3_mt(TILES, NH, h) && iloc < nloc;
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

    // counters (lstm_persist_bwd2.h)
    // counters: lstm_persist_bwd2.h's layout with NH chains in place of its two halves
    const unsigned crec = (unsigned)(((l * RBn + rb) * NH) * TS);
    const unsigned cneed = (unsigned)(((la * RBn + rb) * NH) * TS);
    const unsigned cup = (unsigned)((((l * RBn + rb) * NH) * NU) * TS);
    auto need_word = [&](int k) __attribute__((always_inline)) { return cnt_rec + cneed + (unsigned)((k % NH) * TS + ksa(k)); };
    auto own_word = [&](int k) __attribute__((always_inline)) {
        return is_up ? cnt_up + cup + (unsigned)(((k % NH) * NU + ut) * TS + kstep(k)) : cnt_rec + crec + (unsigned)((k % NH) * TS + kstep(k));
    };
    auto up_word = [&](int k) __attribute__((always_inline)) { return cnt_up + cup + (unsigned)(((k % NH) * NU + ut) * TS + kstep(k)); };
    // RAG: active tiles of chain h at step s (rows dealt round-robin: the active local rows are a prefix)
    auto act_of = [&](int h, int s) __attribute__((always_inline)) -> int {
        const int mth = pb3_mt(TILES, NH, h);
        if constexpr (!RAG) return mth;
        const int nr = nrows_p[s < 0 ? 0 : (s >= TS ? TS - 1 : s)];
        const int tiles = ((nr > rb ? (nr - rb + RBn - 1) / RBn : 0) + 15) >> 4; // local tiles with active rows
        const int t = (tiles - h + NH - 1) / NH;
        return __builtin_amdgcn_readfirstlane(min(mth, max(t, 0)));
    };

    // cell-backward operands of one item (own gates and cell states of the forward pass: default-policy loads, issued on EVERY
    // path -- the UP role gets out-of-range offsets -- so that no load sits under a runtime branch)
    const __amdgpu_buffer_rsrc_t r_cs = pf_rsrc(a.Cs[l], (size_t)(TS + 1) * B * R * 4);
    const unsigned upm = __builtin_amdgcn_readfirstlane(is_up ? PF_OOB : 0u);
    pf_f32x4 e_ig[NE], e_fg[NE], e_og[NE], e_gg[NE], e_cp[NE], e_v2[NE];
    auto ld = [&](const __amdgpu_buffer_rsrc_t &r, unsigned off) __attribute__((always_inline)) { return __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); };
    auto fetch = [&](int h, int s, int e) __attribute__((always_inline)) {
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
    // v2: what is added to the product before the cell backward (the UP(l, h, s) tile -- another workgroup's bytes: after its
    // flag, sc1 -- or, top layer, the head term dHT at the one step where it enters)
    const __amdgpu_buffer_rsrc_t r_v2 = has_up ? r_p : pf_rsrc(a.dHT, (size_t)L * B * R * 4);
    auto fetch_v2 = [&](int h, int s) __attribute__((always_inline)) {
        const bool live_v2 = has_up || (!is_up && s == s_head);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int iloc = eloc(h, e), grow = min(rb + RBn * iloc, B - 1);
            const unsigned off = has_up ? (unsigned)((((size_t)s * B + grow) * R + u0 + 4 * eq) * 4) : (unsigned)((((size_t)l * B + grow) * R + u0 + 4 * eq) * 4);
            e_v2[e] = __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_v2, live_v2 ? off : PF_OOB, 0, 16));
        }
    };

    // ---- reduction of the four K-quarters + cell backward (REC) / tile store (UP) of half-step (h, s) --------------------
    auto spill_acc = [&](auto h_tag, int buf) __attribute__((always_inline)) {
        constexpr int H = decltype(h_tag)::value, MT = pb3_mt(TILES, NH, H);
        float *const S = Sred + buf * SRED;
        pb_nop_before_read();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(wave * ROWSH + 16 * m + 4 * lh + r) * SROW + 16 * nt + li] = acc[m][nt][r];
    };
    // prod = false: no product at this half-step (REC at the last step); head_now: arch1's head term of a layer below the top
    // (only at s = TS-1, which has no product: loaded here, where no prefetch is in flight that it would drain)
    auto cell_item = [&](auto h_tag, auto e_tag, int s, bool prod, int nr, int buf, bool head_now) __attribute__((always_inline)) {
        constexpr int H = decltype(h_tag)::value, MT = pb3_mt(TILES, NH, H), e = decltype(e_tag)::value;
        const float *const S = Sred + buf * SRED;
        const int rho = erow + RPP * e, iloc = eloc(H, e), grow = rb + RBn * iloc;
        if (rho >= 16 * MT || iloc >= nloc || (dbg & 2)) return;
        pf_f32x4 hx = {0.f, 0.f, 0.f, 0.f};
        if (head_now && has_up && s == s_head) hx = *reinterpret_cast<const pf_f32x4 *>(a.dHT + ((size_t)l * B + min(grow, B - 1)) * R + u0 + 4 * eq);
        pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (prod) {
            // the four reads FIRST, then the sum in wave order: written as v += S[..] in a loop hipcc issues one ds_read at a time and
            // waits lgkmcnt(0) before each add -- four exposed LDS round trips per item (lstm_persist_fwd3.h: the same pattern was
            // 1 900 of the forward cell's 2 250 cycles)
            const float *const Sr = S + rho * SROW + 4 * eq;
            pf_f32x4 q[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) q[w] = *reinterpret_cast<const pf_f32x4 *>(&Sr[w * ROWSH * SROW]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < 4; ++w) v += q[w];
        }
        const size_t srow_g = (size_t)s * B + grow;
        const unsigned uo = (unsigned)((srow_g * R + u0 + 4 * eq) * 4);
        if (is_up) {
            // ship the product, already multiplied by Dropout' of the layer boundary (exact: x 0 or x 1 / (1 - p))
            const uint64_t didx = ((((uint64_t)l) * B + esi[H][e]) * TS + s) * R + u0 + 4 * eq;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= dr.scale(NVQA_SITE_LSTM, didx + j);
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
            // (four reads, then four adds and writes: as bs[g] += ... each read-modify-write waited for its own LDS round trip)
            const pf_f32x4 b0 = bs[0], b1 = bs[NVQA_PF_THREADS], b2 = bs[2 * NVQA_PF_THREADS], b3 = bs[3 * NVQA_PF_THREADS];
            __builtin_amdgcn_sched_barrier(0);
            bs[0] = b0 + dgi; bs[NVQA_PF_THREADS] = b1 + dgf; bs[2 * NVQA_PF_THREADS] = b2 + dgo; bs[3 * NVQA_PF_THREADS] = b3 + dgg;
        }
        if constexpr (BF) { // the image the REC / UP products read (the f32 one stays what the weight gradients read)
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            auto img = [&](const pf_f32x4 &x, unsigned gate) __attribute__((always_inline)) {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pf_pack_bf16(x[0], x[1]), pf_pack_bf16(x[2], x[3])}, r_gb,
                                                      go / 2 + gate * (unsigned)R * 2, 0, 16);
            };
            img(dgi, 0); img(dgf, 1); img(dgo, 2); img(dgg, 3);
        }
    };
    auto cell_all = [&](auto h_tag, int s, bool prod, int nr, int buf, bool head_now) __attribute__((always_inline)) {
        [&]<int... E>(std::integer_sequence<int, E...>) __attribute__((always_inline)) { (cell_item(h_tag, std::integral_constant<int, E>{}, s, prod, nr, buf, head_now), ...); }(std::make_integer_sequence<int, NE>{});
    };
    auto signal_now = [&](int k) __attribute__((always_inline)) { // not deferred: drain, barrier, one add
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(k)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the deferred form: every wave drains ITS stores of half-step k with a counted wait (ny = memory operations it has issued
    // since the last of them), then counts itself in in LDS; the wave that arrives last adds to the counter the consumers poll
    auto signal_wave = [&](int k) __attribute__((always_inline)) {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(sigcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && (old & 3u) == 3u)
            __hip_atomic_fetch_add(own_word(__builtin_amdgcn_readfirstlane(k)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    const int KN = NH * TS;             // chain-steps
    const int k0 = is_up ? 0 : NH;      // first chain-step with a product (REC at the last step has none)
    __syncthreads();                    // sigcnt, the carried state
    // ---- REC at the last step: head term (+ the UP tile) and the cell backward only ----------------------------------------
    for (int k = 0; k < k0 && k < KN; ++k) {
        const int s = kstep(k), h = k % NH;
#pragma unroll
        for (int e = 0; e < NE; ++e) fetch(h, s, e);
        if (has_up && !(dbg & 1)) (void)pf_wait_ge(up_word(k), 1u, errw, 0x500u + l, spin_limit);
        fetch_v2(h, s);
        const int nr0 = nrows_p[s];
        [&]<int... Hh>(std::integer_sequence<int, Hh...>) __attribute__((always_inline)) {
            ((h == Hh ? cell_all(std::integral_constant<int, Hh>{}, s, false, nr0, 0, true) : (void)0), ...);
        }(std::make_integer_sequence<int, NH>{});
        signal_now(k);
    }

    // ---- the pipelined half-steps ------------------------------------------------------------------------------------------
    unsigned pend = 0, pend_up = 0;
    int pub = -1;         // half-step whose stores are issued but not yet drained and signalled
    const unsigned dbg_oob = __builtin_amdgcn_readfirstlane((dbg & 8) ? PF_OOB : 0u);
    // base offsets of the lane's rows for a half-step (half H, slice sa, `act` active tiles, enabled or not)
    auto bases = [&](auto h_tag, int sa, int act, bool en, unsigned (&b)[MTA]) __attribute__((always_inline)) {
        constexpr int H = decltype(h_tag)::value, MT = pb3_mt(TILES, NH, H);
        const unsigned enm = __builtin_amdgcn_readfirstlane(en ? 0u : PF_OOB) | dbg_oob;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            unsigned o = (voff[H][m] + (unsigned)sa * step_bytes) | enm;
            if constexpr (RAG) o = m < act ? o : PF_OOB;
            b[m] = o;
        }
    };
    auto ldA = [&](unsigned base, auto g_tag) __attribute__((always_inline)) -> pf_u32x4 {
        constexpr int g = decltype(g_tag)::value;
        return __builtin_amdgcn_raw_buffer_load_b128(r_a, base + 64u * g, 0, 16 /* sc1 */);
    };
    unsigned bcur[MTA], bnxt[MTA];
    if (k0 < KN) {
        // pipeline prologue for chain-step k0 (chain 0): its producers' counter, the first fragments
        if (!(dbg & 1)) (void)pf_wait_ge(need_word(k0), (unsigned)NU, errw, (is_up ? 0x400u : 0x300u) + l, spin_limit);
        bases(std::integral_constant<int, 0>{}, ksa(k0), act_of(0, kstep(k0)), true, bcur);
        if constexpr (!BF) {
            [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) {
                ((af[Pp % PD] = ldA(bcur[Pp % MTA], std::integral_constant<int, Pp / MTA>{})), ...);
            }(std::make_integer_sequence<int, PD>{});
        } else { // bf16: all of chain-step k0 and all of chain-step k0 + 1 (chain 1)
            constexpr int PC = PD / 2;
            [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) {
                ((af[Pp] = ldA(bcur[Pp % MTA], std::integral_constant<int, Pp / MTA>{})), ...);
            }(std::make_integer_sequence<int, PC>{});
            const bool more1 = k0 + 1 < KN;
            if (more1 && !(dbg & 1)) (void)pf_wait_ge(need_word(k0 + 1), (unsigned)NU, errw, (is_up ? 0x400u : 0x300u) + l, spin_limit);
            bases(std::integral_constant<int, 1>{}, ksa(more1 ? k0 + 1 : k0), act_of(1, kstep(more1 ? k0 + 1 : k0)), more1, bnxt);
            [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) {
                ((af[PC + Pp] = ldA(bnxt[Pp % MTA], std::integral_constant<int, Pp / MTA>{})), ...);
            }(std::make_integer_sequence<int, PC>{});
        }
    }

    // one half-step: H = its half (compile time), k its index.  On entry the fragments of its first PD pairs are in flight and
    // bcur holds its row bases; on exit the same holds for half-step k + 1.
    auto half_step = [&](auto h_tag, int k) __attribute__((always_inline)) {
        constexpr int H = decltype(h_tag)::value, HN = (H + 1) % NH;
        constexpr int MT = pb3_mt(TILES, NH, H), MTN = pb3_mt(TILES, NH, HN), P = GKT * MT;
        constexpr int RB0 = pb3_pairs_before(GKT, TILES, NH, H) % PD;   // ring slot of this chain-step's pair 0
        // where the housekeeping sits in the stream of P pairs (one fragment load per pair, all unconditional):
        // Where the housekeeping sits in the stream of P pairs.
        // f32 -- the step is MFMA-bound: the next half-step's first PD fragments are requested in this one's last PD pairs (its
        // producers' counter is looked at just before), so the stream never runs dry.
        // bf16 -- the step is a chain of latencies (a half-step's MFMAs take 2 us; a hand-off -- stores drained, counter, poll, loads
        // -- takes 5): a wait in the MIDDLE of the stream for the other chain's producers, whose signals are only 0.3 us old by then,
        // stalls MFMAs that have their data.  So the next half-step's first PD fragments are requested BEHIND this half-step's
        // stream (by then the producers signalled a whole half-step ago) and travel under the spill, the cell backward and the
        // drain of its stores; and the UP tile is requested early instead of 8 pairs (0.2 us) before its use.
        constexpr bool TAILPF = !BF;           // the next chain-step's first fragments ride in this one's last PD pairs
        // drain + signal of the PREVIOUS chain-step's stores: f32: 40 pairs in (a counted wait: nothing stalls); bf16: behind this
        // chain-step's stream (= P), i.e. a chain-step late -- its MFMAs ran under the stores' trip to memory, and with four chains
        // the consumers still have two chain-steps of slack
        constexpr int PSIG = TAILPF ? 40 : P;
        constexpr int PUPQ = BF ? 0 : P - PD - 6; // the flag of this chain-step's UP tile is requested
        constexpr int PV2 = BF ? 3 : P - 8;    // UP tile (or head term) requested
        constexpr int PREQ = (TAILPF ? P - PD : P) - 6; // the counter of the next chain-step's producers is requested (looked at 6 pairs on)
        // memory operations that are YOUNGER than the previous chain-step's stores at the signal point -- all unconditional: the
        // 5 NE cell operands requested right behind that cell backward, the fragments requested by the pairs in front of it, and
        // whichever of the UP flag, the UP tile and the producers' counter are requested in front of it
        // this chain-step's cell operands (5 NE loads) are requested 16 pairs before the end of its stream -- 1.7 us of MFMAs in f32 --
        // not earlier: 40 registers that are free during the stream are 10 more fragments in flight (PD = 32 in f32: with 16, a
        // ragged batch's stream -- MFMAs of inactive tiles skipped, their fragment slots still cycled -- was bound by 128 loads / 16 in
        // flight x 1.5 us = 12 us per chain-step whatever the row count)
        constexpr int PFETCH = P > 16 ? P - 16 : 0;
        constexpr int NFRAG = TAILPF ? PSIG : 0; // (bf16: the stream requests nothing; chain-step k + 1's fragments are older than the stores)
        constexpr int NYOUNG = (PFETCH < PSIG ? 5 * NE : 0) + NFRAG + (PUPQ < PSIG ? 1 : 0) + (PV2 < PSIG ? NE : 0) + (PREQ < PSIG ? 1 : 0);
        static_assert(NYOUNG <= 63, "vmcnt is a 6-bit field");
        static_assert(PSIG <= P && (!TAILPF || PSIG < P), "the signal point lies in this chain-step");
        static_assert(PUPQ < PV2 && PV2 < P && PREQ >= 0 && PREQ < P, "a counter is requested before it is looked at");
        const auto HT = std::integral_constant<int, H>{};
        const int s = kstep(k);
        // the chain-step whose fragments THIS one requests: the next (f32: in its tail) or the one after (bf16: behind its stream)
        constexpr int LOOK = TAILPF ? 1 : 2, HL = (H + LOOK) % NH, MTL = pb3_mt(TILES, NH, HL);
        const bool morel = k + LOOK < KN;
        const int kl = morel ? k + LOOK : k, sal = ksa(kl);
        const int act = act_of(H, s), actn = act_of(HL, kstep(kl));
        const int nr = nrows_p[s];
#pragma unroll
        for (int m = 0; m < MTA; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[m][nt] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        pb_nop_after_clear();
        bases(std::integral_constant<int, HL>{}, sal, actn, morel, bnxt);
        __builtin_amdgcn_sched_barrier(0);

        auto pair = [&](auto p_tag) __attribute__((always_inline)) {
            constexpr int p = decltype(p_tag)::value, g = p / MT, m = p % MT, slot = (RB0 + p) % PD;
            if constexpr (p == PSIG) {
                if (pub >= 0) { pb_wait_vmcnt<NYOUNG>(); signal_wave(pub); pub = -1; }
            }
            if constexpr (p == PFETCH) {
#pragma unroll
                for (int e = 0; e < NE; ++e) fetch(H, s, e);
            }
            if constexpr (p == PUPQ) {
                pend_up = __hip_atomic_load(has_up ? up_word(k) : cnt_rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::: "memory"); // (pinned here, like the request below)
            }
            if constexpr (p == PREQ) {
                pend = __hip_atomic_load(morel ? need_word(kl) : cnt_rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (pinned HERE: left alone hipcc sinks the two loads to their first use, 6 pairs on, and waits for them with
                // vmcnt(0) there -- loads return in order, so that drains the whole fragment ring once per half-step; requested
                // here they are older than the 6 fragments issued meanwhile and the wait leaves those in flight)
                asm volatile("" ::: "memory");
            }
            if constexpr (TAILPF && p == P - PD) { // the next chain-step's first fragments are requested below: its producers must be done
                if (morel && !(dbg & 1) && pend < (unsigned)NU)
                    (void)pf_wait_ge(need_word(kl), (unsigned)NU, errw, (is_up ? 0x400u : 0x300u) + l, spin_limit);
            }
            if constexpr (p == PV2) { // the UP tile of this half-step (normally long since there: UP runs ahead)
                if (has_up && !(dbg & 1) && pend_up < 1u) (void)pf_wait_ge(up_word(k), 1u, errw, 0x500u + l, spin_limit);
                fetch_v2(H, s);
            }
            if (!RAG || m < act) mfma_pair(std::integral_constant<int, g>{}, std::integral_constant<int, m>{}, af[slot]);
            // the fragment PD pairs ahead takes the slot just consumed
            if constexpr (TAILPF) {
                if constexpr (p + PD < P) af[slot] = ldA(bcur[(p + PD) % MT], std::integral_constant<int, (p + PD) / MT>{});
                else af[slot] = ldA(bnxt[(p + PD - P) % MTN], std::integral_constant<int, (p + PD - P) / MTN>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... Pp>(std::integer_sequence<int, Pp...>) __attribute__((always_inline)) { (pair(std::integral_constant<int, Pp>{}), ...); }(std::make_integer_sequence<int, P>{});
        if constexpr (PSIG == P) {
            if (pub >= 0) { pb_wait_vmcnt<NYOUNG>(); signal_wave(pub); pub = -1; }
        }
        if constexpr (!TAILPF) { // chain-step k + 2's fragments, into the ring half this chain-step has just consumed
            if (morel && !(dbg & 1) && pend < (unsigned)NU)
                (void)pf_wait_ge(need_word(kl), (unsigned)NU, errw, (is_up ? 0x400u : 0x300u) + l, spin_limit);
            [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
                ((af[(RB0 + Q) % PD] = ldA(bnxt[Q % MTL], std::integral_constant<int, Q / MTL>{})), ...);
            }(std::make_integer_sequence<int, P>{});
            __builtin_amdgcn_sched_barrier(0);
        }

        spill_acc(HT, k & 1);
        __syncthreads();
        cell_all(HT, s, true, nr, k & 1, false);
        pub = k; // drained and signalled PSIG pairs into the next half-step (its first MFMAs run under the stores' trip to memory)
#pragma unroll
        for (int m = 0; m < MTA; ++m) bcur[m] = bnxt[m];
    };

    for (int k = k0; k < KN; k += NH) {
        [&]<int... Hh>(std::integer_sequence<int, Hh...>) __attribute__((always_inline)) {
            (half_step(std::integral_constant<int, Hh>{}, k + Hh), ...);
        }(std::make_integer_sequence<int, NH>{});
    }
    if (pub >= 0) signal_now(pub);

    // bias gradients of this (layer, row block, unit tile): exactly lstm_persist_bwd2.h's exit
    if (!is_up && a.bias_part) {
        __syncthreads();
        const __amdgpu_buffer_rsrc_t r_bp = pf_rsrc(a.bias_part, (size_t)L * RBn * 4 * R * 4);
        if (tid < QPR) { // thread eq: the RPP threads (erow = 0 .. RPP-1) that own the same unit quad
            pf_f32x4 s4[4] = {pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int r = 0; r < RPP; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) s4[g] += *reinterpret_cast<const pf_f32x4 *>(bsum + (g * NVQA_PF_THREADS + r * QPR + tid) * 4);
            const unsigned doff = (unsigned)((((size_t)l * RBn + rb) * 4 * R + u0 + 4 * tid) * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, s4[g]), r_bp, doff + (unsigned)g * R * 4, 0, 16 /* sc1 */);
        }
        if (a.bias_cnt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *const flag = reinterpret_cast<unsigned *>(Sred); // (the partial tiles are dead by now)
            if (tid == 0) *flag = __hip_atomic_fetch_add(a.bias_cnt + l * NU + ut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                if (tid == 0) __hip_atomic_store(a.bias_cnt + l * NU + ut, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if ((dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 2] = wall_clock64();
}

} // namespace nvqa
