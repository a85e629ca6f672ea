// lstm_persist_bwd.h -- BPTT through the LSTM stack (misc/RNNUtils.lua:182-209 driving the nngraph backward of
// misc/LSTM.lua:12-73) as ONE persistent, weight-stationary launch: the counterpart of lstm_persist.h.
//
// Per (layer l, step s) the chain needs   dh^l_s = dG^l_{s+1} W_h2h^l  +  Dropout'(dG^{l+1}_s W_i2h^{l+1})  + head term
// and then the cell backward, which turns the stored gates of (l, s) into dG^l_s in place.  As launches that was
// one split-K GEMM level (33 us) plus a slab-summing finisher (8 us) per wavefront diagonal, the matrix pipe 55 % busy.
//
// Here every workgroup (one per CU, 4 waves, 512 registers each) owns for the whole launch ONE product tile
//   (role, row block rb, unit tile ut of 32 hidden units)      K = 4R (the gate pre-activations), N = 32 units
// with role REC(l) = the recurrent product of layer l + the cell backward of its (rows, units), or UP(l) = the
// product that carries the gradient from layer l+1 down to layer l (l < L-1).  Wave w keeps the K-quarter of gate w
// of its weight block [4R x 32] in registers (2 x 128 B fragments of v_mfma_f32_16x16x4_f32); per step the
// workgroup streams its rows of dG (sc1 loads -> registers -> LDS, one 16-wide K group per gate and chunk, 3-stage
// ring), every wave multiplies its quarter, the four partial tiles are summed through LDS in wave order
// (deterministic), and the epilogue is the fused cell backward (REC) or a write-through store of the tile (UP).
//
// Hand-offs as in lstm_persist.h (sc1 stores, drain, barrier, one agent-scope counter add; consumers poll with sc1
// loads and read the bytes with sc1 buffer loads):  REC(l, s) waits for REC(l, s+1) [its A operand, all unit tiles
// of the row block] and, below the top layer, for the one UP(l, s) tile with its rows and units; UP(l, s) waits for
// REC(l+1, s).  No cycle: the top layer waits only for itself.  Every spin is bounded (err word).
//
// bf16 instance (nvqa_set_precision(1)): v_mfma_f32_16x16x32_bf16, the weights as packed bf16 (half the registers), so a
// workgroup holds 64 units (4 column tiles per wave) instead of 32 and HALF as many workgroups re-read each dG slice;
// the A operand comes from a bf16 image of dG that the cell backward writes next to the f32 one (half the bytes again);
// the LDS image of a chunk is byte-for-byte the f32 layout (4 gates x 32 k x 2 B = 4 x 16 x 4 B per row) and a 16-byte
// piece IS the lane's A fragment; 4 chunks in flight instead of 2 (a chunk is 0.1 us of MFMAs here, the step is a chain
// of L2 round trips).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "lstm_persist.h"

namespace nvqa {

struct PersistBwdArgs {
    const float *Wh[NVQA_PF_MAXL], *Wi[NVQA_PF_MAXL]; // W_h2h^l [4R][R]; W_i2h^l [4R][R] for l >= 1
    float *Gt[NVQA_PF_MAXL];                          // [TS*B][4R] gates in, d(pre-activations) out
    const float *Cs[NVQA_PF_MAXL];                    // [(TS+1)*B][R]
    const float *dCT, *dHT;                           // [L][B][R] head -> final cell / hidden state gradients
    float *Pup[NVQA_PF_MAXL];                         // Pup[l], l < L-1: [TS*B][R] products of the UP(l) role
    unsigned short *Gb[NVQA_PF_MAXL];                 // bf16 instance: [TS*B][4R] bf16 image of dG (written by REC, read as A)
    float *bias_part;                                 // [L][RB][4R]: column sums of dG over the steps and the rows of a row
                                                      // block (the LSTM bias gradients, summed over rb by k_bias_sum)
    const int *nrows, *sort_idx, *tlast;              // tlast (arch2): dHT enters at step *tlast; NULL (arch1): at TS-1
    unsigned *cnt_rec;                                // [L][RB][TS] arrivals of the REC(l) unit tiles
    unsigned *cnt_up;                                 // [L][RB][NU][TS] flag of the UP(l) tile
    unsigned *err;
    unsigned spin_limit;
    unsigned long long *ts;  // dbg & 32: per workgroup {start, weights resident, steps done} in 100 MHz ticks
    int dbg;                 // measurement only (NVQA_PB_DBG): 1 no flag waits, 2 no cell math / stores, 8 A loads without memory traffic
    int B, R, L, TS, RB, NU; // NU = R / (16 NTN) unit tiles
    Drop dr;
};

template <int MT, int NTN> struct PersistBwdGeom {
    static constexpr int ROWS = 16 * MT, NST = 3, STAGE = ROWS * 64, UNITS = 16 * NTN;
    static constexpr int BSUM_FLOATS = NVQA_PF_THREADS * 16; // per thread: 4 gates x 4 units of bias-gradient partial sums
    static constexpr size_t LDS_BYTES = (size_t)(NST * STAGE + 4 * ROWS * UNITS + BSUM_FLOATS) * 4; // ring + the four waves' partial tiles + bias sums
};

// GK: K groups (= chunks) per gate: R / 16 (f32), R / 32 (bf16); MT row tiles of 16 rows and NTN column tiles of 16 units
// per workgroup
// RAG: ragged arch1 batches -- as in lstm_persist.h, the row tiles of a block without rows active at step s skip
// their MFMAs and their A loads
template <int GK, int MT, int NTN, bool BF, bool RAG>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_bwd_persist(PersistBwdArgs a)
{
    typedef PersistBwdGeom<MT, NTN> GE;
    constexpr int D = BF ? 4 : 2; // chunks in flight = staging-register sets
    static_assert(GK % D == 0, "chunks per step must be a multiple of the prefetch distance (static staging-register sets)");
    static_assert(MT >= 3, "at least two row-tile pairs: one before the chunk barrier, one after");
    constexpr int ROWS = GE::ROWS, NST = GE::NST, STAGE = GE::STAGE, NT = GK, UNITS = GE::UNITS;
    constexpr int ES = BF ? 2 : 4, KG = BF ? 32 : 16; // bytes per A element, K per group
    extern __shared__ __attribute__((aligned(16))) float pb_smem[];
    float *const ring = pb_smem;               // [NST][ROWS][64 words]: per row 4 gates x 16 B x 4, 16-byte pieces XOR-swizzled
    float *const Sred = pb_smem + NST * STAGE; // [4 waves][ROWS][UNITS] partial tiles
    float *const bsum = Sred + 4 * ROWS * UNITS;   // [thread][4 gates][4 units]: sum of the thread's dG over its rows and all steps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int B = a.B, R = a.R, TS = a.TS, L = a.L, RBn = a.RB;

    // workgroup -> (role, row block, unit tile).  roles: 0 .. L-1 = REC(l = L-1-role) top layer first; L .. 2L-2 = UP(l = 2L-2-role)
    // The NU unit tiles of one (role, row block) group exchange dG among themselves: they sit on ONE XCD (workgroups are
    // dealt round-robin to the 8 XCDs: id % 8), two groups per XCD where NU = 16 (32 CUs each), REC(l+1, rb) next to the
    // UP(l, rb) that reads its output.  Speed only: the protocol does not depend on the placement.
    const int ngroups = (2 * L - 1) * RBn, gpx = 32 / a.NU > 0 ? 32 / a.NU : 1; // groups per XCD
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int gslot = xcd * gpx + slot / a.NU, ut = slot % a.NU;
    if (slot / a.NU >= gpx || gslot >= ngroups) return; // (grid = 8 XCDs x 32 slots; unused slots leave at once)
    // group order: (REC(L-1, rb), UP(L-2, rb)) pairs first, then the REC groups of the lower layers
    int role, rb;
    if (L == 1) { role = 0; rb = gslot; }
    else if (gslot < 2 * RBn) { role = (gslot & 1) ? L : 0; rb = gslot >> 1; }
    else { role = 1; rb = gslot - 2 * RBn; }
    const bool is_up = role >= L;
    const int l = is_up ? 2 * L - 2 - role : L - 1 - role; // the layer whose dh this tile belongs to
    const int la = is_up ? l + 1 : l;                      // the layer whose dG is the A operand
    const int u0 = ut * UNITS;
    const bool has_up = !is_up && l + 1 < L;               // REC below the top layer: adds the UP(l) tile
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4] = wall_clock64();

    // ---- weights: rows k = wave * R + kk (gate `wave`), columns u0 .. u0+31 of W [4R][R]; resident B fragments ---------
    const float *W = is_up ? a.Wi[l + 1] : a.Wh[l];
    pf_u32x4 bw[NTN][GK]; // f32: 4 k = 16 g + 4 lh + w; bf16: 8 k = 32 g + 8 lh + j (packed pairs)
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int g = 0; g < GK; ++g) {
            const float *w0 = W + (size_t)(wave * R + KG * g + (KG / 4) * lh) * R + u0 + 16 * nt + li;
            if constexpr (!BF) {
                bw[nt][g] = __builtin_bit_cast(pf_u32x4, pf_f32x4{w0[0], w0[(size_t)R], w0[2 * (size_t)R], w0[3 * (size_t)R]});
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bw[nt][g][j] = pf_pack_bf16(w0[(size_t)(2 * j) * R], w0[(size_t)(2 * j + 1) * R]);
            }
        }

    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 1] = wall_clock64();
    const size_t gt_bytes = (size_t)TS * B * 4 * R * 4, pup_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_a = BF ? pf_rsrc(a.Gb[la], gt_bytes / 2) : pf_rsrc(a.Gt[la], gt_bytes); // A operand: dG of layer la
    const __amdgpu_buffer_rsrc_t r_gb = BF ? pf_rsrc(a.Gb[l], gt_bytes / 2) : r_a;   // REC, bf16: the image of dG it writes
    const __amdgpu_buffer_rsrc_t r_g = pf_rsrc(a.Gt[l], gt_bytes);                   // REC: gates in / dG out
    const __amdgpu_buffer_rsrc_t r_p = pf_rsrc(l + 1 < L ? a.Pup[l] : a.Gt[l], l + 1 < L ? pup_bytes : gt_bytes);

    // staging map of a chunk (K group g): thread -> (row = tid / 16 + 16 j, piece kq = tid % 16 = gate kq / 4, 4 (kq % 4) .. +3)
    const int srow = tid >> 4, skq = tid & 15;
    const unsigned grow0 = (unsigned)(rb + RBn * srow);
    const unsigned toff = (grow0 * 4u * R + (unsigned)(skq >> 2) * R + (16u / ES) * (skq & 3)) * ES;
    const unsigned rstride = 16u * RBn * 4u * R * ES, step_bytes = (unsigned)B * 4u * R * ES;
    const int jmax = (int)grow0 < B ? (B - (int)grow0 + 16 * RBn - 1) / (16 * RBn) : 0;

    pf_u32x4 stg[D][MT];
    unsigned pf_o0 = PF_OOB;
    int mt_cur = MT; // RAG: row tiles with rows active at this step
    // A slice `sa` of Gt[la] (dG of step sa), K group g; en = false: zeros without memory traffic
    auto prefetch_begin = [&](int sa, auto g_tag, bool en) {
        constexpr int g = decltype(g_tag)::value;
        const unsigned enm = __builtin_amdgcn_readfirstlane(en ? 0u : PF_OOB);
        pf_o0 = (toff + (unsigned)sa * step_bytes + 64u * g) | enm;
    };
    auto prefetch_piece = [&](auto set_tag, auto j0_tag, auto j1_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < MT; ++j) {
            const unsigned off = j < (RAG ? min(jmax, mt_cur) : jmax) ? pf_o0 + (unsigned)j * rstride : PF_OOB;
            stg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(r_a, off, 0, 16 /* sc1 */);
        }
    };
    auto commit_piece = [&](auto set_tag, int stage, auto j0_tag, auto j1_tag) {
        constexpr int SET = decltype(set_tag)::value;
        float *dst = ring + stage * STAGE;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < MT; ++j) {
            const int row = srow + 16 * j;
            *reinterpret_cast<pf_u32x4 *>(&dst[row * 64 + 4 * (skq ^ (row & 15))]) = stg[SET][j];
        }
    };
    const auto J0 = std::integral_constant<int, 0>{};
    const auto JN = std::integral_constant<int, MT>{};

    pf_f32x4 acc[MT][NTN];
    pf_u32x4 af[MT];
    auto refill = [&](const float *src, auto m0_tag, auto m1_tag) { // this wave's 4 k of every row tile: piece 4 wave + lh
#pragma unroll
        for (int m = decltype(m0_tag)::value; m < decltype(m1_tag)::value && m < MT; ++m)
            af[m] = *reinterpret_cast<const pf_u32x4 *>(&src[(m * 16 + li) * 64 + 4 * ((4 * wave + lh) ^ li)]);
    };
    // MFMAs of row tiles mp, mp+1 for K group g (f32: 2 x NTN x 4, an accumulator is reused every 2 NTN-th MFMA)
    auto pair = [&](auto g_tag, auto mp_tag) {
        constexpr int g = decltype(g_tag)::value, mp = decltype(mp_tag)::value;
        if (RAG && mp >= mt_cur) return;
        if constexpr (!BF) {
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int m = mp; m < mp + 2 && m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(pf_f32x4, af[m])[w],
                                                                          __builtin_bit_cast(pf_f32x4, bw[nt][g])[w], acc[m][nt], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = mp; m < mp + 2 && m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, af[m]),
                                                                         __builtin_bit_cast(pf_bf16x8, bw[nt][g]), acc[m][nt], 0, 0, 0);
        }
    };

    // epilogue ownership: thread -> (row = tid / QPR + RPP e, units u0 + 4 (tid % QPR) .. +3), QPR = quads per row
    constexpr int QPR = UNITS / 4, RPP = NVQA_PF_THREADS / QPR, NE = (ROWS + RPP - 1) / RPP;
    const int eq = tid % QPR, erow = tid / QPR;
    float dcst[NE][4]; // REC: the carried cell gradient of the owned (row, unit)s
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int row = erow + RPP * e, grow = rb + RBn * row;
        pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!is_up && row < ROWS && grow < B) v = *reinterpret_cast<const pf_f32x4 *>(a.dCT + ((size_t)l * B + grow) * R + u0 + 4 * eq);
#pragma unroll
        for (int j = 0; j < 4; ++j) dcst[e][j] = v[j];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<pf_f32x4 *>(bsum + (tid * 4 + g) * 4) = pf_f32x4{0.f, 0.f, 0.f, 0.f};
    int esi[NE]; // original batch row of the owned rows (indexes the dropout stream): the same at every step
#pragma unroll
    for (int e = 0; e < NE; ++e) esi[e] = a.sort_idx[min(rb + RBn * (erow + RPP * e), B - 1)];
    const unsigned crec = (unsigned)((l * RBn + rb) * TS);                 // REC(l) counters of this row block
    const unsigned cneed = (unsigned)((la * RBn + rb) * TS);               // counters of the producers of the A operand
    const unsigned cup = (unsigned)(((l * RBn + rb) * a.NU + ut) * TS);    // UP(l) flag of this tile

    for (int s = TS - 1; s >= 0; --s) {
        // REC: the cell backward's operands (own gates and cell states of the forward pass: nothing in this launch writes
        // them before this workgroup does) are requested HERE, ahead of the counter wait and the whole product, for the
        // first two of the thread's NE items -- as many as fit beside the resident weights; the other items are requested
        // as the first two are consumed.  Requested in the epilogue they were 4 exposed HBM round trips per step (4.4 us).
        const int nr = a.nrows[s];
        if constexpr (RAG) mt_cur = __builtin_amdgcn_readfirstlane(min(MT, ((nr > rb ? (nr - rb + RBn - 1) / RBn : 0) + 15) >> 4));
        const bool head_now = a.tlast ? (*a.tlast == s) : (s == TS - 1);
        pf_f32x4 e_ig[2], e_fg[2], e_og[2], e_gg[2], e_cc[2], e_cp[2];
        pf_f32x4 e_v2[NE]; // the UP(l, s) products of the owned cells (another workgroup's bytes: after its flag, sc1 loads)
        auto fetch = [&](int e, int k) { // unconditional, row clamped into the batch
            if (a.dbg & 4) return; // measurement: the cell backward's operands are not loaded
            const int row = erow + RPP * e, grow = min(rb + RBn * row, B - 1);
            const size_t srow_g = (size_t)s * B + grow;
            const float *gt = a.Gt[l] + srow_g * 4 * R + u0 + 4 * eq; // own gates of the forward pass: plain loads
            e_ig[k] = *reinterpret_cast<const pf_f32x4 *>(gt);
            e_fg[k] = *reinterpret_cast<const pf_f32x4 *>(gt + R);
            e_og[k] = *reinterpret_cast<const pf_f32x4 *>(gt + 2 * R);
            e_gg[k] = *reinterpret_cast<const pf_f32x4 *>(gt + 3 * R);
            e_cc[k] = *reinterpret_cast<const pf_f32x4 *>(a.Cs[l] + ((size_t)(s + 1) * B + grow) * R + u0 + 4 * eq);
            e_cp[k] = *reinterpret_cast<const pf_f32x4 *>(a.Cs[l] + srow_g * R + u0 + 4 * eq);
        };
        if (!is_up) {
            fetch(0, 0);
            if constexpr (NE > 1) fetch(1, 1);
        }
        // A operand: REC(l, s): dG^l_{s+1} (absent at the last step); UP(l, s): dG^{l+1}_s
        const int sa = is_up ? s : s + 1;
        const bool live = sa < TS && !(a.dbg & 8);
        if (sa < TS && !(a.dbg & 1)) {
            // every wave polls for itself (lstm_persist.h): all NU unit tiles of REC(la) at step sa
            (void)pf_wait_ge(a.cnt_rec + cneed + sa, (unsigned)a.NU, a.err, (is_up ? 0x400u : 0x300u) + l, a.spin_limit);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[m][nt] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
        // pipeline prologue (nothing can be requested before the producers' step is complete: the chain is serial here)
        prefetch_begin(sa, std::integral_constant<int, 0>{}, live);
        prefetch_piece(std::integral_constant<int, 0>{}, J0, JN);
        [&]<int... Q>(std::integer_sequence<int, Q...>) { // chunks 1 .. D-1
            ([&] {
                prefetch_begin(sa, std::integral_constant<int, Q + 1>{}, live);
                prefetch_piece(std::integral_constant<int, Q + 1>{}, J0, JN);
            }(), ...);
        }(std::make_integer_sequence<int, D - 1>{});
        commit_piece(std::integral_constant<int, 0>{}, 0, J0, JN);
        __syncthreads();
        refill(ring, J0, JN);
        auto iter = [&](auto q_tag) {
            constexpr int q = decltype(q_tag)::value;
            const float *nxt = ring + ((q + 1) % NST) * STAGE;
            // pairs 0 .. NP-2 before the barrier with the next chunks' housekeeping under them, the last pair after it
            constexpr int NP = (MT + 1) / 2;
            if constexpr (q + D < NT) prefetch_begin(sa, std::integral_constant<int, (q + D < NT ? q + D : 0)>{}, live);
            [&]<int... P>(std::integer_sequence<int, P...>) {
                ([&] {
                    pair(q_tag, std::integral_constant<int, 2 * P>{});
                    if constexpr (q + D < NT) // loads of chunk q+D: spread over the pairs before the barrier
                        prefetch_piece(std::integral_constant<int, q % D>{}, std::integral_constant<int, (MT * P) / (NP - 1)>{},
                                       std::integral_constant<int, (MT * (P + 1)) / (NP - 1)>{});
                    // chunk q+1 -> LDS (its loads were issued a chunk ago), one pair ahead of the barrier where there is one:
                    // the writes then land under that pair's MFMAs instead of in front of the barrier
                    if constexpr (q + 1 < NT && P == (NP >= 3 ? NP - 3 : NP - 2))
                        commit_piece(std::integral_constant<int, (q + 1) % D>{}, (q + 1) % NST, J0, JN);
                    __builtin_amdgcn_sched_barrier(0);
                }(), ...);
            }(std::make_integer_sequence<int, NP - 1>{});
            __syncthreads();
            // the fragments of the next chunk for every row tile, then the last pair (its MFMAs cover their latency)
            if constexpr (q + 1 < NT) refill(nxt, J0, std::integral_constant<int, 2 * (NP - 1)>{});
            __builtin_amdgcn_sched_barrier(0);
            pair(q_tag, std::integral_constant<int, 2 * (NP - 1)>{});
            if constexpr (q + 1 < NT) refill(nxt, std::integral_constant<int, 2 * (NP - 1)>{}, JN);
            __builtin_amdgcn_sched_barrier(0);
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (iter(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NT>{});

        // ---- the four K-quarters of the tile -> LDS, summed in wave order by the owner of each (row, unit quad) ----------
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sred[(wave * ROWS + 16 * m + 4 * lh + r) * UNITS + 16 * nt + li] = acc[m][nt][r];
        auto finish = [&](int e, int k) {
            const int row = erow + RPP * e, grow = rb + RBn * row;
            if (row >= ROWS || grow >= B || (a.dbg & 2)) return;
            pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const pf_f32x4 *>(&Sred[(w * ROWS + row) * UNITS + 4 * eq]);
            const size_t srow_g = (size_t)s * B + grow;
            const unsigned uo = (unsigned)((srow_g * R + u0 + 4 * eq) * 4);
            if (is_up) { // ship the product; the cell of layer l adds it
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, v), r_p, uo, 0, 16);
                return;
            }
            const unsigned go = (unsigned)((srow_g * 4 * R + u0 + 4 * eq) * 4);
            pf_f32x4 dgi = {0.f, 0.f, 0.f, 0.f}, dgf = dgi, dgo = dgi, dgg = dgi, dcn = dgi;
            if (grow < nr) {
                const pf_f32x4 ig = e_ig[k], fg = e_fg[k], og = e_og[k], gg = e_gg[k], cc = e_cc[k], cp = e_cp[k], v2 = e_v2[e];
                pf_f32x4 hx = {0.f, 0.f, 0.f, 0.f};
                if (head_now) hx = *reinterpret_cast<const pf_f32x4 *>(a.dHT + ((size_t)l * B + grow) * R + u0 + 4 * eq);
                const uint64_t didx = ((((uint64_t)l) * B + esi[e]) * TS + s) * R + u0 + 4 * eq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dsc = has_up ? a.dr.scale(NVQA_SITE_LSTM, didx + j) : 0.f;
                    const float dh = v[j] + dsc * v2[j] + hx[j];
                    const float tc = pf_tanh(cc[j]);
                    const float dcv = dcst[e][j] + dh * og[j] * (1.0f - tc * tc);
                    dgi[j] = dcv * gg[j] * ig[j] * (1.0f - ig[j]);
                    dgf[j] = dcv * cp[j] * fg[j] * (1.0f - fg[j]);
                    dgo[j] = dh * tc * og[j] * (1.0f - og[j]);
                    dgg[j] = dcv * ig[j] * (1.0f - gg[j] * gg[j]);
                    dcn[j] = dcv * fg[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dcst[e][j] = dcn[j];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgi), r_g, go, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgf), r_g, go + (unsigned)R * 4, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgo), r_g, go + 2u * R * 4, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, dgg), r_g, go + 3u * R * 4, 0, 16);
            if (grow < nr) { // bias gradient: column sums of dG (own LDS slot: no other thread touches it)
                pf_f32x4 *bs = reinterpret_cast<pf_f32x4 *>(bsum + tid * 16);
                bs[0] += dgi; bs[1] += dgf; bs[2] += dgo; bs[3] += dgg;
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
        if (!is_up) {
            if (has_up && !(a.dbg & 1)) // the UP(l, s) tile with these rows and units (normally long since there: UP runs ahead)
                (void)pf_wait_ge(a.cnt_up + cup + s, 1u, a.err, 0x500u + l, a.spin_limit);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int row = erow + RPP * e, grow = min(rb + RBn * row, B - 1);
                e_v2[e] = __builtin_bit_cast(pf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                           r_p, has_up ? (unsigned)((((size_t)s * B + grow) * R + u0 + 4 * eq) * 4) : PF_OOB, 0, 16));
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            finish(e, e & 1); // dcst is indexed by the compile-time e after unrolling
            if (!is_up && e + 2 < NE) fetch(e + 2, e & 1); // the slot just consumed takes the item after next
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its write-through stores
        __syncthreads();                                 // (also: Sred and the ring are free for the next step)
        if (tid == 0) {
            unsigned *word = is_up ? a.cnt_up + cup + (unsigned)__builtin_amdgcn_readfirstlane(s)
                                   : a.cnt_rec + crec + (unsigned)__builtin_amdgcn_readfirstlane(s);
            __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // bias gradients of this (layer, row block, unit tile): the row groups' partial sums added in a fixed order
    if (!is_up && a.bias_part) {
        __syncthreads();
        if (tid < QPR) { // thread eq: the RPP threads (erow = 0 .. RPP-1) that own the same unit quad
            pf_f32x4 s4[4] = {pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}, pf_f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int r = 0; r < RPP; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) s4[g] += *reinterpret_cast<const pf_f32x4 *>(bsum + ((r * QPR + tid) * 4 + g) * 4);
            float *dst = a.bias_part + ((size_t)l * RBn + rb) * 4 * R + u0 + 4 * tid;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<pf_f32x4 *>(dst + (size_t)g * R) = s4[g];
        }
    }
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 2] = wall_clock64();
}

} // namespace nvqa
