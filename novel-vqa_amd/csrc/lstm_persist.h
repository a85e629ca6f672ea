// lstm_persist.h -- the forward LSTM unroll (misc/RNNUtils.lua:128-154 driving misc/LSTM.lua:12-73) as ONE
// persistent, weight-stationary launch for all steps and layers.
//
// Why: a wavefront level is only 3.4 GFLOP (B = 512 rows x 2 layers x 2048 gate columns x K <= 1024).  As one launch
// per level every workgroup re-reads its weight tile and its activation rows from L2 for each of the 27 levels
// (240 MB per level into the CUs), pays the launch boundary, a prologue and an epilogue, and the matrix pipe ends
// up 53-56 % busy (profiles/r01_c_*).  The weights of both layers are 14 MB; the chip has 128 MB of registers.
//
// Here a workgroup (one per CU, 4 waves = one per SIMD with the whole 512-register budget) owns ONE tile
//   (layer l, row block rb of 16*MT sorted batch rows, unit tile ut of 16 hidden units = 64 gate columns)
// for every time step.  Wave w keeps the weights of gate w of its 16 units -- [W_i2h | W_h2h] rows, all of K --
// in registers as B fragments of v_mfma_f32_16x16x4_f32 for the whole launch; per step it streams its rows of
// [x_t | h_{t-1}] (sc1 loads -> registers -> LDS, 64-deep K chunks, 3-stage ring, one barrier per chunk), feeds
// them to the matrix pipe as A fragments (one ds_read_b128 = 4 MFMAs), runs the fused cell (gates via LDS to the
// thread that owns the (row, unit)), writes gates / c / h / Dropout(h) and signals.
//
// Cross-workgroup hand-off (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms" row 1): h_t and
// Dropout(h_t) are stored write-through (sc1, 16 B), every storing wave drains vmcnt, the workgroup barriers, ONE
// lane adds 1 to the (layer, row block, step) counter (agent-scope atomic); a consumer's wave 0 polls that counter
// with sc1 loads, the workgroup barriers, and EVERY load of handed-off bytes is an sc1 buffer load to registers --
// so no cache-invalidating acquire is needed.  Dependencies: (l, rb, t) needs (l, rb, t-1) [h_{t-1}, all unit
// tiles] and (l-1, rb, t) [Dropout(h^{l-1}_t)]; row blocks never talk to each other (the recurrence is per batch
// row), layer 0 never waits for a layer above it: no cycle.  Every spin is bounded (err word, all waves leave).
//
// Results are bit-reproducible run to run (fixed K order per output) but the K order differs from the per-level
// kernels of gemm_f32.h (f32 rounding only).
#pragma once
#include "ride_jobs.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "epilogues.h"

namespace nvqa {

#define NVQA_PF_MAXL 4
#define NVQA_PF_THREADS 256
#define NVQA_PF_SPIN_LIMIT (1u << 21) // polls before a wave gives up (a few seconds); after the first give-up anywhere
                                      // (err != 0) every other wait ends within 1024 polls: the launch drains

struct PersistFwdArgs {
    const float *Wi[NVQA_PF_MAXL], *Wh[NVQA_PF_MAXL], *bi[NVQA_PF_MAXL], *bh[NVQA_PF_MAXL];
    const float *X0;        // [TS*B][E] layer-0 inputs
    float *U[NVQA_PF_MAXL]; // U[l], l >= 1: [TS*B][R] Dropout(h^{l-1}) = input of layer l
    float *Hs[NVQA_PF_MAXL], *Cs[NVQA_PF_MAXL]; // [(TS+1)*B][R]; slice 0 = initial state
    float *Gt[NVQA_PF_MAXL];                    // [TS*B][4R] activated gates (kept for BPTT)
    // bf16 instance only: bf16 images of Hs / U, written next to the f32 ones by the cell epilogue and read back by the
    // K chunks (the f32 buffers stay what the backward pass and the head read).  The step is bound by the activation
    // bytes every workgroup pulls per step (B x K per (layer, unit tile)), not by its 16x cheaper MFMAs: half the bytes.
    unsigned short *Hb[NVQA_PF_MAXL], *Ub[NVQA_PF_MAXL];
    const int *nrows, *sort_idx;
    unsigned *cnt; // [L][RB][TS] arrival counters, zeroed before the launch
    unsigned *err; // != 0: a spin timed out (the launch still drains)
    int B, R, E, L, TS, RB, NU;
    unsigned spin_limit; // polls before a wave gives up (NVQA_PF_SPIN_LIMIT unless NVQA_PF_SPIN overrides it)
    int dbg;       // measurement only (NVQA_PF_DBG): 1 no flag waits, 2 no cell math / stores, 8 activation loads without
                   // memory traffic, 16 libdevice instead of hardware exp / rcp in the cell
    unsigned long long *ts; // dbg & 32: per workgroup {start, weights resident, steps done} in 100 MHz ticks (tools/pfdbg.sh)
    int h0_top;    // arch2 NVQA_QUIRK_H0: the top layer's h_{-1} (slice 0 of Hs) is live at step 0
    Drop dr;
    // f32 instances: a product for the layer-0 workgroups to multiply AFTER their last step (fr_on).  Layer 0 has 70 % of
    // layer 1's work per step (K = E + R against 2R) and nothing waits for it, so its workgroups are done ~0.28 ms before
    // the launch ends; arch1's image projection W_v Dropout(v) of the head (4.3 GFLOP, independent of the LSTM) is 128
    // tiles of 64 x 64 x 4096 for them -- off the step's critical path instead of 40 us on it (nvqa_api.hip: arch1_forward).
    RideGemm fr;   // A_KC x B_KC, full K per tile, EpiStore
    int fr_on;
};

// hardware exp2 / reciprocal forms of the gate non-linearities (v_exp_f32, v_rcp_f32: 1 ulp each): absolute error
// ~1e-7 on values in [-1, 1], ~8 instructions instead of ~30 each; the cell epilogue is on every step's critical path
__device__ __forceinline__ float pf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float pf_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pf_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7ffffff0u ? 0x7ffffff0u : bytes), 0x00020000);
}
#define PF_OOB 0x7ffffffcu // beyond every buffer: an out-of-range raw buffer load returns zeros, a store is dropped

// bounded wait for *word >= want (sc1 loads; called by one wave); false = timed out
__device__ __forceinline__ bool pf_wait_ge(unsigned *word, unsigned want, unsigned *err, unsigned code, unsigned limit = NVQA_PF_SPIN_LIMIT)
{
    unsigned spins = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) return true;
        if (++spins > limit) {
            // first give-up wins the record: {code, word index (set by the caller's code), value seen, workgroup}
            unsigned expect = 0;
            if (__hip_atomic_compare_exchange_strong(err, &expect, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err + 1, (unsigned)(word - (err - 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(err + 2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(err + 3, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
        }
        if ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        __builtin_amdgcn_s_sleep(4);
    }
}

// A "group" is the K extent of one A-fragment read (ds_read_b128): f32: 16 k = 4 x v_mfma_f32_16x16x4_f32 per row tile;
// BF (nvqa_set_precision(1): operands rounded to bf16, f32 accumulate): 32 k = ONE v_mfma_f32_16x16x32_bf16 (the gfx950
// form).  A chunk (one ring stage, one barrier) is always 64 k: 4 groups in f32, 2 in bf16; the bf16 LDS image is half
// the bytes (the float4 a thread staged is rounded with v_cvt_pk_bf16_f32 on its way into LDS), the resident weights are
// half the registers (4 packed registers per group of 32 k).
template <int G0, int G1, int MT, bool BF> struct PersistGeom {
    static constexpr int GPC = BF ? 2 : 4;                        // groups per 64-deep chunk
    static constexpr int ROWS = 16 * MT, NC0 = (G0 + GPC - 1) / GPC, NC1 = G1 / GPC, NT = NC0 + NC1, NST = 3;
    static constexpr int STAGE = ROWS * (BF ? 32 : 64);           // floats per ring stage
    static constexpr size_t LDS_BYTES = (size_t)(NST * STAGE + ROWS * 64) * 4 + 16; // ring + gate staging
};
typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pf_pack_bf16(float x, float y) // round-to-nearest-even, like the level kernels' BF mode
{
    const pf_f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pf_bf16x2));
}

// G0 / G1: K groups of the input / recurrent segment (ceil(Kin / KG), R / KG with KG = 16 (f32) or 32 (bf16)); MT row
// tiles of 16 rows per workgroup.
// INB: the input segment is read from a bf16 image (layers >= 1 of the bf16 instance: Ub); the recurrent segment is
// (Hb) whenever BF.
// RAG: ragged arch1 batches (rows sorted by length, row r active from step T - len[r] on: nrows[t] grows with t).  The
// rows of a block are dealt round-robin, so at step t the active rows of EVERY block are a prefix of its local rows and
// all blocks have the same share of them: the row-tile pairs beyond ceil(active / 16) skip their MFMAs (a wave-uniform
// branch per pair; it costs the full-length case 8 %, which therefore keeps the branch-free instance).  Loads, LDS
// traffic, barriers and flags do not change (inactive rows arrive as zeros), so the step time follows the MFMA count,
// i.e. nrows[t].
template <int G0, int G1, int MT, bool BF, bool INB, bool RAG>
__device__ __forceinline__ void persist_fwd_layer(const PersistFwdArgs &a, const int l, const int rb, const int ut,
                                                  float *smem)
{
    typedef PersistGeom<G0, G1, MT, BF> GE;
    constexpr int GPC = GE::GPC;
    static_assert(G1 % GPC == 0, "R must be a multiple of 64");
    // D: chunks between a chunk's loads and its multiplication (= staging-register sets).  f32: a chunk is 1.7 us of MFMAs,
    // two of them cover the L2 round trip.  bf16: 0.1 us -- the step is a chain of load latencies, 16 / D of them, so as
    // deep as the input segment allows (the first recurrent chunk, and the poll for it a chunk earlier, must be requested
    // after the previous step's epilogue: NC0 >= D + 1).
    constexpr int D = !BF ? 2 : (GE::NC0 >= 5 ? 4 : 3);
    static_assert(GE::NT % D == 0, "chunks per step must be a multiple of the prefetch distance (static staging-register sets across steps)");
    static_assert(GE::NC0 >= D + 1, "the first recurrent chunk (and the poll for it, a chunk earlier) must be requested after the previous step's epilogue");
    constexpr int ROWS = GE::ROWS, NC0 = GE::NC0, NT = GE::NT, NST = GE::NST, STAGE = GE::STAGE;
    float *const ring = smem;               // [NST][ROWS][64], 16-byte chunks XOR-swizzled by the row
    float *const Sg = smem + NST * STAGE;   // [ROWS][4 gates][16 units] pre-activations of the step (f32 in both modes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int B = a.B, R = a.R, TS = a.TS;
    const int Kin = l == 0 ? a.E : R;
    // Rows are dealt to the row blocks round-robin: local row i of block rb is sorted batch row rb + RB i.
    const int RBn = a.RB;
    const int u0 = ut * 16;

    // ---- weights of gate `wave`, units u0 .. u0+15, all of K: B fragments, resident for the whole launch ----------
    // MFMA 16x16x4: lane (li, lh) supplies B[k = lh][n = li]; with the A fragments read as 4 consecutive k per lane
    // (one ds_read_b128 per 16-wide group) the k of MFMA w in group g is 16 g + 4 lh + w for A and B alike.
    // (bf16: MFMA 16x16x32: lane (li, lh) supplies B[k = 8 lh + j][n = li], j = 0..7: 8 consecutive k, 4 packed registers.)
    pf_u32x4 bw[G0 + G1]; // bit patterns: 4 f32 (k = 16 g + 4 lh + w) or 8 bf16 (k = 32 g + 8 lh + j)
    {
        auto frag = [&](const float *row, int kw, int g) -> pf_u32x4 {
            if constexpr (!BF) {
                pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (16 * g + 4 * lh < kw) v = *reinterpret_cast<const pf_f32x4 *>(row + 16 * g + 4 * lh); // kw % 4 == 0
                return __builtin_bit_cast(pf_u32x4, v);
            } else {
                pf_f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                const int k = 32 * g + 8 * lh;
                if (k < kw) v0 = *reinterpret_cast<const pf_f32x4 *>(row + k);
                if (k + 4 < kw) v1 = *reinterpret_cast<const pf_f32x4 *>(row + k + 4);
                return pf_u32x4{pf_pack_bf16(v0[0], v0[1]), pf_pack_bf16(v0[2], v0[3]), pf_pack_bf16(v1[0], v1[1]), pf_pack_bf16(v1[2], v1[3])};
            }
        };
        const float *wi = a.Wi[l] + (size_t)(wave * R + u0 + li) * Kin;
#pragma unroll
        for (int g = 0; g < G0; ++g) bw[g] = frag(wi, Kin, g);
        const float *wh = a.Wh[l] + (size_t)(wave * R + u0 + li) * R;
#pragma unroll
        for (int g = 0; g < G1; ++g) bw[G0 + g] = frag(wh, R, g);
    }
    if ((a.dbg & 32) && tid == 0) a.ts[blockIdx.x * 4 + 1] = wall_clock64();
    // epilogue ownership: thread -> (row = tid / 4 + 64 e, units u0 + 4 (tid % 4) .. +3), e = 0 .. NE-1
    constexpr int NE = (ROWS + 63) / 64;
    const int eq = tid & 3, erow = tid >> 2;
    float cst[NE][4]; // the cell state of the owned (row, unit)s lives in registers across steps
    auto load_bias = [&](int g) {
        return *reinterpret_cast<const pf_f32x4 *>(a.bi[l] + g * R + u0 + 4 * eq) + *reinterpret_cast<const pf_f32x4 *>(a.bh[l] + g * R + u0 + 4 * eq);
    };
    pf_f32x4 bias_r[BF ? 4 : 1]; // bf16 instance: the packed weights leave room to keep the biases in registers
    if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_r[g] = load_bias(g);
    }
    int esi[NE];      // original batch row of the owned rows (the dropout stream is indexed by it): the same at every step
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int grow = rb + RBn * (erow + 64 * e);
        esi[e] = erow + 64 * e < ROWS && grow < a.B ? a.sort_idx[grow] : 0;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) cst[e][j] = 0.f;

    // ---- buffers that other workgroups write during the launch: sc1 buffer accesses only -----------------------
    const size_t hs_bytes = (size_t)(TS + 1) * B * R * 4, u_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_inf = l == 0 ? pf_rsrc(a.X0, (size_t)TS * B * a.E * 4) : pf_rsrc(a.U[l], u_bytes);
    const __amdgpu_buffer_rsrc_t r_h = pf_rsrc(a.Hs[l], hs_bytes);
    const bool has_next = l + 1 < a.L;
    const __amdgpu_buffer_rsrc_t r_un = has_next ? pf_rsrc(a.U[l + 1], u_bytes) : r_h;
    const __amdgpu_buffer_rsrc_t r_hb = BF ? pf_rsrc(a.Hb[l], hs_bytes / 2) : r_h;
    const __amdgpu_buffer_rsrc_t r_unb = BF && has_next ? pf_rsrc(a.Ub[l + 1], u_bytes / 2) : r_h;
    const __amdgpu_buffer_rsrc_t r_in = INB ? pf_rsrc(a.Ub[l], u_bytes / 2) : r_inf; // what the input chunks load from
    const __amdgpu_buffer_rsrc_t r_rec = BF ? r_hb : r_h;                             // what the recurrent chunks load from

    // staging map of a K chunk: thread -> float4 (row = tid / 16 + 16 j, 16-byte chunk kq = tid % 16)
    const int srow = tid >> 4, skq = tid & 15;

    // active steps of this row block form one contiguous range [t_lo, t_hi): arch1 rows start late and stay
    // (misc/RNNUtils.lua:136-145), arch2 rows all stop at tmax (Encoder_lstm.lua:185-189)
    int t_lo = 0, t_hi = 0;
    {
        int t = 0;
        while (t < TS && rb >= a.nrows[t]) ++t;
        t_lo = t;
        while (t < TS && rb < a.nrows[t]) ++t;
        t_hi = t;
    }

    // per-thread / per-workgroup invariants of the activation loads (all 32-bit: every buffer is < 2 GB)
    const unsigned step_bytes0 = (unsigned)B * Kin * 4, step_bytes1 = (unsigned)B * R * 4;
    const unsigned rstride0 = 16u * RBn * Kin * 4, rstride1 = 16u * RBn * R * 4;
    const unsigned grow0 = (unsigned)(rb + RBn * srow); // sorted batch row of this thread's first staged row
    const unsigned toff0 = (grow0 * Kin + 4 * skq) * 4, toff1 = (grow0 * R + 4 * skq) * 4;
    const int jmax = (int)grow0 < B ? (B - (int)grow0 + 16 * RBn - 1) / (16 * RBn) : 0; // rows grow0 + 16 RB j < B  <=>  j < jmax
    // bf16 source: a 64-deep chunk of a row is 128 bytes = 8 pieces of 16 B: thread -> (row = tid / 8 + 32 j, piece tid % 8),
    // j < MT / 2 -- half the load instructions, and the piece goes to LDS as it is
    const int brow = tid >> 3, bkq = tid & 7;
    const unsigned growb = (unsigned)(rb + RBn * brow);
    const unsigned step_bytesb = (unsigned)B * R * 2, rstrideb = 32u * RBn * R * 2, toffb = (growb * R + 8 * bkq) * 2;
    const int jmaxb = (int)growb < B ? (B - (int)growb + 32 * RBn - 1) / (32 * RBn) : 0;

    int mt_cur = MT, mt_nxt = MT; // RAG: row tiles with active rows at this step / at the next one
    // The layer-0 instance of a model whose layers differ (arch1: G0 != G1) does not multiply the recurrent chunks of its
    // FIRST active step (h_{-1} = 0, and before a late row block starts its h slices are written as zeros: their loads, LDS traffic and barriers stay -- the chunk pipeline is untouched -- only the
    // MFMAs go, behind a wave-uniform branch per pair).  Layer 0 is not the critical path (K = 712 against 1024), so the
    // branches cost nothing there, and every layer above starts 8 chunks x 1.7 us earlier: the launch is that much shorter.
    constexpr bool SKIP0 = !BF && G0 != G1; // (bf16: the step is issue-bound, not MFMA-bound: the branches cost more than the MFMAs)
    bool skip_rec = false;
    pf_u32x4 stg[D][MT];
    unsigned pend = 0; // value of the counter the next flagged chunk depends on, requested a chunk ahead of its use
    // ask for the counter that chunk q of step t waits for (always a load -- of counter 0 when nothing is awaited --
    // so that no load sits under a runtime branch)
    auto poll_request = [&](int t, auto q_tag, bool en) {
        constexpr int q = decltype(q_tag)::value;
        static_assert(q == 0 || q == NC0, "only the first chunk of a segment waits");
        const bool flagged = en && (q == 0 ? l > 0 : t > 0);
        const size_t idx = !flagged ? 0 : (q == 0 ? ((size_t)(l - 1) * a.RB + rb) * TS + t : ((size_t)l * a.RB + rb) * TS + (t - 1));
        pend = __hip_atomic_load(a.cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // request chunk q of step t into staging set SET; where the bytes are another workgroup's, wait for its counter.
    // The loads are issued on EVERY path (en = false: out-of-range offsets, which return zeros without touching
    // memory): a load under a runtime branch makes hipcc wait vmcnt(0) at the join, i.e. for the chunk just requested
    // instead of the one requested an iteration ago (cdna_hip_programming.md section 5, trap (c)).
    unsigned pf_o0 = PF_OOB; // offset of the thread's first piece of the chunk being requested
    int pf_jlim = 0, pf_jlimb = 0; // pieces of that chunk with rows worth loading (RAG: the active row tiles of ITS step)
    auto prefetch_begin = [&](int t, auto q_tag, bool en, bool next_step = false) {
        constexpr int q = decltype(q_tag)::value;
        pf_jlim = RAG ? min(jmax, next_step ? mt_nxt : mt_cur) : jmax;
        pf_jlimb = RAG ? min(jmaxb, ((next_step ? mt_nxt : mt_cur) + 1) >> 1) : jmaxb; // (a bf16-image piece spans two row tiles)
        constexpr bool s1 = q >= NC0;
        // (step 0 multiplies the recurrent chunks too: slice 0 of Hs holds h_{-1} = 0, or the carried h0 of
        // NVQA_QUIRK_H0 -- one step of 26 with 50 % more MFMAs is cheaper than a second copy of the unrolled chunk
        // loop, which would push the per-step code of the two layers past the 64 KB instruction cache)
        // hand-off check, per wave (MI355X_MICROARCH.md "Valid forms" row 1: a wave loads handed-off bytes after ITS poll of
        // the producers' counter matched): the counter was requested one chunk earlier (poll_request), so this is
        // normally a register compare; a late producer costs a bounded spin.  A timeout sets err and goes on -- the
        // launch always drains, the host reports the step as failed.
        if constexpr (q == 0 || q == NC0) {
            const bool flagged = q == 0 ? l > 0 : t > 0;
            if (flagged && en && !(a.dbg & 1) && pend < (unsigned)a.NU) {
                unsigned *need = q == 0 ? a.cnt + ((size_t)(l - 1) * a.RB + rb) * TS + t : a.cnt + ((size_t)l * a.RB + rb) * TS + (t - 1);
                (void)pf_wait_ge(need, (unsigned)a.NU, a.err, (q == 0 ? 0x100u : 0x200u) + l, a.spin_limit);
            }
        }
        if (a.dbg & 8) en = false;
        // OR-ed into every offset; readfirstlane keeps it a scalar the optimiser does not look through, or it clones
        // the loads into an `en` and a `!en` branch and the waits at the join cover both
        const unsigned enm = __builtin_amdgcn_readfirstlane(en ? 0u : PF_OOB);
        constexpr int c = s1 ? q - NC0 : q;
        // byte offset of this thread's first float4 of the chunk: step base (scalar) + thread part (precomputed) + 256 c;
        // row j of the thread's MT rows is 16 rows further: + j * rstride.  Out-of-range pieces get PF_OOB (| or + keeps
        // them beyond every buffer), 2 vector instructions per load in all.
        if constexpr (s1 ? BF : INB) { // bf16 image (K = R, a multiple of 64: no tail)
            pf_o0 = (toffb + (unsigned)t * step_bytesb + 128u * c) | enm;
        } else {
            const unsigned sbase = (unsigned)t * (s1 ? step_bytes1 : step_bytes0) + 256u * c;
            pf_o0 = ((64 * c + 4 * skq < (s1 ? R : Kin)) ? (s1 ? toff1 : toff0) + sbase : PF_OOB) | enm;
        }
    };
    // loads j0 .. j1-1 of the chunk prepared by prefetch_begin (one or two per MFMA pair: a burst of MT loads between
    // two groups idles the matrix pipe for the time it takes to issue them)
    // (pieces are numbered 0 .. MT-1 in both forms; a bf16-image chunk has a load / LDS write on the even ones only)
    auto prefetch_piece = [&](auto q_tag, auto set_tag, auto j0_tag, auto j1_tag) {
        constexpr int q = decltype(q_tag)::value, SET = decltype(set_tag)::value;
        constexpr bool s1 = q >= NC0;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < MT; ++j) {
            if constexpr (s1 ? BF : INB) {
                if (j & 1) continue;
                const unsigned off = (j >> 1) < pf_jlimb ? pf_o0 + (unsigned)(j >> 1) * rstrideb : PF_OOB;
                stg[SET][j >> 1] = __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rec : r_in, off, 0, 16 /* sc1 */);
            } else {
                const unsigned off = j < pf_jlim ? pf_o0 + (unsigned)j * (s1 ? rstride1 : rstride0) : PF_OOB;
                stg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rec : r_in, off, 0, 16 /* sc1 */);
            }
        }
    };
    auto prefetch = [&](int t, auto q_tag, auto set_tag, bool en) {
        prefetch_begin(t, q_tag, en);
        prefetch_piece(q_tag, set_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, MT>{});
    };
    // chunk q (compile time: it decides the form of the staged pieces) -> ring stage
    auto commit_piece = [&](auto q_tag, auto set_tag, int stage, auto j0_tag, auto j1_tag) {
        constexpr int q = decltype(q_tag)::value, SET = decltype(set_tag)::value;
        constexpr bool s1 = q >= NC0;
        float *dst = ring + stage * STAGE;
#pragma unroll
        for (int j = decltype(j0_tag)::value; j < decltype(j1_tag)::value && j < MT; ++j) {
            if constexpr (s1 ? BF : INB) {
                if (j & 1) continue;
                const int row = brow + 32 * (j >> 1);
                *reinterpret_cast<pf_u32x4 *>(&dst[row * 32 + 4 * (bkq ^ ((row >> 1) & 7))]) = stg[SET][j >> 1];
            } else {
                const int row = srow + 16 * j;
                if constexpr (!BF) {
                    *reinterpret_cast<pf_u32x4 *>(&dst[row * 64 + 4 * (skq ^ (row & 15))]) = stg[SET][j];
                } else { // 128-byte rows: the 16-byte chunk index (skq / 2) is XOR-ed with row / 2, this thread's 4 k are half a chunk
                    const pf_f32x4 v = __builtin_bit_cast(pf_f32x4, stg[SET][j]);
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t h = {pf_pack_bf16(v[0], v[1]), pf_pack_bf16(v[2], v[3])};
                    *reinterpret_cast<u32x2_t *>(&dst[row * 32 + 4 * ((skq >> 1) ^ ((row >> 1) & 7)) + 2 * (skq & 1)]) = h;
                }
            }
        }
    };
    auto commit = [&](auto q_tag, auto set_tag, int stage) {
        commit_piece(q_tag, set_tag, stage, std::integral_constant<int, 0>{}, std::integral_constant<int, MT>{});
    };

    pf_f32x4 acc[MT];
    pf_u32x4 af[MT]; // A fragments of the group about to be multiplied (carried across chunks and steps): 4 f32 or 8 bf16
    // MFMAs of group g of chunk q (compile time: they select the resident B fragments).  The A fragments of a
    // row-tile pair are replaced right after the pair's MFMAs -- by the next group of the chunk, or (last group) by
    // the first group of the NEXT chunk from ring stage `nxt` -- so a full group of other pairs' MFMAs hides the LDS
    // latency with one fragment set; the two row tiles of a pair alternate so that an accumulator is reused every
    // second MFMA (40-cycle dependent latency, 32-cycle issue).
    auto mfma_group = [&](auto q_tag, auto g_tag, const float *cur, const float *nxt, auto &&hook) {
        constexpr int q = decltype(q_tag)::value, g = decltype(g_tag)::value;
        constexpr bool s1 = q >= NC0;
        constexpr int c = s1 ? q - NC0 : q;
        constexpr int GB = s1 ? G0 + GPC * c : GPC * c;                                 // first B-fragment group of the chunk
        constexpr int NG = s1 ? GPC : (GPC * c + GPC <= G0 ? GPC : G0 - GPC * c);       // groups with data in the chunk
        if constexpr (g < NG) {
            auto frag_at = [&](const float *src, int m, int gi) -> pf_u32x4 {
                const int row = m * 16 + li;
                if constexpr (!BF) return *reinterpret_cast<const pf_u32x4 *>(&src[row * 64 + 4 * ((4 * gi + lh) ^ li)]);
                else return *reinterpret_cast<const pf_u32x4 *>(&src[row * 32 + 4 * ((4 * gi + lh) ^ ((row >> 1) & 7))]);
            };
            auto pair = [&](auto mp_tag) {
                constexpr int mp = decltype(mp_tag)::value;
                if ((!RAG || mp < mt_cur) && !(SKIP0 && s1 && skip_rec)) {
                    if constexpr (!BF) {
#pragma unroll
                        for (int w = 0; w < 4; ++w)
#pragma unroll
                            for (int m = mp; m < mp + 2 && m < MT; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(pf_f32x4, af[m])[w],
                                                                              __builtin_bit_cast(pf_f32x4, bw[GB + g])[w], acc[m], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int m = mp; m < mp + 2 && m < MT; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, af[m]),
                                                                             __builtin_bit_cast(pf_bf16x8, bw[GB + g]), acc[m], 0, 0, 0);
                    }
                }
                const float *src = g + 1 < NG ? cur : nxt;
                constexpr int gn = g + 1 < NG ? g + 1 : 0;
                // (the fragments fetched by the last group of a step are the next step's: its activity decides)
                if (!RAG || mp < (q == NT - 1 && g == NG - 1 ? mt_nxt : mt_cur)) {
#pragma unroll
                    for (int m = mp; m < mp + 2 && m < MT; ++m) af[m] = frag_at(src, m, gn);
                }
                hook(mp_tag); // a piece of the next chunks' housekeeping, under this pair's MFMAs
                // keep the refill (and the piece) HERE: left alone, hipcc sinks these reads to just above their first use
                // (shorter live range) and every pair then opens with an exposed LDS round trip -- 15 % of the step
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... P>(std::integer_sequence<int, P...>) { (pair(std::integral_constant<int, 2 * P>{}), ...); }(std::make_integer_sequence<int, (MT + 1) / 2>{});
        }
    };
    auto chunk_groups = [](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        constexpr bool s1 = q >= NC0;
        constexpr int c = s1 ? q - NC0 : q;
        return std::integral_constant<int, (s1 ? GPC : (GPC * c + GPC <= G0 ? GPC : G0 - GPC * c))>{};
    };

    // fused cell for step t (active = false: the whole row block has not started / has stopped: zeros), then publish
    const unsigned cbase = (unsigned)((l * a.RB + rb) * TS); // this workgroup's (layer, row block) counters
    auto publish = [&](int t) {
        const unsigned idx = cbase + (unsigned)__builtin_amdgcn_readfirstlane(t); // scalar, 32-bit: see `pub` below
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto epilogue = [&](int t, bool act, bool defer) {
        if (act) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sg[((16 * m + 4 * lh + r) * 4 + wave) * 16 + li] = acc[m][r];
        }
        __syncthreads();
        const int nr = a.nrows[t];
        pf_f32x4 bias[4]; // b_i2h + b_h2h of the owned unit quad (f32: re-read per step, 8 cached loads instead of 16 registers)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (BF) bias[g] = bias_r[g];
            else bias[g] = load_bias(g);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = erow + 64 * e, grow = rb + RBn * row;
            if (row >= ROWS || grow >= B) continue;
            const bool on = act && grow < nr && !(a.dbg & 2);
            if ((a.dbg & 2) && e >= 0) continue;
            pf_f32x4 gi = {0.f, 0.f, 0.f, 0.f}, gf = gi, go = gi, gg = gi, cn = gi, hn = gi, un = gi;
            if (on) {
                const pf_f32x4 p0 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 0) * 16 + 4 * eq]);
                const pf_f32x4 p1 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 1) * 16 + 4 * eq]);
                const pf_f32x4 p2 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 2) * 16 + 4 * eq]);
                const pf_f32x4 p3 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 3) * 16 + 4 * eq]);
                const uint64_t didx = ((((uint64_t)l) * B + esi[e]) * TS + t) * R + u0 + 4 * eq;
                if (a.dbg & 16) { // libdevice expf / tanhf (A/B against the hardware forms)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gi[j] = sigmoidf_(p0[j] + bias[0][j]);
                        gf[j] = sigmoidf_(p1[j] + bias[1][j]);
                        go[j] = sigmoidf_(p2[j] + bias[2][j]);
                        gg[j] = tanhf_(p3[j] + bias[3][j]);
                        cn[j] = gf[j] * cst[e][j] + gi[j] * gg[j];
                        hn[j] = go[j] * tanhf_(cn[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gi[j] = pf_sigmoid(p0[j] + bias[0][j]);
                        gf[j] = pf_sigmoid(p1[j] + bias[1][j]);
                        go[j] = pf_sigmoid(p2[j] + bias[2][j]);
                        gg[j] = pf_tanh(p3[j] + bias[3][j]);
                        cn[j] = gf[j] * cst[e][j] + gi[j] * gg[j];
                        hn[j] = go[j] * pf_tanh(cn[j]);
                    }
                }
                if (has_next) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) un[j] = a.dr.scale(NVQA_SITE_LSTM, didx + j) * hn[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cst[e][j] = cn[j];
            const size_t srow_g = (size_t)t * B + grow;
            float *gt = a.Gt[l] + srow_g * 4 * R + u0 + 4 * eq;
            *reinterpret_cast<pf_f32x4 *>(gt) = gi;
            *reinterpret_cast<pf_f32x4 *>(gt + R) = gf;
            *reinterpret_cast<pf_f32x4 *>(gt + 2 * R) = go;
            *reinterpret_cast<pf_f32x4 *>(gt + 3 * R) = gg;
            const size_t so = ((size_t)(t + 1) * B + grow) * R + u0 + 4 * eq;
            *reinterpret_cast<pf_f32x4 *>(a.Cs[l] + so) = cn;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, hn), r_h, (unsigned)(so * 4), 0, 16);
            if (has_next)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pf_u32x4, un), r_un,
                                                       (unsigned)((srow_g * R + u0 + 4 * eq) * 4), 0, 16);
            if constexpr (BF) { // the images the K chunks read: same values, rounded once here instead of at every reader
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pf_pack_bf16(hn[0], hn[1]), pf_pack_bf16(hn[2], hn[3])}, r_hb,
                                                      (unsigned)(so * 2), 0, 16);
                if (has_next)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{pf_pack_bf16(un[0], un[1]), pf_pack_bf16(un[2], un[3])}, r_unb,
                                                          (unsigned)((srow_g * R + u0 + 4 * eq) * 2), 0, 16);
            }
        }
        if (defer) return; // drained and signalled from the next step's first chunk, under its MFMAs (publish_deferred)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its write-through stores
        __syncthreads();
        publish(t);
    };

    // ---- steps before the row block starts ----------------------------------------------------------------------
    for (int t = 0; t < t_lo; ++t) epilogue(t, false, false);

    if (t_lo < t_hi) {
        unsigned n = 0; // running chunk counter: ring stage = n % NST
        // pipeline prologue: chunks 0 .. D-1 of the first active step (all input chunks: NC0 > D)
        poll_request(t_lo, std::integral_constant<int, 0>{}, true);
        prefetch(t_lo, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, true);
        commit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
        [&]<int... Q>(std::integer_sequence<int, Q...>) {
            (prefetch(t_lo, std::integral_constant<int, Q + 1>{}, std::integral_constant<int, Q + 1>{}, true), ...);
        }(std::make_integer_sequence<int, D - 1>{});
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + li;
            af[m] = BF ? *reinterpret_cast<const pf_u32x4 *>(&ring[row * 32 + 4 * (lh ^ ((row >> 1) & 7))])
                       : *reinterpret_cast<const pf_u32x4 *>(&ring[row * 64 + 4 * (lh ^ li)]);
        }
        // step whose h / Dropout(h) stores are issued but not yet drained and signalled.  (publish() takes it through
        // readfirstlane into 32-bit index arithmetic: as a sign-extended 64-bit vector index, at the 512-register
        // limit this kernel runs at, hipcc 7.2 lost the high half in one build and the add went astray.)
        int pub = -1;
        for (int t = t_lo; t < t_hi; ++t) {
            if constexpr (SKIP0) skip_rec = __builtin_amdgcn_readfirstlane((t == t_lo && !(a.h0_top && l == a.L - 1)) ? 1 : 0) != 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            const bool more = t + 1 < t_hi;
            if constexpr (RAG) {
                auto mt_of = [&](int tt) {
                    const int nr = a.nrows[tt], nloc = nr > rb ? (nr - rb + RBn - 1) / RBn : 0;
                    return __builtin_amdgcn_readfirstlane(min(MT, (nloc + 15) >> 4));
                };
                mt_cur = mt_of(t);
                mt_nxt = more ? mt_of(t + 1) : mt_cur;
            }
            auto iter = [&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value;
                constexpr int NG = decltype(chunk_groups(q_tag))::value;
                const float *cur = ring + (n % NST) * STAGE, *nxt = ring + ((n + 1) % NST) * STAGE;
                // The housekeeping of the NEXT chunks sits between the MFMA groups of this one, so that the matrix
                // pipe never waits for it (one wave per SIMD: nobody else would fill the gap):
                //   group 0 | request chunk q+2 | group 1 | chunk q+1 -> LDS | group 2 | barrier | group 3 (+ first
                //   fragments of chunk q+1).  Chunks with fewer groups keep the order of the remaining pieces.
                // chunk q+2: flag check + offsets now, its loads two per MFMA pair under group 0
                if constexpr (q + D < NT) prefetch_begin(t, std::integral_constant<int, q + D>{}, true);
                else prefetch_begin(more ? t + 1 : t, std::integral_constant<int, q + D - NT>{}, more, true);
                constexpr int QN = q + D < NT ? q + D : q + D - NT; // index of that chunk inside its step
                auto none = [](auto) {};
                auto loads = [&](auto mp_tag) {
                    constexpr int mp = decltype(mp_tag)::value;
                    prefetch_piece(std::integral_constant<int, QN>{}, std::integral_constant<int, q % D>{},
                                   std::integral_constant<int, mp>{}, std::integral_constant<int, mp + 2>{});
                };
                auto writes = [&](auto mp_tag) { // chunk q+1 -> LDS (a chunk that does not exist arrives as zeros, never multiplied)
                    constexpr int mp = decltype(mp_tag)::value;
                    commit_piece(std::integral_constant<int, (q + 1) % NT>{}, std::integral_constant<int, (q + 1) % D>{}, (int)((n + 1) % NST),
                                 std::integral_constant<int, mp>{}, std::integral_constant<int, mp + 2>{});
                };
                if constexpr (NG > 1) mfma_group(q_tag, std::integral_constant<int, 0>{}, cur, nxt, loads);
                else prefetch_piece(std::integral_constant<int, QN>{}, std::integral_constant<int, q % D>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, MT>{});
                // counter of the flagged chunk that is requested NEXT iteration (chunk q+3), one chunk ahead of its use
                if constexpr (q + D + 1 == NC0) poll_request(t, std::integral_constant<int, NC0>{}, true);
                if constexpr (q + D + 1 == NT) poll_request(more ? t + 1 : t, std::integral_constant<int, 0>{}, more);
                if constexpr (NG > 2) mfma_group(q_tag, std::integral_constant<int, 1>{}, cur, nxt, writes);
                else commit(std::integral_constant<int, (q + 1) % NT>{}, std::integral_constant<int, (q + 1) % D>{}, (int)((n + 1) % NST));
                if constexpr (NG > 3) mfma_group(q_tag, std::integral_constant<int, 2>{}, cur, nxt, none);
                if constexpr (q == 0) {
                    // the previous step's write-through stores are older than the MT loads of the prefetch above: drained
                    // here, signalled behind this chunk's barrier (publish R1: every storing wave drains, barrier, one lane)
                    if (pub >= 0) { // (chunk D is an input chunk: MT loads, MT / 2 from a bf16 image)
                        if constexpr ((INB ? MT / 2 : MT) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else if constexpr ((INB ? MT / 2 : MT) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    }
                }
                __syncthreads();
                if constexpr (q == 0) {
                    if (pub >= 0) { publish(pub); pub = -1; }
                }
                // last group of the chunk: its fragment refills come from chunk q+1, published by the barrier above
                mfma_group(q_tag, std::integral_constant<int, NG - 1>{}, cur, nxt, none);
                ++n;
            };
            // the chunks of a step, unrolled: each one names its own resident B fragments
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (iter(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NT>{});
            epilogue(t, true, more);
            if (more) pub = t;
        }
    }
    // ---- steps after the row block has stopped (arch2: t >= tmax) ---------------------------------------------------
    for (int t = t_hi; t < TS; ++t) epilogue(t, false, false);
}

// KA: K of layer 0's input (E, rounded up to the group width); workgroup id -> (layer, row block, unit tile): the NU unit
// tiles of one (layer, row block) share an id modulo the group count, i.e. one XCD when there are 8 groups.
template <int KA, int KR, int MT, bool BF, bool RAG>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_fwd_persist(PersistFwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pf_smem[];
    constexpr int KG = BF ? 32 : 16, G0A = (KA + KG - 1) / KG, GR = KR / KG;
    const int groups = a.L * a.RB;
    const int grp = blockIdx.x % groups, ut = blockIdx.x / groups;
    const int l = grp / a.RB, rb = grp % a.RB;
    if ((a.dbg & 32) && threadIdx.x == 0) a.ts[blockIdx.x * 4] = wall_clock64();
    if constexpr (BF && KA == KR) {
        // bf16, E = R (arch2): layer 0 reads its input chunks from the bf16 image of X0 (a.Ub[0], left by the embedding kernel)
        // and is the SAME instance as the layers above -- one copy of the unrolled step loop fewer in the instruction cache;
        // without the image (NVQA_X0_B16=0) it stages and rounds the f32 rows as before
        if (l == 0 && !a.Ub[0]) persist_fwd_layer<G0A, GR, MT, BF, false, RAG>(a, l, rb, ut, pf_smem);
        else persist_fwd_layer<GR, GR, MT, BF, BF, RAG>(a, l, rb, ut, pf_smem);
    } else {
        if (l == 0) persist_fwd_layer<G0A, GR, MT, BF, false, RAG>(a, l, rb, ut, pf_smem);
        else persist_fwd_layer<GR, GR, MT, BF, BF, RAG>(a, l, rb, ut, pf_smem);
    }
    if constexpr (!BF) {
        if (l == 0 && a.fr_on) { // the riding product (PersistFwdArgs::fr): tiles dealt round-robin to the layer-0 workgroups
            __syncthreads();
            const int me = ut * a.RB + rb, n = a.RB * a.NU, tx = a.fr.tx, total = tx * a.fr.ty;
            for (int t = me; t < total; t += n) {
                gemm_f32_body<CfgRide, A_KC, B_KC, false, EpiStore, 0, true>(a.fr.g, a.fr.e, t % tx, t / tx, 0, pf_smem);
                __syncthreads();
            }
        }
    }
    if ((a.dbg & 32) && threadIdx.x == 0) a.ts[blockIdx.x * 4 + 2] = wall_clock64();
}
template <int KA, int KR, int MT, bool BF> constexpr size_t persist_fwd_lds()
{
    constexpr int KG = BF ? 32 : 16, G0A = (KA + KG - 1) / KG, GR = KR / KG;
    return PersistGeom<G0A, GR, MT, BF>::LDS_BYTES > PersistGeom<GR, GR, MT, BF>::LDS_BYTES ? PersistGeom<G0A, GR, MT, BF>::LDS_BYTES
                                                                                          : PersistGeom<GR, GR, MT, BF>::LDS_BYTES;
}

} // namespace nvqa
