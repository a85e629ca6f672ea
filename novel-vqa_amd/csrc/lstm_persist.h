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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "epilogues.h"

namespace nvqa {

#define NVQA_PF_MAXL 4
#define NVQA_PF_THREADS 256
#define NVQA_PF_SPIN_LIMIT (1u << 23) // polls before a workgroup gives up (seconds)

struct PersistFwdArgs {
    const float *Wi[NVQA_PF_MAXL], *Wh[NVQA_PF_MAXL], *bi[NVQA_PF_MAXL], *bh[NVQA_PF_MAXL];
    const float *X0;        // [TS*B][E] layer-0 inputs
    float *U[NVQA_PF_MAXL]; // U[l], l >= 1: [TS*B][R] Dropout(h^{l-1}) = input of layer l
    float *Hs[NVQA_PF_MAXL], *Cs[NVQA_PF_MAXL]; // [(TS+1)*B][R]; slice 0 = initial state
    float *Gt[NVQA_PF_MAXL];                    // [TS*B][4R] activated gates (kept for BPTT)
    const int *nrows, *sort_idx;
    unsigned *cnt; // [L][RB][TS] arrival counters, zeroed before the launch
    unsigned *err; // != 0: a spin timed out (the launch still drains)
    int B, R, E, L, TS, RB, NU;
    int h0_top;    // arch2 NVQA_QUIRK_H0: the top layer's h_{-1} (slice 0 of Hs) is live at step 0
    Drop dr;
};

typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pf_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7ffffff0u ? 0x7ffffff0u : bytes), 0x00020000);
}
#define PF_OOB 0x7ffffffcu // beyond every buffer: an out-of-range raw buffer load returns zeros, a store is dropped

// bounded wait for *word >= want (sc1 loads; called by one wave); false = timed out
__device__ __forceinline__ bool pf_wait_ge(unsigned *word, unsigned want, unsigned *err, unsigned code)
{
    unsigned spins = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) return true;
        if (++spins > NVQA_PF_SPIN_LIMIT) {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

template <int G0, int G1, int MT> struct PersistGeom {
    static constexpr int ROWS = 16 * MT, NC0 = (G0 + 3) / 4, NC1 = G1 / 4, NT = NC0 + NC1, NST = 3;
    static constexpr int STAGE = ROWS * 64;                       // floats per ring stage
    static constexpr size_t LDS_BYTES = (size_t)(NST + 1) * STAGE * 4 + 16; // ring + gate staging + flag word
};

// G0: 16-wide K groups of the input segment (ceil(Kin / 16)); G1 = R / 16 groups of the recurrent segment; MT row
// tiles of 16 rows per workgroup.
template <int G0, int G1, int MT>
__device__ __forceinline__ void persist_fwd_layer(const PersistFwdArgs &a, const int l, const int rb, const int ut,
                                                  float *smem)
{
    typedef PersistGeom<G0, G1, MT> GE;
    static_assert(G1 % 4 == 0, "R must be a multiple of 64");
    static_assert(GE::NT % 2 == 0, "chunks per step must be even (static staging-register sets across steps)");
    static_assert(GE::NC0 >= 2, "the first recurrent chunk must be requested after the previous step's epilogue");
    constexpr int ROWS = GE::ROWS, NC0 = GE::NC0, NT = GE::NT, NST = GE::NST, STAGE = GE::STAGE;
    float *const ring = smem;               // [NST][ROWS][64], 16-byte chunks XOR-swizzled by the row
    float *const Sg = smem + NST * STAGE;   // [ROWS][4 gates][16 units] pre-activations of the step
    int *const sm_flag = reinterpret_cast<int *>(smem + (NST + 1) * STAGE); // broadcast of wave 0's poll result
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lh = lane >> 4;
    const int B = a.B, R = a.R, TS = a.TS;
    const int Kin = l == 0 ? a.E : R;
    const int r0 = rb * ROWS;
    const int u0 = ut * 16;

    // ---- weights of gate `wave`, units u0 .. u0+15, all of K: B fragments, resident for the whole launch ----------
    // MFMA 16x16x4: lane (li, lh) supplies B[k = lh][n = li]; with the A fragments read as 4 consecutive k per lane
    // (one ds_read_b128 per 16-wide group) the k of MFMA w in group g is 16 g + 4 lh + w for A and B alike.
    float bw[(G0 + G1) * 4];
    {
        const float *wi = a.Wi[l] + (size_t)(wave * R + u0 + li) * Kin + 4 * lh;
#pragma unroll
        for (int g = 0; g < G0; ++g) {
            pf_f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (16 * g + 4 * lh < Kin) v = *reinterpret_cast<const pf_f32x4 *>(wi + 16 * g); // Kin % 4 == 0
            bw[4 * g + 0] = v[0]; bw[4 * g + 1] = v[1]; bw[4 * g + 2] = v[2]; bw[4 * g + 3] = v[3];
        }
        const float *wh = a.Wh[l] + (size_t)(wave * R + u0 + li) * R + 4 * lh;
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const pf_f32x4 v = *reinterpret_cast<const pf_f32x4 *>(wh + 16 * g);
            bw[4 * (G0 + g) + 0] = v[0]; bw[4 * (G0 + g) + 1] = v[1]; bw[4 * (G0 + g) + 2] = v[2]; bw[4 * (G0 + g) + 3] = v[3];
        }
    }
    // epilogue ownership: thread -> (row = tid / 4 + 64 e, units u0 + 4 (tid % 4) .. +3), e = 0 .. NE-1
    constexpr int NE = (ROWS + 63) / 64;
    const int eq = tid & 3, erow = tid >> 2;
    float cst[NE][4]; // the cell state of the owned (row, unit)s lives in registers across steps
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) cst[e][j] = 0.f;

    // ---- buffers that other workgroups write during the launch: sc1 buffer accesses only -----------------------
    const size_t hs_bytes = (size_t)(TS + 1) * B * R * 4, u_bytes = (size_t)TS * B * R * 4;
    const __amdgpu_buffer_rsrc_t r_in = l == 0 ? pf_rsrc(a.X0, (size_t)TS * B * a.E * 4) : pf_rsrc(a.U[l], u_bytes);
    const __amdgpu_buffer_rsrc_t r_h = pf_rsrc(a.Hs[l], hs_bytes);
    const bool has_next = l + 1 < a.L;
    const __amdgpu_buffer_rsrc_t r_un = has_next ? pf_rsrc(a.U[l + 1], u_bytes) : r_h;

    // staging map of a K chunk: thread -> float4 (row = tid / 16 + 16 j, 16-byte chunk kq = tid % 16)
    const int srow = tid >> 4, skq = tid & 15;
    const bool h0_live = a.h0_top && l == a.L - 1;

    // active steps of this row block form one contiguous range [t_lo, t_hi): arch1 rows start late and stay
    // (misc/RNNUtils.lua:136-145), arch2 rows all stop at tmax (Encoder_lstm.lua:185-189)
    int t_lo = 0, t_hi = 0;
    {
        int t = 0;
        while (t < TS && r0 >= a.nrows[t]) ++t;
        t_lo = t;
        while (t < TS && r0 < a.nrows[t]) ++t;
        t_hi = t;
    }

    pf_u32x4 stg[2][MT];
    // request chunk q of step t into staging set SET; where the bytes are another workgroup's, wait for its counter
    auto prefetch = [&](int t, auto q_tag, auto set_tag) -> bool {
        constexpr int q = decltype(q_tag)::value, SET = decltype(set_tag)::value;
        constexpr bool s1 = q >= NC0;
        if (s1 && !(t > 0 || h0_live)) return true; // h_{-1} = 0: no recurrent product at step 0
        unsigned *need = nullptr;
        unsigned code = 0;
        if (q == 0 && l > 0) { need = a.cnt + ((size_t)(l - 1) * a.RB + rb) * TS + t; code = 0x100u + l; }
        if (q == NC0 && t > 0) { need = a.cnt + ((size_t)l * a.RB + rb) * TS + (t - 1); code = 0x200u + l; }
        if (need) {
            if (wave == 0) {
                const bool ok = pf_wait_ge(need, (unsigned)a.NU, a.err, code);
                if (lane == 0) *sm_flag = ok ? 1 : 0;
            }
            __syncthreads();
            const int ok = *sm_flag;
            __syncthreads(); // the flag word may be rewritten by the next wait
            if (!ok) return false;
        }
        constexpr int c = s1 ? q - NC0 : q;
        const int kw = s1 ? R : Kin;
        const size_t base = (size_t)t * B * kw; // Hs slice t = h_{t-1}; input slice t
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = r0 + srow + 16 * j, k = 64 * c + 4 * skq;
            const unsigned off = (row < B && k < kw) ? (unsigned)((base + (size_t)row * kw + k) * 4) : PF_OOB;
            stg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_h : r_in, off, 0, 16 /* sc1 */);
        }
        return true;
    };
    auto commit = [&](auto set_tag, int stage) {
        constexpr int SET = decltype(set_tag)::value;
        float *dst = ring + stage * STAGE;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = srow + 16 * j;
            *reinterpret_cast<pf_u32x4 *>(&dst[row * 64 + 4 * (skq ^ (row & 15))]) = stg[SET][j];
        }
    };

    pf_f32x4 acc[MT];
    // MFMAs of chunk q (compile time: it selects the resident B fragments) from ring stage `stage`
    auto compute = [&](auto q_tag, int stage) {
        constexpr int q = decltype(q_tag)::value;
        constexpr bool s1 = q >= NC0;
        constexpr int c = s1 ? q - NC0 : q;
        constexpr int GB = s1 ? G0 + 4 * c : 4 * c;                          // first B-fragment group of the chunk
        constexpr int NG = s1 ? 4 : (4 * c + 4 <= G0 ? 4 : G0 - 4 * c);      // groups with data in the chunk
        const float *src = ring + stage * STAGE;
        // A fragments of a row-tile pair are re-read for the next group right after the pair's MFMAs of this group
        // (a full group of other pairs' MFMAs hides the LDS latency) -- one fragment set instead of two; the two row
        // tiles of a pair alternate so that an accumulator is reused every second MFMA (40-cycle dependent latency)
        pf_f32x4 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const pf_f32x4 *>(&src[(m * 16 + li) * 64 + 4 * (lh ^ li)]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int mp = 0; mp < MT; mp += 2) {
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int m = mp; m < mp + 2 && m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][w], bw[4 * (GB + g) + w], acc[m], 0, 0, 0);
                if (g + 1 < NG) {
#pragma unroll
                    for (int m = mp; m < mp + 2 && m < MT; ++m)
                        af[m] = *reinterpret_cast<const pf_f32x4 *>(&src[(m * 16 + li) * 64 + 4 * ((4 * (g + 1) + lh) ^ li)]);
                }
            }
        }
    };

    // fused cell for step t (active = false: the whole row block has not started / has stopped: zeros), then publish
    auto epilogue = [&](int t, bool act) {
        if (act) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sg[((16 * m + 4 * lh + r) * 4 + wave) * 16 + li] = acc[m][r];
        }
        __syncthreads();
        const int nr = a.nrows[t];
        pf_f32x4 bias[4]; // b_i2h + b_h2h of the owned unit quad (re-read per step: 8 cached loads instead of 16 registers)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const pf_f32x4 x = *reinterpret_cast<const pf_f32x4 *>(a.bi[l] + g * R + u0 + 4 * eq);
            const pf_f32x4 y = *reinterpret_cast<const pf_f32x4 *>(a.bh[l] + g * R + u0 + 4 * eq);
            bias[g] = x + y;
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = erow + 64 * e, grow = r0 + row;
            if (row >= ROWS || grow >= B) continue;
            const bool on = act && grow < nr;
            pf_f32x4 gi = {0.f, 0.f, 0.f, 0.f}, gf = gi, go = gi, gg = gi, cn = gi, hn = gi, un = gi;
            if (on) {
                const pf_f32x4 p0 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 0) * 16 + 4 * eq]);
                const pf_f32x4 p1 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 1) * 16 + 4 * eq]);
                const pf_f32x4 p2 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 2) * 16 + 4 * eq]);
                const pf_f32x4 p3 = *reinterpret_cast<const pf_f32x4 *>(&Sg[(row * 4 + 3) * 16 + 4 * eq]);
                const uint64_t didx = ((((uint64_t)l) * B + a.sort_idx[grow]) * TS + t) * R + u0 + 4 * eq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gi[j] = sigmoidf_(p0[j] + bias[0][j]);
                    gf[j] = sigmoidf_(p1[j] + bias[1][j]);
                    go[j] = sigmoidf_(p2[j] + bias[2][j]);
                    gg[j] = tanhf_(p3[j] + bias[3][j]);
                    cn[j] = gf[j] * cst[e][j] + gi[j] * gg[j];
                    hn[j] = go[j] * tanhf_(cn[j]);
                    if (has_next) un[j] = a.dr.scale(NVQA_SITE_LSTM, didx + j) * hn[j];
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
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + ((size_t)l * a.RB + rb) * TS + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- steps before the row block starts ----------------------------------------------------------------------
    for (int t = 0; t < t_lo; ++t) epilogue(t, false);

    if (t_lo < t_hi) {
        unsigned n = 0; // running chunk counter: ring stage = n % NST
        bool ok = true;
        // pipeline prologue: chunks 0 and 1 of the first active step
        ok = prefetch(t_lo, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if (ok) {
            commit(std::integral_constant<int, 0>{}, 0);
            ok = prefetch(t_lo, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
        for (int t = t_lo; t < t_hi && ok; ++t) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = pf_f32x4{0.f, 0.f, 0.f, 0.f};
            const bool s1_now = t > 0 || h0_live, more = t + 1 < t_hi;
            auto iter = [&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value;
                if (!ok) return;
                // A: request chunk q+2 (this step or the next) into the staging set chunk q has just left
                if constexpr (q + 2 < NT) {
                    ok = prefetch(t, std::integral_constant<int, q + 2>{}, std::integral_constant<int, q & 1>{});
                } else {
                    if (more) ok = prefetch(t + 1, std::integral_constant<int, q + 2 - NT>{}, std::integral_constant<int, q & 1>{});
                }
                if (!ok) return;
                // B: chunk q+1 -> LDS
                bool have_next;
                if constexpr (q + 1 < NT) have_next = (q + 1 < NC0) || s1_now;
                else have_next = more;
                if (have_next) commit(std::integral_constant<int, (q + 1) & 1>{}, (int)((n + 1) % NST));
                __syncthreads();
                // C: multiply chunk q
                if ((q < NC0) || s1_now) compute(q_tag, (int)(n % NST));
                ++n;
            };
            // the chunks of a step, unrolled: each one names its own resident B fragments
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (iter(std::integral_constant<int, Q>{}), ...); }(std::make_integer_sequence<int, NT>{});
            if (!ok) break;
            epilogue(t, true);
        }
        if (!ok) return; // a wait timed out: err is set, every wave of the workgroup leaves together
    }
    // ---- steps after the row block has stopped (arch2: t >= tmax) ---------------------------------------------------
    for (int t = t_hi; t < TS; ++t) epilogue(t, false);
}

// G0A: K groups of layer 0's input (ceil(E / 16)); GR = R / 16.  Workgroup id -> (layer, row block, unit tile): the
// NU unit tiles of one (layer, row block) share an id modulo the group count, i.e. one XCD when there are 8 groups.
template <int G0A, int GR, int MT>
__global__ __launch_bounds__(NVQA_PF_THREADS, 1) void k_lstm_fwd_persist(PersistFwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pf_smem[];
    const int groups = a.L * a.RB;
    const int grp = blockIdx.x % groups, ut = blockIdx.x / groups;
    const int l = grp / a.RB, rb = grp % a.RB;
    if (l == 0) persist_fwd_layer<G0A, GR, MT>(a, l, rb, ut, pf_smem);
    else persist_fwd_layer<GR, GR, MT>(a, l, rb, ut, pf_smem);
}

} // namespace nvqa
