// epilogues.h -- fused GEMM epilogues of the VQA step (functors for gemm_f32.h).
// Each one replaces a run of separate Torch7 pointwise modules; the reference line of the
// module chain is cited at each functor.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/nvqa_rng.h"

namespace nvqa {

struct Drop {
    int mode;
    float p, inv_keep;
    uint64_t seed, step;
    __device__ __forceinline__ float scale(uint32_t site, uint64_t idx) const
    {
        return mode ? nvqa_dropout_scale(seed, step, site, idx, p, inv_keep) : 1.0f;
    }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanhf from libdevice is accurate to ~1 ulp; keep it (logit tolerance is 1e-4 relative)
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

// C[z][m][n] = v   (split-K partial slabs when gridDim.z > 1)
struct EpiStore {
    float *C;
    int ldc;
    size_t slab;
    __device__ __forceinline__ void operator()(int z, int m, int n, float v) const
    {
        C[(size_t)z * slab + (size_t)m * ldc + n] = v;
    }
};

// nn.Linear bias add (two biases: b_i2h + b_h2h of misc/LSTM.lua:41-43 folded into the
// time-batched i2h product)
struct EpiBias2 {
    float *C;
    int ldc;
    const float *b1, *b2;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        C[(size_t)m * ldc + n] = v + b1[n] + (b2 ? b2[n] : 0.0f);
    }
};

// tanh(Linear(.)) -- misc/netdef.lua:10 (qc branch)
struct EpiBiasTanh {
    float *C;
    int ldc;
    const float *b;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        C[(size_t)m * ldc + n] = tanhf_(v + b[n]);
    }
};

// ic = tanh(Linear(.)); z = qc (*) ic; zd = Dropout(z) -- netdef.lua:11-12, 002_train_baseline.lua:153
struct EpiFuse {
    float *ic, *zd;
    const float *qc, *b;
    int ldc;
    Drop dr;
    int askip; // netdef.AskipB (misc/netdef.lua:16-25): qc + qc (*) ic
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        const size_t o = (size_t)m * ldc + n;
        const float i = tanhf_(v + b[n]);
        ic[o] = i;
        if (askip == 2) { // netdef.A_B: zd = Dropout([qc | ic]), 2 ldc wide
            const size_t z = (size_t)m * 2 * ldc + n;
            zd[z] = dr.scale(NVQA_SITE_Z, z) * qc[o];
            zd[z + ldc] = dr.scale(NVQA_SITE_Z, z + ldc) * i;
            return;
        }
        zd[o] = dr.scale(NVQA_SITE_Z, o) * (askip ? qc[o] + qc[o] * i : qc[o] * i);
    }
};

// backward of Dropout -> CMulTable -> the two Tanh (netdef.lua:10-12): v = d(zd)
struct EpiHeadBwd {
    float *dqc, *dic;
    const float *qc, *ic;
    int ldc;
    Drop dr;
    int askip;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        if (askip == 2) { // netdef.A_B: v = d(zd)[m][n], n < 2 ldc: the qc half, then the ic half
            const float dz = dr.scale(NVQA_SITE_Z, (size_t)m * 2 * ldc + n) * v;
            if (n < ldc) { const float q = qc[(size_t)m * ldc + n]; dqc[(size_t)m * ldc + n] = dz * (1.0f - q * q); }
            else { const float i = ic[(size_t)m * ldc + n - ldc]; dic[(size_t)m * ldc + n - ldc] = dz * (1.0f - i * i); }
            return;
        }
        const size_t o = (size_t)m * ldc + n;
        const float dz = dr.scale(NVQA_SITE_Z, o) * v;
        const float q = qc[o], i = ic[o];
        dqc[o] = dz * (askip ? 1.0f + i : i) * (1.0f - q * q);
        dic[o] = dz * q * (1.0f - i * i);
    }
};

// d(qd) -> Dropout backward -> re-sort into BPTT order and split into the per-layer
// (c, h) state gradients: 002_train_baseline.lua:313 + misc/LSTM.lua:70 (JoinTable backward)
struct EpiResort {
    float *dC, *dH; // [L][B][R]
    const int *sort_inv;
    int B, R, Q;
    Drop dr;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        const float s = dr.scale(NVQA_SITE_Q, (uint64_t)m * Q + n) * v;
        const int r = sort_inv[m], l = n / (2 * R), part = (n / R) & 1, j = n % R;
        float *dst = part ? dH : dC;
        dst[((size_t)l * B + r) * R + j] = s;
    }
};

// arch2 head: d(hd) -> Dropout backward -> dL/dh of the top layer at step tmax
struct EpiHead2 {
    float *dH;
    int R;
    Drop dr;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        dH[(size_t)m * R + n] = dr.scale(NVQA_SITE_Q, (uint64_t)m * R + n) * v;
    }
};

// Fused LSTM cell forward (misc/LSTM.lua:43-59): a[4] = W_h2h h_{t-1} for gates (i,f,o,g);
// gx holds W_i2h x_t + b_i2h + b_h2h on entry and the ACTIVATED gates on exit (kept for BPTT).
// Rows >= *nrows have not started yet and keep a zero state (misc/RNNUtils.lua:136-145).
struct EpiLstmFwd {
    float *gx;           // [B][4R] of this step
    const float *c_prev; // [B][R]
    float *c, *h;        // [B][R]
    float *u_next;       // optional [B][R]: Dropout(h) = input of the next layer at this step
    const float *bias1, *bias2; // non-NULL: a[] already holds both products (two-segment K), add the
                                // biases instead of reading a precomputed input projection from gx
    const int *nrows, *sort_idx;
    int R, B, T, t, lnext_m1; // lnext_m1 = (l+1)-1 = l for the dropout index of layer l+1
    Drop dr;
    // epilogue inputs that do not depend on the product: fetched BEFORE the K loop so that their
    // L2/HBM latency hides under it (the step kernels are latency-critical)
    struct Pre { float p0, p1, p2, p3, cp; int nr; };
    __device__ __forceinline__ Pre preload(int m, int u) const
    {
        const size_t gi = (size_t)m * 4 * R + u;
        Pre q;
        if (bias1) {
            q.p0 = bias1[u] + bias2[u]; q.p1 = bias1[R + u] + bias2[R + u];
            q.p2 = bias1[2 * R + u] + bias2[2 * R + u]; q.p3 = bias1[3 * R + u] + bias2[3 * R + u];
        } else {
            q.p0 = gx[gi]; q.p1 = gx[gi + R]; q.p2 = gx[gi + 2 * R]; q.p3 = gx[gi + 3 * R];
        }
        q.cp = c_prev[(size_t)m * R + u];
        q.nr = *nrows;
        return q;
    }
    __device__ __forceinline__ void operator()(int m, int u, const float (&a)[4]) const { (*this)(m, u, a, preload(m, u)); }
    __device__ __forceinline__ void operator()(int m, int u, const float (&a)[4], const Pre &q) const
    {
        const size_t gi = (size_t)m * 4 * R + u, si = (size_t)m * R + u;
        if (m >= q.nr) {
            gx[gi] = 0.f; gx[gi + R] = 0.f; gx[gi + 2 * R] = 0.f; gx[gi + 3 * R] = 0.f;
            c[si] = 0.f; h[si] = 0.f;
            if (u_next) u_next[si] = 0.f;
            return;
        }
        const float ig = sigmoidf_(a[0] + q.p0);
        const float fg = sigmoidf_(a[1] + q.p1);
        const float og = sigmoidf_(a[2] + q.p2);
        const float gg = tanhf_(a[3] + q.p3);
        const float cn = fg * q.cp + ig * gg;
        const float hn = og * tanhf_(cn);
        gx[gi] = ig; gx[gi + R] = fg; gx[gi + 2 * R] = og; gx[gi + 3 * R] = gg;
        c[si] = cn; h[si] = hn;
        if (u_next) {
            const uint64_t idx = ((((uint64_t)lnext_m1) * B + sort_idx[m]) * T + t) * R + u;
            u_next[si] = dr.scale(NVQA_SITE_LSTM, idx) * hn;
        }
    }
};

// Fused LSTM cell backward at step s (nngraph backward of misc/LSTM.lua:43-59, driven by
// misc/RNNUtils.lua:182-209): v = (dG_{s+1} W_h2h)[m][u] is the recurrent part of dh_s.
// gates holds the activated gates of step s on entry and d(pre-activations) on exit.
struct EpiLstmBwd {
    float *gates;              // [B][4R] of step s (in: i,f,o,g ; out: da)
    const float *c_prev, *c;   // [B][R] cell before / after step s
    float *dc;                 // [B][R] carried cell gradient (in: dL/dc_s from s+1, out: dL/dc_{s-1})
    const float *dh_ext, *dh_ext2; // optional extra dL/dh_s terms (dh_ext unused now; dh_ext2 = head)
    const int *tlast;              // optional: dh_ext2 enters only at step *tlast (arch2: tmax-1)
    const int *nrows;
    int R;
    // two-accumulator form (SEG == 2): v2 = (dG^{l+1}_s W_i2h^{l+1})[m][u], the gradient reaching
    // h^l_s through the inter-layer Dropout of misc/LSTM.lua:37
    const int *sort_idx;
    int B, T, s, lm1;
    Drop dr;
    int has_upper; // 0 for the top layer (v2 is identically zero)
    struct Pre { float ig, fg, og, gg, c, cp, dc, dhx, dscale; int nr; };
    __device__ __forceinline__ Pre preload(int m, int u) const
    {
        const size_t gi = (size_t)m * 4 * R + u, si = (size_t)m * R + u;
        Pre q;
        q.ig = gates[gi]; q.fg = gates[gi + R]; q.og = gates[gi + 2 * R]; q.gg = gates[gi + 3 * R];
        q.c = c[si]; q.cp = c_prev[si]; q.dc = dc[si];
        q.dhx = (dh_ext ? dh_ext[si] : 0.f) + ((dh_ext2 && (!tlast || *tlast == s)) ? dh_ext2[si] : 0.f);
        q.dscale = has_upper ? dr.scale(NVQA_SITE_LSTM, ((((uint64_t)lm1) * B + sort_idx[m]) * T + s) * R + u) : 0.f;
        q.nr = *nrows;
        return q;
    }
    __device__ __forceinline__ void operator()(int z, int m, int u, float v, float v2) const { (*this)(z, m, u, v, v2, preload(m, u)); }
    __device__ __forceinline__ void operator()(int z, int m, int u, float v) const { (*this)(z, m, u, v, 0.f, preload(m, u)); }
    __device__ __forceinline__ void operator()(int, int m, int u, float v, float v2, const Pre &q) const
    {
        const size_t gi = (size_t)m * 4 * R + u, si = (size_t)m * R + u;
        if (m >= q.nr) { // gradient rows of not-yet-started questions are dropped (RNNUtils.lua:192-196)
            gates[gi] = 0.f; gates[gi + R] = 0.f; gates[gi + 2 * R] = 0.f; gates[gi + 3 * R] = 0.f;
            dc[si] = 0.f;
            return;
        }
        const float dh = v + q.dscale * v2 + q.dhx;
        const float ig = q.ig, fg = q.fg, og = q.og, gg = q.gg;
        const float tc = tanhf_(q.c);
        const float dcv = q.dc + dh * og * (1.0f - tc * tc);
        gates[gi] = dcv * gg * ig * (1.0f - ig);
        gates[gi + R] = dcv * q.cp * fg * (1.0f - fg);
        gates[gi + 2 * R] = dh * tc * og * (1.0f - og);
        gates[gi + 3 * R] = dcv * ig * (1.0f - gg * gg);
        dc[si] = dcv * fg;
    }
};

// epilogues with a preload() step (see EpiLstmFwd::Pre); vec4: a tile-native 16-byte store form (no epilogue uses it now)
template <class E> struct EpiTraits { static constexpr bool prefetch = false; static constexpr bool vec4 = false; struct Pre {}; };
template <> struct EpiTraits<EpiLstmFwd> { static constexpr bool prefetch = true; static constexpr bool vec4 = false; typedef EpiLstmFwd::Pre Pre; };
template <> struct EpiTraits<EpiLstmBwd> { static constexpr bool prefetch = true; static constexpr bool vec4 = false; typedef EpiLstmBwd::Pre Pre; };

} // namespace nvqa
