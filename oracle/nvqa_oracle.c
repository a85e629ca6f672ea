/*
 * nvqa_oracle.c -- CPU restatement of the novel-vqa training step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (Lua/Torch7) cannot run in the build
 * environment and ships no tests, fixtures or golden vectors, so this file is
 * a restatement of the equations read from the cited lines; it is
 * cross-checked against an independent autograd model (tests/ref_autograd.py),
 * against PyTorch's own nn / optim modules -- LSTMCell, Linear, Embedding,
 * CrossEntropyLoss, optim.RMSprop, the descendants of the Torch7 packages the
 * reference calls (tests/test_oracle_torch_modules.py, agreement to 1e-10) --
 * and against finite differences, not against Torch7 output.
 *
 * Follows (paths relative to the reference checkout):
 *   arch1 step     002_train_vqa_arch1/002_train_baseline.lua:272-335
 *   packing        002_train_vqa_arch1/misc/RNNUtils.lua:84-124  (sort, time-major pack)
 *   embedding      002_train_baseline.lua:141-144 (Linear(V,E) on one-hot == column gather + bias)
 *   LSTM cell      002_train_vqa_arch1/misc/LSTM.lua:12-73 (gate order in, forget, out, transform)
 *   unroll / BPTT  misc/RNNUtils.lua:128-154, 182-209 (growing batch, zero padding)
 *   fusion + head  misc/netdef.lua:6-14, 002_train_baseline.lua:151-154
 *   criterion      nn.CrossEntropyCriterion, 002_train_baseline.lua:157,308-310
 *   optimiser      misc/rmsprop_lrscale.lua:13-34 (without the lrs factor)
 *   arch2 step     003_train_vqa_arch2/002_train_baseline.lua:277-333,
 *                  misc/Encoder_lstm.lua:152-263, misc/LSTM_encoder.lua:5-57
 *
 * Torch7 semantics assumed (the un-pinned part): nn.Linear y = x W^T + b with
 * W [out x in]; nn.Dropout(p) v2 y = x*mask/(1-p); CrossEntropy = mean over
 * the batch of -log softmax[label]; nn.LookupTable row gather / scatter-add.
 *
 * Compile: gcc -O3 -march=native -fopenmp -shared -fPIC [-DORACLE_REAL=double]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nvqa.h"
#include "../include/nvqa_layout.h"
#include "../include/nvqa_rng.h"

#ifndef ORACLE_REAL
#define ORACLE_REAL float
#endif
typedef ORACLE_REAL real;

int oracle_real_bytes(void) { return (int)sizeof(real); }

static real *zalloc(size_t n) { return (real *)calloc(n ? n : 1, sizeof(real)); }

static real drop_scale(const nvqa_dropout *dr, uint32_t site, uint64_t idx)
{
    if (!dr || dr->mode == 0) return (real)1;
    return (real)nvqa_dropout_scale(dr->seed, dr->step, site, idx, dr->p, 1.0f / (1.0f - dr->p));
}

/* y[n x out] = x[n x in] W^T + b            (nn.Linear forward) */
/* bf16 operand mode (BASELINE config "arch2 ... bf16", nvqa_set_precision): every operand of a dense
 * product is rounded to bf16 (round-to-nearest-even, as v_cvt_pk_bf16_f32 does) before it is multiplied;
 * sums, biases, bias gradients and everything else stay in `real`. */
static int g_bf16 = 0;
void oracle_set_precision(int bf16) { g_bf16 = bf16; }
static real bf16_round(real v)
{
    float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v; /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return (real)f;
}
/* rounded copy of an operand (NULL when the mode is off: the caller then uses the original) */
static real *bf16_copy(const real *p, size_t n)
{
    if (!g_bf16) return NULL;
    real *q = (real *)malloc((n ? n : 1) * sizeof(real));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) q[i] = bf16_round(p[i]);
    return q;
}

static void lin_fwd(int n, int out, int in, const real *x, const real *W, const real *b, real *y)
{
    real *xb = bf16_copy(x, (size_t)n * in), *Wb = bf16_copy(W, (size_t)out * in);
    if (xb) { x = xb; W = Wb; }
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < n; ++r)
        for (int o = 0; o < out; ++o) {
            const real *xr = x + (size_t)r * in, *w = W + (size_t)o * in;
            real acc = 0;
#pragma omp simd reduction(+ : acc)
            for (int k = 0; k < in; ++k) acc += xr[k] * w[k];
            y[(size_t)r * out + o] = acc + (b ? b[o] : (real)0);
        }
    free(xb);
    free(Wb);
}

/* dx[n x in] (+)= dy[n x out] W            (nn.Linear updateGradInput) */
static void lin_bwd_dx(int n, int out, int in, const real *dy, const real *W, real *dx, int accumulate)
{
    real *gb = bf16_copy(dy, (size_t)n * out), *Wb = bf16_copy(W, (size_t)out * in);
    if (gb) { dy = gb; W = Wb; }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        real *d = dx + (size_t)r * in;
        if (!accumulate) memset(d, 0, sizeof(real) * (size_t)in);
        for (int o = 0; o < out; ++o) {
            const real g = dy[(size_t)r * out + o];
            const real *w = W + (size_t)o * in;
#pragma omp simd
            for (int k = 0; k < in; ++k) d[k] += g * w[k];
        }
    }
    free(gb);
    free(Wb);
}

/* dW[out x in] += dy^T x ; db[out] += colsum(dy)   (nn.Linear accGradParameters) */
static void lin_bwd_dw(int n, int out, int in, const real *dy, const real *x, real *dW, real *db)
{
    real *gb = bf16_copy(dy, (size_t)n * out), *xb = bf16_copy(x, (size_t)n * in);
    const real *dyq = gb ? gb : dy; /* the bias gradient is a plain column sum of the unrounded dy */
    if (xb) x = xb;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < out; ++o) {
        real *w = dW + (size_t)o * in;
        real bs = 0;
        for (int r = 0; r < n; ++r) {
            const real g = dyq[(size_t)r * out + o];
            const real *xr = x + (size_t)r * in;
            bs += dy[(size_t)r * out + o];
#pragma omp simd
            for (int k = 0; k < in; ++k) w[k] += g * xr[k];
        }
        if (db) db[o] += bs;
    }
    free(gb);
    free(xb);
}

static real sigm(real x) { return (real)1 / ((real)1 + (real)exp(-(double)x)); }

/* One LSTM cell forward for n rows (misc/LSTM.lua:41-59).  a [n x 4R] holds the
 * pre-activations on entry and the activated gates (i,f,o,g) on exit. */
static void cell_fwd(int n, int R, real *a, const real *c_prev, real *c, real *h)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        real *ar = a + (size_t)r * 4 * R;
        for (int j = 0; j < R; ++j) {
            const real ig = sigm(ar[j]), fg = sigm(ar[R + j]), og = sigm(ar[2 * R + j]);
            const real gg = (real)tanh((double)ar[3 * R + j]);
            const real cn = fg * c_prev[(size_t)r * R + j] + ig * gg;
            ar[j] = ig; ar[R + j] = fg; ar[2 * R + j] = og; ar[3 * R + j] = gg;
            c[(size_t)r * R + j] = cn;
            h[(size_t)r * R + j] = og * (real)tanh((double)cn);
        }
    }
}

/* Cell backward: given dh, dc (incoming), activated gates, c_prev, c -> da [n x 4R], dc_prev */
static void cell_bwd(int n, int R, const real *gates, const real *c_prev, const real *c,
                     const real *dh, const real *dc_in, real *da, real *dc_prev)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        const real *g4 = gates + (size_t)r * 4 * R;
        real *d4 = da + (size_t)r * 4 * R;
        for (int j = 0; j < R; ++j) {
            const real ig = g4[j], fg = g4[R + j], og = g4[2 * R + j], gg = g4[3 * R + j];
            const real tc = (real)tanh((double)c[(size_t)r * R + j]);
            const real dhv = dh[(size_t)r * R + j];
            const real dcv = dc_in[(size_t)r * R + j] + dhv * og * ((real)1 - tc * tc);
            d4[j] = dcv * gg * ig * ((real)1 - ig);
            d4[R + j] = dcv * c_prev[(size_t)r * R + j] * fg * ((real)1 - fg);
            d4[2 * R + j] = dhv * tc * og * ((real)1 - og);
            d4[3 * R + j] = dcv * ig * ((real)1 - gg * gg);
            dc_prev[(size_t)r * R + j] = dcv * fg;
        }
    }
}

/* mean cross-entropy + dscores = (softmax - onehot)/B ; labels 1-based */
static real softmax_ce(int B, int A, const real *scores, const int32_t *labels, real *dscores,
                       int32_t *argmax)
{
    double loss = 0;
    for (int b = 0; b < B; ++b) {
        const real *s = scores + (size_t)b * A;
        real mx = s[0];
        int am = 0;
        for (int a = 1; a < A; ++a)
            if (s[a] > mx) { mx = s[a]; am = a; }
        if (argmax) argmax[b] = am + 1; /* torch.max returns the first maximal index */
        if (!labels) continue;
        double sum = 0;
        for (int a = 0; a < A; ++a) sum += exp((double)(s[a] - mx));
        const double lse = (double)mx + log(sum);
        loss += lse - (double)s[labels[b] - 1];
        if (dscores)
            for (int a = 0; a < A; ++a)
                dscores[(size_t)b * A + a] =
                    (real)((exp((double)s[a] - lse) - (a == labels[b] - 1 ? 1.0 : 0.0)) / B);
    }
    return (real)(loss / B);
}

/* ------------------------------------------------------------------------- */
/* arch1                                                                      */
/* ------------------------------------------------------------------------- */
/* train != 0: training-mode forward (dropout per `dr`) + backward into grads
 * (flat, reference layout, UNclamped).  train == 0: evaluate mode, forward
 * only.  scores [B x A] / argmax [B] optional outputs.  Returns 0 / <0. */
/* fusion graph: 0 = netdef.AxB (misc/netdef.lua:6-14, the baseline), 1 = netdef.AskipB
 * (misc/netdef.lua:16-25: output = qc + qc (*) ic, used by 003_train_ae_based_wp.lua:151) */
static int g_fusion_askip = 0;
void oracle_set_fusion(int askip) { g_fusion_askip = askip; }

int oracle_arch1_step(const nvqa_dims *d, const real *params, const int32_t *tok,
                      const int32_t *len, const real *img, const int32_t *labels,
                      const nvqa_dropout *dr_in, int train, real *loss_out, real *grads,
                      real *scores_out, int32_t *argmax_out)
{
    nvqa_layout lo;
    if (d->arch != NVQA_ARCH1 || nvqa_layout_init_fusion(d, g_fusion_askip, &lo)) return -1;
    const int B = d->B, T = d->T, V = d->V, E = d->E, R = d->R, L = d->L, I = d->I, C = d->C,
              A = d->A;
    const int Q = 2 * R * L;
    const nvqa_dropout *dr = train ? dr_in : NULL;

    /* ---- packing: sort_encoding_onehot_right_align (RNNUtils.lua:84-124) ---- */
    /* stable descending sort by length (torch.sort is unstable; order of ties unpinned, Q6) */
    int *sidx = (int *)malloc(sizeof(int) * B), *sinv = (int *)malloc(sizeof(int) * B);
    {
        int p = 0;
        for (int l = T; l >= 0; --l)
            for (int b = 0; b < B; ++b)
                if (len[b] == l) sidx[p++] = b;
        if (p != B) { free(sidx); free(sinv); return -2; } /* a length outside [0,T] */
        for (int r = 0; r < B; ++r) sinv[sidx[r]] = r;
    }
    const int Lmax = len[sidx[0]];
    if (Lmax < 1) { free(sidx); free(sinv); return -2; }
    int *nb = (int *)malloc(sizeof(int) * Lmax), *off = (int *)malloc(sizeof(int) * (Lmax + 1));
    off[0] = 0;
    for (int i = 0; i < Lmax; ++i) {
        int n = 0;
        for (int r = 0; r < B; ++r) n += len[sidx[r]] >= Lmax - i;
        nb[i] = n;
        off[i + 1] = off[i] + n;
    }
    const int NP = off[Lmax]; /* sum of lengths */

    /* ---- embedding: tanh(dropout(W_e[:,tok] + b_e)) (002_train_baseline.lua:141-144,300) ---- */
    const real *We = params + lo.w_e, *be = params + lo.b_e;
    real *X = zalloc((size_t)NP * E), *De = zalloc((size_t)NP * E);
    int *ptok = (int *)malloc(sizeof(int) * (NP ? NP : 1));
    for (int i = 0; i < Lmax; ++i) {
        const int col = T - Lmax + i;
        for (int r = 0; r < nb[i]; ++r) {
            const int b = sidx[r], k = off[i] + r, w = tok[(size_t)b * T + col];
            if (w < 1 || w > V) { return -3; }
            ptok[k] = w - 1;
            for (int e = 0; e < E; ++e) {
                const real s = drop_scale(dr, NVQA_SITE_EMB, ((uint64_t)b * T + col) * E + e);
                De[(size_t)k * E + e] = s;
                X[(size_t)k * E + e] = (real)tanh((double)(s * (We[(size_t)e * V + (w - 1)] + be[e])));
            }
        }
    }

    /* ---- LSTM unroll: rnn_forward (RNNUtils.lua:128-154) ---- */
    /* Cs/Hs [L][Lmax+1][B][R], zero rows = "not started yet" padding (:136-145) */
    const size_t SB = (size_t)B * R;
    real *Cs = zalloc((size_t)L * (Lmax + 1) * SB), *Hs = zalloc((size_t)L * (Lmax + 1) * SB);
    real *G = zalloc((size_t)L * NP * 4 * R);   /* activated gates per packed row */
    real *U = zalloc((size_t)L * NP * R);       /* dropped-out input of layers >= 2 */
    real *Dl = zalloc((size_t)L * NP * R);      /* its dropout scales */
    real *tmp = zalloc((size_t)B * 4 * R);
#define CS(l, i) (Cs + ((size_t)(l) * (Lmax + 1) + (i)) * SB)
#define HS(l, i) (Hs + ((size_t)(l) * (Lmax + 1) + (i)) * SB)
    for (int i = 0; i < Lmax; ++i) {
        const int n = nb[i], col = T - Lmax + i;
        for (int l = 0; l < L; ++l) {
            const int in = l == 0 ? E : R;
            const real *u;
            if (l == 0) {
                u = X + (size_t)off[i] * E;
            } else {
                real *ul = U + ((size_t)l * NP + off[i]) * R, *dl = Dl + ((size_t)l * NP + off[i]) * R;
                const real *hb = HS(l - 1, i + 1);
                for (int r = 0; r < n; ++r)
                    for (int j = 0; j < R; ++j) {
                        const real s = drop_scale(
                            dr, NVQA_SITE_LSTM,
                            ((((uint64_t)(l - 1)) * B + sidx[r]) * T + col) * R + j);
                        dl[(size_t)r * R + j] = s;
                        ul[(size_t)r * R + j] = s * hb[(size_t)r * R + j];
                    }
                u = ul;
            }
            real *a = G + ((size_t)l * NP + off[i]) * 4 * R;
            lin_fwd(n, 4 * R, in, u, params + lo.w_i2h[l], params + lo.b_i2h[l], a);
            lin_fwd(n, 4 * R, R, HS(l, i), params + lo.w_h2h[l], params + lo.b_h2h[l], tmp);
            for (size_t z = 0; z < (size_t)n * 4 * R; ++z) a[z] += tmp[z]; /* CAddTable */
            cell_fwd(n, R, a, CS(l, i), CS(l, i + 1), HS(l, i + 1));
        }
    }

    /* ---- question vector, un-sort (002_train_baseline.lua:306) ---- */
    real *q = zalloc((size_t)B * Q), *qd = zalloc((size_t)B * Q), *Dq = zalloc((size_t)B * Q);
    for (int b = 0; b < B; ++b) {
        const int r = sinv[b];
        for (int l = 0; l < L; ++l) {
            memcpy(q + (size_t)b * Q + (size_t)2 * l * R, CS(l, Lmax) + (size_t)r * R, sizeof(real) * R);
            memcpy(q + (size_t)b * Q + (size_t)(2 * l + 1) * R, HS(l, Lmax) + (size_t)r * R, sizeof(real) * R);
        }
        for (int j = 0; j < Q; ++j) {
            Dq[(size_t)b * Q + j] = drop_scale(dr, NVQA_SITE_Q, (uint64_t)b * Q + j);
            qd[(size_t)b * Q + j] = Dq[(size_t)b * Q + j] * q[(size_t)b * Q + j];
        }
    }
    /* ---- AxB fusion + classifier (netdef.lua:6-14, 002_train_baseline.lua:151-154) ---- */
    real *vd = zalloc((size_t)B * I), *Dv = zalloc((size_t)B * I);
    for (size_t z = 0; z < (size_t)B * I; ++z) {
        Dv[z] = drop_scale(dr, NVQA_SITE_V, z);
        vd[z] = Dv[z] * img[z];
    }
    /* ZW: width of what the classifier reads: C, or 2C for netdef.A_B = JoinTable(2)({qc, ic}) (netdef.lua:27-35) */
    const int join = g_fusion_askip == 2, ZW = join ? 2 * C : C;
    real *qc = zalloc((size_t)B * C), *ic = zalloc((size_t)B * C), *zd = zalloc((size_t)B * ZW),
         *Dz = zalloc((size_t)B * ZW);
    lin_fwd(B, C, Q, qd, params + lo.w_q, params + lo.b_q, qc);
    lin_fwd(B, C, I, vd, params + lo.w_v, params + lo.b_v, ic);
    for (size_t z = 0; z < (size_t)B * C; ++z) {
        qc[z] = (real)tanh((double)qc[z]);
        ic[z] = (real)tanh((double)ic[z]);
    }
    if (join) {
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < ZW; ++j) {
                const size_t z = (size_t)b * ZW + j; /* the Dropout after the fusion module sees the joined [B x 2C] tensor */
                Dz[z] = drop_scale(dr, NVQA_SITE_Z, z);
                zd[z] = Dz[z] * (j < C ? qc[(size_t)b * C + j] : ic[(size_t)b * C + j - C]);
            }
    } else {
        for (size_t z = 0; z < (size_t)B * C; ++z) {
            Dz[z] = drop_scale(dr, NVQA_SITE_Z, z);
            zd[z] = Dz[z] * (g_fusion_askip ? qc[z] + qc[z] * ic[z] : qc[z] * ic[z]);
        }
    }
    real *scores = zalloc((size_t)B * A), *dscores = zalloc((size_t)B * A);
    lin_fwd(B, A, ZW, zd, params + lo.w_o, params + lo.b_o, scores);
    const real loss = softmax_ce(B, A, scores, labels, (train && labels) ? dscores : NULL, argmax_out);
    if (loss_out) *loss_out = loss;
    if (scores_out) memcpy(scores_out, scores, sizeof(real) * (size_t)B * A);

    if (train && grads && labels) {
        memset(grads, 0, sizeof(real) * lo.total);
        /* ---- multimodal backward (002_train_baseline.lua:312) ---- */
        real *dzd = zalloc((size_t)B * ZW), *dqc = zalloc((size_t)B * C), *dic = zalloc((size_t)B * C);
        lin_bwd_dw(B, A, ZW, dscores, zd, grads + lo.w_o, grads + lo.b_o);
        lin_bwd_dx(B, A, ZW, dscores, params + lo.w_o, dzd, 0);
        if (join) {
            for (int b = 0; b < B; ++b)
                for (int j = 0; j < C; ++j) {
                    const size_t z = (size_t)b * C + j, zq = (size_t)b * ZW + j, zi = zq + C;
                    dqc[z] = Dz[zq] * dzd[zq] * ((real)1 - qc[z] * qc[z]);
                    dic[z] = Dz[zi] * dzd[zi] * ((real)1 - ic[z] * ic[z]);
                }
        } else {
            for (size_t z = 0; z < (size_t)B * C; ++z) {
                const real dz = Dz[z] * dzd[z];
                dqc[z] = dz * (g_fusion_askip ? (real)1 + ic[z] : ic[z]) * ((real)1 - qc[z] * qc[z]);
                dic[z] = dz * qc[z] * ((real)1 - ic[z] * ic[z]);
            }
        }
        lin_bwd_dw(B, C, Q, dqc, qd, grads + lo.w_q, grads + lo.b_q);
        lin_bwd_dw(B, C, I, dic, vd, grads + lo.w_v, grads + lo.b_v);
        real *dqd = zalloc((size_t)B * Q);
        lin_bwd_dx(B, C, Q, dqc, params + lo.w_q, dqd, 0);
        /* ---- re-sort (:313) and BPTT (RNNUtils.lua:182-209) ---- */
        real *dC = zalloc((size_t)L * SB), *dH = zalloc((size_t)L * SB);
        for (int r = 0; r < B; ++r) {
            const int b = sidx[r];
            for (int l = 0; l < L; ++l)
                for (int j = 0; j < R; ++j) {
                    dC[(size_t)l * SB + (size_t)r * R + j] =
                        Dq[(size_t)b * Q + 2 * l * R + j] * dqd[(size_t)b * Q + 2 * l * R + j];
                    dH[(size_t)l * SB + (size_t)r * R + j] =
                        Dq[(size_t)b * Q + (2 * l + 1) * R + j] * dqd[(size_t)b * Q + (2 * l + 1) * R + j];
                }
        }
        real *da = zalloc((size_t)B * 4 * R), *du = zalloc((size_t)B * (R > E ? R : E));
        real *dX = zalloc((size_t)NP * E), *dcp = zalloc(SB), *dhp = zalloc(SB);
        for (int i = Lmax - 1; i >= 0; --i) {
            const int n = nb[i];
            /* rows >= n of the carried gradient are dropped (RNNUtils.lua:192-196) */
            for (int l = L - 1; l >= 0; --l) {
                const int in = l == 0 ? E : R;
                const real *gates = G + ((size_t)l * NP + off[i]) * 4 * R;
                cell_bwd(n, R, gates, CS(l, i), CS(l, i + 1), dH + (size_t)l * SB, dC + (size_t)l * SB,
                         da, dcp);
                const real *u = l == 0 ? X + (size_t)off[i] * E : U + ((size_t)l * NP + off[i]) * R;
                lin_bwd_dw(n, 4 * R, in, da, u, grads + lo.w_i2h[l], grads + lo.b_i2h[l]);
                lin_bwd_dw(n, 4 * R, R, da, HS(l, i), grads + lo.w_h2h[l], grads + lo.b_h2h[l]);
                lin_bwd_dx(n, 4 * R, R, da, params + lo.w_h2h[l], dhp, 0);
                lin_bwd_dx(n, 4 * R, in, da, params + lo.w_i2h[l], du, 0);
                memcpy(dC + (size_t)l * SB, dcp, sizeof(real) * (size_t)n * R);
                memcpy(dH + (size_t)l * SB, dhp, sizeof(real) * (size_t)n * R);
                if (l > 0) {
                    const real *dl = Dl + ((size_t)l * NP + off[i]) * R;
                    real *dhl = dH + (size_t)(l - 1) * SB;
                    for (size_t z = 0; z < (size_t)n * R; ++z) dhl[z] += dl[z] * du[z];
                } else {
                    memcpy(dX + (size_t)off[i] * E, du, sizeof(real) * (size_t)n * E);
                }
            }
        }
        /* ---- embedding backward (:319-320): tanh', dropout scale, Linear on one-hot ---- */
        real *gWe = grads + lo.w_e, *gbe = grads + lo.b_e;
        for (int k = 0; k < NP; ++k)
            for (int e = 0; e < E; ++e) {
                const real x = X[(size_t)k * E + e];
                const real dp = De[(size_t)k * E + e] * (dX[(size_t)k * E + e] * ((real)1 - x * x));
                gWe[(size_t)e * V + ptok[k]] += dp;
                gbe[e] += dp;
            }
        free(dzd); free(dqc); free(dic); free(dqd); free(dC); free(dH); free(da); free(du);
        free(dX); free(dcp); free(dhp);
    }
#undef CS
#undef HS
    free(sidx); free(sinv); free(nb); free(off); free(X); free(De); free(ptok); free(Cs); free(Hs);
    free(G); free(U); free(Dl); free(tmp); free(q); free(qd); free(Dq); free(vd); free(Dv); free(qc);
    free(ic); free(zd); free(Dz); free(scores); free(dscores);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* arch2 (003_train_vqa_arch2): image-as-first-token encoder                  */
/* ------------------------------------------------------------------------- */
/* tok [B x T] LEFT-aligned, 0 = null (the reference's seq is its [T x B] transpose,
 * 002_train_baseline.lua:216).
 *
 * Two things the reference's arch2 code does that its authors cannot have intended are reproduced only on request
 * (oracle_set_ref_quirks; default 0 = the model as designed, a fresh single step is the same either way for Q1):
 *
 *  bit 0, Q1 "aliased init state" (misc/Encoder_lstm.lua:238-239 with :30-47 and :164).  updateGradInput does
 *    `dstate_enc = {[tmax] = self.init_state_enc}; dstate_enc[tmax][num_state] = gradOutput`: the Lua TABLE of
 *    initial-state tensors is aliased, so entry num_state = 2L (the top layer's h0, misc/LSTM_encoder.lua:11-14,54-55)
 *    becomes the gradOutput tensor itself -- multimodal_net.gradInput, i.e. the nn.Dropout module's gradInput buffer.
 *    _createInitState never re-zeroes it while the batch size stays the same.  Consequences, reproduced here:
 *      - every forward after the first backward (training and validate() alike) starts the top layer from
 *        h0 = dL/d(encoder output) as the LAST backward left it (dropout mask and 1/(1-p) included);
 *      - inside a backward, that same tensor already holds the CURRENT step's gradient when the step-1 clone runs
 *        accGradParameters, so dW_h2h of the top layer receives dgates_1^T x (current dL/dh), not the h0 the
 *        forward pass used.  (The gradient flowing INTO h0 is discarded either way.)
 *  bit 1, Q11 "lookup table never trained" (misc/Encoder_lstm.lua:49-58 with 002_train_baseline.lua:186,273).
 *    encoder_model:getParameters() flattens self.lookup_table BEFORE createClones(); createClones builds
 *    lookup_tables_encoder[1] = self.lookup_table:clone('weight') -- weight shared, gradWeight NOT -- and the later
 *    clones share gradWeight with clone [1].  backward accumulates into the clones' gradWeight, which is not part
 *    of encoder_dw_q: the lookup slice of the flat gradient is zeroed every iteration (:294) and never written, so
 *    W_lk only ever sees weightDecay through RMSprop.  With this bit the lookup gradient is returned as zero. */
#define ORACLE_QUIRK_H0 1
#define ORACLE_QUIRK_LOOKUP 2
static int g_quirks = 0;
static real *g_h0 = NULL; /* the aliased tensor: survives between steps */
static size_t g_h0_n = 0;
void oracle_set_ref_quirks(int flags)
{
    g_quirks = flags;
    free(g_h0);
    g_h0 = NULL;
    g_h0_n = 0;
}

int oracle_arch2_step(const nvqa_dims *d, const real *params, const int32_t *tok, const real *img,
                      const int32_t *labels, const nvqa_dropout *dr_in, int train, real *loss_out,
                      real *grads, real *scores_out, int32_t *argmax_out)
{
    nvqa_layout lo;
    if (d->arch != NVQA_ARCH2 || nvqa_layout_init(d, &lo)) return -1;
    const int B = d->B, T = d->T, V = d->V, E = d->E, R = d->R, L = d->L, I = d->I, A = d->A;
    const int TS = T + 2;
    const nvqa_dropout *dr = train ? dr_in : NULL;
    const size_t SB = (size_t)B * R;

    /* tmax: stop at the first all-null time row (Encoder_lstm.lua:185-189,219) */
    int tmax = 2;
    for (int t = 3; t <= TS; ++t) {
        long s = 0;
        for (int b = 0; b < B; ++b) s += tok[(size_t)b * T + (t - 3)];
        if (s == 0) break;
        tmax = t;
    }
    /* inputs x_t [tmax][B][E] and token rows */
    real *Xs = zalloc((size_t)tmax * B * E);
    int *rows = (int *)malloc(sizeof(int) * (size_t)tmax * B);
    lin_fwd(B, E, I, img, params + lo.w_p, params + lo.b_p, Xs); /* t=1: cnn_projection (:308) */
    const real *Wlk = params + lo.w_lk;
    for (int t = 2; t <= tmax; ++t)
        for (int b = 0; b < B; ++b) {
            int w = t == 2 ? V + 1 : tok[(size_t)b * T + (t - 3)];
            if (w == 0) w = 1; /* null -> token 1 (Encoder_lstm.lua:197) */
            if (w < 1 || w > V + 1) return -3;
            rows[(size_t)(t - 1) * B + b] = w - 1;
            memcpy(Xs + ((size_t)(t - 1) * B + b) * E, Wlk + (size_t)(w - 1) * E, sizeof(real) * E);
        }
    real *Cs = zalloc((size_t)L * (tmax + 1) * SB), *Hs = zalloc((size_t)L * (tmax + 1) * SB);
    real *G = zalloc((size_t)L * tmax * B * 4 * R), *U = zalloc((size_t)L * tmax * SB),
         *Dl = zalloc((size_t)L * tmax * SB), *tmp = zalloc((size_t)B * 4 * R);
#define CS(l, i) (Cs + ((size_t)(l) * (tmax + 1) + (i)) * SB)
#define HS(l, i) (Hs + ((size_t)(l) * (tmax + 1) + (i)) * SB)
    if (g_quirks & ORACLE_QUIRK_H0) { /* Q1: top-layer h0 = what the last backward left in the aliased tensor */
        if (g_h0_n != SB) { /* first use, or the batch size changed (:37-39 resize + zero) */
            free(g_h0);
            g_h0 = zalloc(SB);
            g_h0_n = SB;
        }
        memcpy(HS(L - 1, 0), g_h0, sizeof(real) * SB);
    }
    for (int i = 0; i < tmax; ++i)
        for (int l = 0; l < L; ++l) {
            const int in = l == 0 ? E : R;
            const real *u;
            if (l == 0) {
                u = Xs + (size_t)i * B * E;
            } else {
                real *ul = U + ((size_t)l * tmax + i) * SB, *dl = Dl + ((size_t)l * tmax + i) * SB;
                const real *hb = HS(l - 1, i + 1);
                for (int b = 0; b < B; ++b)
                    for (int j = 0; j < R; ++j) {
                        const real s = drop_scale(dr, NVQA_SITE_LSTM,
                                                  ((((uint64_t)(l - 1)) * B + b) * TS + i) * R + j);
                        dl[(size_t)b * R + j] = s;
                        ul[(size_t)b * R + j] = s * hb[(size_t)b * R + j];
                    }
                u = ul;
            }
            real *a = G + ((size_t)l * tmax + i) * B * 4 * R;
            lin_fwd(B, 4 * R, in, u, params + lo.w_i2h[l], params + lo.b_i2h[l], a);
            lin_fwd(B, 4 * R, R, HS(l, i), params + lo.w_h2h[l], params + lo.b_h2h[l], tmp);
            for (size_t z = 0; z < (size_t)B * 4 * R; ++z) a[z] += tmp[z];
            cell_fwd(B, R, a, CS(l, i), CS(l, i + 1), HS(l, i + 1));
        }
    /* head: Dropout(0.5) -> Linear(R, A) (002_train_baseline.lua:162-164,313) */
    const real *hout = HS(L - 1, tmax);
    real *hd = zalloc(SB), *Dh = zalloc(SB);
    for (size_t z = 0; z < SB; ++z) {
        Dh[z] = drop_scale(dr, NVQA_SITE_Q, z);
        hd[z] = Dh[z] * hout[z];
    }
    real *scores = zalloc((size_t)B * A), *dscores = zalloc((size_t)B * A);
    lin_fwd(B, A, R, hd, params + lo.w_o, params + lo.b_o, scores);
    const real loss = softmax_ce(B, A, scores, labels, (train && labels) ? dscores : NULL, argmax_out);
    if (loss_out) *loss_out = loss;
    if (scores_out) memcpy(scores_out, scores, sizeof(real) * (size_t)B * A);

    if (train && grads && labels) {
        memset(grads, 0, sizeof(real) * lo.total);
        real *dhd = zalloc(SB);
        lin_bwd_dw(B, A, R, dscores, hd, grads + lo.w_o, grads + lo.b_o);
        lin_bwd_dx(B, A, R, dscores, params + lo.w_o, dhd, 0);
        real *dC = zalloc((size_t)L * SB), *dH = zalloc((size_t)L * SB);
        for (size_t z = 0; z < SB; ++z) dH[(size_t)(L - 1) * SB + z] = Dh[z] * dhd[z];
        if (g_quirks & ORACLE_QUIRK_H0) { /* Q1: the aliased h0 tensor now IS this gradient (see above) */
            memcpy(g_h0, dH + (size_t)(L - 1) * SB, sizeof(real) * SB);
            memcpy(HS(L - 1, 0), g_h0, sizeof(real) * SB);
        }
        real *da = zalloc((size_t)B * 4 * R), *du = zalloc((size_t)B * (R > E ? R : E)),
             *dcp = zalloc(SB), *dhp = zalloc(SB);
        for (int i = tmax - 1; i >= 0; --i)
            for (int l = L - 1; l >= 0; --l) {
                const int in = l == 0 ? E : R;
                const real *gates = G + ((size_t)l * tmax + i) * B * 4 * R;
                cell_bwd(B, R, gates, CS(l, i), CS(l, i + 1), dH + (size_t)l * SB, dC + (size_t)l * SB, da, dcp);
                const real *u = l == 0 ? Xs + (size_t)i * B * E : U + ((size_t)l * tmax + i) * SB;
                lin_bwd_dw(B, 4 * R, in, da, u, grads + lo.w_i2h[l], grads + lo.b_i2h[l]);
                lin_bwd_dw(B, 4 * R, R, da, HS(l, i), grads + lo.w_h2h[l], grads + lo.b_h2h[l]);
                lin_bwd_dx(B, 4 * R, R, da, params + lo.w_h2h[l], dhp, 0);
                lin_bwd_dx(B, 4 * R, in, da, params + lo.w_i2h[l], du, 0);
                memcpy(dC + (size_t)l * SB, dcp, sizeof(real) * SB);
                memcpy(dH + (size_t)l * SB, dhp, sizeof(real) * SB);
                if (l > 0) {
                    const real *dl = Dl + ((size_t)l * tmax + i) * SB;
                    real *dhl = dH + (size_t)(l - 1) * SB;
                    for (size_t z = 0; z < SB; ++z) dhl[z] += dl[z] * du[z];
                } else if (i == 0) {
                    /* dx_1 -> cnn_projection:backward (:321-322) */
                    lin_bwd_dw(B, E, I, du, img, grads + lo.w_p, grads + lo.b_p);
                } else if (!(g_quirks & ORACLE_QUIRK_LOOKUP)) {
                    /* LookupTable accGradParameters into the shared gradWeight (Encoder_lstm.lua:256) */
                    real *gW = grads + lo.w_lk;
                    for (int b = 0; b < B; ++b) {
                        real *row = gW + (size_t)rows[(size_t)i * B + b] * E;
                        for (int e = 0; e < E; ++e) row[e] += du[(size_t)b * E + e];
                    }
                }
            }
        free(dhd); free(dC); free(dH); free(da); free(du); free(dcp); free(dhp);
    }
#undef CS
#undef HS
    free(Xs); free(rows); free(Cs); free(Hs); free(G); free(U); free(Dl); free(tmp); free(hd);
    free(Dh); free(scores); free(dscores);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* clamp + weight decay + RMSprop (002_train_baseline.lua:329,408;            */
/* misc/rmsprop_lrscale.lua:16-34; arch2 wd after the clamp, Q7)              */
/* ------------------------------------------------------------------------- */
void oracle_rmsprop(size_t n, real *x, real *g, real *m, real lr, real alpha, real eps, real wd,
                    real clamp)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        real gi = g[i];
        if (clamp > 0) gi = gi > clamp ? clamp : (gi < -clamp ? -clamp : gi);
        if (wd != 0) gi += wd * x[i];
        g[i] = gi;
        const real mi = alpha * m[i] + ((real)1 - alpha) * gi * gi;
        m[i] = mi;
        x[i] += -lr * gi / ((real)sqrt((double)mi) + eps);
    }
}

/* onehot(words) * W^T + b, the literal form of the arch1 embedding
 * (RNNUtils.lua:42-53 + nn.Linear); used by tests to show it equals the gather. */
void oracle_onehot_linear(int n, int V, int E, const int32_t *words /*1-based*/, const real *We,
                          const real *be, real *out)
{
    real *oh = zalloc((size_t)n * V);
    for (int k = 0; k < n; ++k) oh[(size_t)k * V + (words[k] - 1)] = 1;
    lin_fwd(n, E, V, oh, We, be, out);
    free(oh);
}

/* Offsets of include/nvqa_layout.h, exported so the tests can check the Python mirror
 * (oracle.layout) and the library against the one C definition. out[] order:
 * total, seg0, seg1, seg2, then per layer (w_i2h, b_i2h, w_h2h, b_h2h), then
 * w_e, b_e, w_q, b_q, w_v, b_v, w_o, b_o, w_p, b_p, w_lk.  Returns the count written. */
int oracle_layout(const nvqa_dims *d, uint64_t *out)
{
    nvqa_layout lo;
    if (nvqa_layout_init(d, &lo)) return -1;
    int n = 0;
    out[n++] = lo.total; out[n++] = lo.seg[0]; out[n++] = lo.seg[1]; out[n++] = lo.seg[2];
    for (int l = 0; l < d->L; ++l) {
        out[n++] = lo.w_i2h[l]; out[n++] = lo.b_i2h[l]; out[n++] = lo.w_h2h[l]; out[n++] = lo.b_h2h[l];
    }
    out[n++] = lo.w_e; out[n++] = lo.b_e; out[n++] = lo.w_q; out[n++] = lo.b_q; out[n++] = lo.w_v;
    out[n++] = lo.b_v; out[n++] = lo.w_o; out[n++] = lo.b_o; out[n++] = lo.w_p; out[n++] = lo.b_p;
    out[n++] = lo.w_lk;
    return n;
}

uint32_t oracle_hash32(uint64_t seed, uint64_t step, uint32_t site, uint64_t idx)
{
    return nvqa_hash32(seed, step, site, idx);
}
