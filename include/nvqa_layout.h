/*
 * nvqa_layout.h -- offsets of every tensor inside the flat parameter /
 * gradient vector at the C ABI (the vector optim.rmsprop sees:
 * 002_train_vqa_arch1/002_train_baseline.lua:183,190,328;
 * 003_train_vqa_arch2/002_train_baseline.lua:189,198,326).
 *
 * Segment order is the reference's.  Inside a segment the order produced by
 * Torch's getParameters() on an nngraph.gModule depends on nngraph's
 * forward-node order and cannot be established from the reference alone, so
 * this header DEFINES the canonical order (see nvqa.h).
 */
#ifndef NVQA_LAYOUT_H
#define NVQA_LAYOUT_H

#include "nvqa.h"

#define NVQA_MAX_LAYERS 4

typedef struct nvqa_layout {
    /* segment sizes in reference order */
    size_t seg[3];
    size_t total;
    /* LSTM layers (both archs) */
    size_t w_i2h[NVQA_MAX_LAYERS], b_i2h[NVQA_MAX_LAYERS];
    size_t w_h2h[NVQA_MAX_LAYERS], b_h2h[NVQA_MAX_LAYERS];
    int32_t in_dim[NVQA_MAX_LAYERS];
    /* arch1 */
    size_t w_e, b_e;                     /* embedding: W_e [E x V], b_e [E] */
    size_t w_q, b_q, w_v, b_v, w_o, b_o; /* multimodal */
    /* arch2 */
    size_t w_p, b_p; /* cnn projection W_p [E x I], b_p [E] */
    size_t w_lk;     /* lookup table [(V+1) x E] */
} nvqa_layout;

/* fusion: 0 netdef.AxB / 1 netdef.AskipB: the classifier reads the C-wide product; 2 netdef.A_B (misc/netdef.lua:27-35):
 * it reads JoinTable(qc, ic), 2C wide, so W_o is [A x 2C] (arch1 only) */
static inline int nvqa_layout_init_fusion(const nvqa_dims *d, int fusion, nvqa_layout *lo)
{
    size_t off = 0;
    const size_t R = (size_t)d->R, E = (size_t)d->E, V = (size_t)d->V, I = (size_t)d->I;
    const size_t C = (size_t)d->C, A = (size_t)d->A, L = (size_t)d->L;
    if (d->L < 1 || d->L > NVQA_MAX_LAYERS) return -1;
    if (d->arch == NVQA_ARCH1) {
        /* segment 0: encoder */
        for (size_t l = 0; l < L; ++l) {
            const size_t in = l == 0 ? E : R;
            lo->in_dim[l] = (int32_t)in;
            lo->w_i2h[l] = off; off += 4 * R * in;
            lo->b_i2h[l] = off; off += 4 * R;
            lo->w_h2h[l] = off; off += 4 * R * R;
            lo->b_h2h[l] = off; off += 4 * R;
        }
        lo->seg[0] = off;
        /* segment 1: embedding */
        lo->w_e = off; off += E * V;
        lo->b_e = off; off += E;
        lo->seg[1] = off - lo->seg[0];
        /* segment 2: multimodal */
        lo->w_q = off; off += C * 2 * R * L;
        lo->b_q = off; off += C;
        lo->w_v = off; off += C * I;
        lo->b_v = off; off += C;
        lo->w_o = off; off += A * (fusion == 2 ? 2 * C : C);
        lo->b_o = off; off += A;
        lo->seg[2] = off - lo->seg[0] - lo->seg[1];
        lo->w_p = lo->b_p = lo->w_lk = 0;
    } else if (d->arch == NVQA_ARCH2) {
        /* segment 0: cnn projection */
        lo->w_p = off; off += E * I;
        lo->b_p = off; off += E;
        lo->seg[0] = off;
        /* segment 1: encoder = LSTM layers then lookup (Encoder_lstm.lua:66-83) */
        for (size_t l = 0; l < L; ++l) {
            const size_t in = l == 0 ? E : R;
            lo->in_dim[l] = (int32_t)in;
            lo->w_i2h[l] = off; off += 4 * R * in;
            lo->b_i2h[l] = off; off += 4 * R;
            lo->w_h2h[l] = off; off += 4 * R * R;
            lo->b_h2h[l] = off; off += 4 * R;
        }
        lo->w_lk = off; off += (V + 1) * E;
        lo->seg[1] = off - lo->seg[0];
        /* segment 2: classifier */
        lo->w_o = off; off += A * R;
        lo->b_o = off; off += A;
        lo->seg[2] = off - lo->seg[0] - lo->seg[1];
        lo->w_e = lo->b_e = lo->w_q = lo->b_q = lo->w_v = lo->b_v = 0;
    } else {
        return -1;
    }
    lo->total = off;
    return 0;
}
static inline int nvqa_layout_init(const nvqa_dims *d, nvqa_layout *lo) { return nvqa_layout_init_fusion(d, 0, lo); }

#endif /* NVQA_LAYOUT_H */
