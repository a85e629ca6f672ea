/*
 * nvqa.h -- C ABI of libnvqa, the MI355X (gfx950) implementation of the VQA
 * training step of srama2512/novel-vqa.
 *
 * Every entry point below replaces one piece of the reference's Lua/Torch7
 * interface for the hot path (paths relative to the reference checkout):
 *
 *   nvqa_create / nvqa_destroy     net construction + :cuda()
 *                                  002_train_vqa_arch1/002_train_baseline.lua:141-171
 *                                  003_train_vqa_arch2/002_train_baseline.lua:138-177
 *   nvqa_param_count / _segments   sizes={...}; join_vector({...})
 *                                  002_train_baseline.lua:183,190 (arch2 :189,198)
 *   nvqa_init_params               *_w:uniform(-0.08,0.08)        002_train_baseline.lua:174-181
 *   nvqa_set_params/_get_params    split_vector + :copy            002_train_baseline.lua:273-286,
 *                                  004_eval_model.lua:154-163
 *   nvqa_step                      the optim closure JdJ(x)        002_train_baseline.lua:272-335
 *                                  (arch2 :277-333); batch = dataset:next_batch() :195-222
 *   nvqa_get_grads                 `gradients` returned by JdJ incl. clamp :328-329
 *   nvqa_rmsprop_update            optim.rmsprop(JdJ, x, config, state) :408,
 *                                  formula misc/rmsprop_lrscale.lua:13-34
 *   nvqa_forward                   eval-mode forward + argmax      004_eval_model.lua:202-233
 *   nvqa_dataset_load /            dataset[...] tensors + dataset:next_batch() gather
 *   nvqa_step_indices              002_train_baseline.lua:93-121,195-222
 *   nvqa_comm_*                    (absent in the reference: data-parallel gradient all-reduce)
 *
 * Conventions: plain C, POD arguments only.  All pointers are caller-owned
 * host memory borrowed for the duration of the call; device buffers, streams,
 * optimiser state and the RCCL communicator are owned by the context.  No
 * entry point throws or longjmps: 0 = success, negative = error, message via
 * nvqa_last_error() (thread-local).  Indices follow the reference's data
 * files: tokens 1..V (0 = padding), labels 1..A, image positions 1..N_img.
 */
#ifndef NVQA_H
#define NVQA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVQA_ARCH1 1 /* 002_train_vqa_arch1: Linear(V,E) embedding, n-layer LSTM, AxB fusion */
#define NVQA_ARCH2 2 /* 003_train_vqa_arch2: image-as-first-token nn.Encoder, Linear(R,A) head */

/* Model dimensions (SURVEY.md symbol table). */
typedef struct nvqa_dims {
    int32_t arch; /* NVQA_ARCH1 | NVQA_ARCH2 */
    int32_t B;    /* minibatch (QA pairs) per device             -batch_size          */
    int32_t T;    /* question buffer length (columns of ques_*)  buffer_size_q        */
    int32_t V;    /* question vocabulary size                    vocabulary_size_q    */
    int32_t E;    /* word embedding width                        -input_encoding_size */
    int32_t R;    /* LSTM hidden size                            -rnn_size            */
    int32_t L;    /* LSTM layers                                 -rnn_layer/-num_layers */
    int32_t I;    /* image feature width                         -nhimage             */
    int32_t C;    /* common embedding width (arch1 only)         -common_embedding_size */
    int32_t A;    /* answer classes                              -num_output          */
} nvqa_dims;

/* Dropout control.  Torch7's MT19937 bernoulli stream cannot be reproduced, so
 * masks come from the counter-based generator of nvqa_rng.h, identical on the
 * CPU oracle and on the device.  mode 0 = every D_* = 1 (gradient/logit parity
 * runs); mode 1 = Bernoulli(1-p) masks scaled by 1/(1-p), keyed by
 * (seed, step, site, element). */
typedef struct nvqa_dropout {
    int32_t mode;  /* 0 off, 1 seeded */
    float p;       /* drop probability, reference value 0.5 */
    uint64_t seed; /* reference default -seed 123 */
    uint64_t step; /* iteration counter, mixed into every mask */
} nvqa_dropout;

typedef struct nvqa_ctx nvqa_ctx;

/* ---- lifetime ------------------------------------------------------- */
int nvqa_create(const nvqa_dims *dims, int device, nvqa_ctx **out);
int nvqa_destroy(nvqa_ctx *ctx);
const char *nvqa_last_error(void);
int nvqa_sync(nvqa_ctx *ctx); /* wait for all enqueued work (loss_out becomes valid) */

/* ---- parameters: one flat fp32 vector, reference segment order ------- */
/* arch1: [encoder | embedding | multimodal]; arch2: [cnn | encoder | multimodal].
 * Intra-segment order (documented in DESIGN.md):
 *  arch1 encoder   : per layer W_i2h[4R x in], b_i2h[4R], W_h2h[4R x R], b_h2h[4R]
 *  arch1 embedding : W_e[E x V] (Torch nn.Linear layout), b_e[E]
 *  arch1 multimodal: W_q[C x 2RL], b_q[C], W_v[C x I], b_v[C], W_o[A x C], b_o[A]
 *  arch2 cnn       : W_p[E x I], b_p[E]
 *  arch2 encoder   : LSTM layers as above, then lookup W_lk[(V+1) x E]
 *  arch2 multimodal: W_o[A x R], b_o[A]                                         */
size_t nvqa_param_count(const nvqa_ctx *ctx);
int nvqa_segments(const nvqa_ctx *ctx, size_t sizes_out[3]);
int nvqa_init_params(nvqa_ctx *ctx, uint64_t seed, float lo, float hi);
int nvqa_set_params(nvqa_ctx *ctx, const float *params);
int nvqa_get_params(nvqa_ctx *ctx, float *params_out);
/* Gradient of the last step, reference layout; clamp > 0 applies
 * clamp(-clamp, clamp) on the way out (JdJ returns clamped gradients). */
int nvqa_get_grads(nvqa_ctx *ctx, float *grads_out, float clamp);

/* ---- the hot path ---------------------------------------------------- */
/* One forward+backward over a host-resident minibatch (the body of JdJ).
 *  arch1: tokens [B x T] right-aligned, 0 = left padding (right_align,
 *         misc/RNNUtils.lua:54-61); lengths [B] >= 1.
 *  arch2: tokens [B x T] left-aligned, 0 = null token (the library forms the
 *         reference's [T x B] transpose itself); lengths may be NULL.
 *  img [B x I] (already L2-normalised, 002_train_baseline.lua:117-121);
 *  labels [B], 1-based.  The mean cross-entropy is written to *loss_out once
 *  the work has drained (nvqa_sync or the next call). Gradients stay on the
 *  device for nvqa_rmsprop_update / nvqa_get_grads. */
int nvqa_step(nvqa_ctx *ctx, const int32_t *tokens, const int32_t *lengths, const float *img,
              const int32_t *labels, const nvqa_dropout *dropout, float *loss_out);

/* Loss of the last step (waits for the stream). With loss_out == NULL nvqa_step only
 * enqueues work; with a non-NULL loss_out it blocks like the reference's JdJ does. */
int nvqa_get_loss(nvqa_ctx *ctx, float *loss_out);

/* Evaluate-mode forward (all dropout = identity): scores [n x A] and/or
 * 1-based argmax [n] (either may be NULL), n <= B rows. */
int nvqa_forward(nvqa_ctx *ctx, int32_t n, const int32_t *tokens, const int32_t *lengths,
                 const float *img, float *scores_out, int32_t *argmax_out);

/* Evaluate-mode forward with everything the reference's validation and test loops take from it, on the device:
 *  labels [n] (optional, 1-based): *loss_out = mean cross-entropy over the n rows -- validate(),
 *         002_train_baseline.lua:337-381 (arch2: 003_.../002_train_baseline.lua:335-378);
 *  mc_ans [n x n_mc] (optional; 1-based answer ids, 0 = empty slot, n_mc <= 64): mc_argmax_out[i] = the candidate
 *         with the highest score, first one on ties, as torch.max over the candidates in slot order gives it --
 *         multiple-choice answers of 004_eval_model.lua:259-271 (MC_ans_test has 18 slots); rows without any
 *         candidate return 0;
 *  scores_out [n x A], argmax_out [n]: as nvqa_forward.  Every output pointer may be NULL. */
int nvqa_evaluate(nvqa_ctx *ctx, int32_t n, const int32_t *tokens, const int32_t *lengths, const float *img,
                  const int32_t *labels, const int32_t *mc_ans, int32_t n_mc, float *scores_out,
                  int32_t *argmax_out, int32_t *mc_argmax_out, float *loss_out);

/* clamp -> (+ wd * x) -> m = alpha m + (1-alpha) g^2 -> x -= lr g / (sqrt(m) + eps).
 * With a communicator (nvqa_comm_init) nvqa_step itself sums the gradient over the ranks,
 * one all-reduce per parameter segment overlapped with the backward pass; the update then
 * divides by the world size and the clamp acts on that average.  nvqa_get_grads returns the
 * same average. */
int nvqa_rmsprop_update(nvqa_ctx *ctx, float lr, float alpha, float eps, float wd, float clamp);

/* ---- model variants of the reference's other training scripts -------------- */
/* The fusion module in front of the classifier (002_train_vqa_arch1/misc/netdef.lua), arch1 only:
 *   mode 0  netdef.AxB    (:6-14)   qc (*) ic                      -- what every training script of the reference uses
 *   mode 1  netdef.AskipB (:16-25)  qc + qc (*) ic                 -- 003_train_ae_based_wp.lua:151
 *   mode 2  netdef.A_B    (:27-35)  JoinTable(2)({qc, ic}): the classifier reads 2C values, so W_o becomes [A x 2C]:
 *           nvqa_param_count / nvqa_segments / include/nvqa_layout.h (nvqa_layout_init_fusion) change.  Switching to or
 *           from mode 2 is only accepted on a context that has not been given parameters yet (right after nvqa_create). */
int nvqa_set_fusion(nvqa_ctx *ctx, int mode);
/* BASELINE config "arch2 ... bf16": bf16 = 1 runs every dense product of the step (forward, dgrad, wgrad)
 * on the bf16 matrix cores: both operands rounded to bf16 (round-to-nearest-even) as they are read,
 * products accumulated in f32.  Parameters, activations, gradients, the optimiser and everything at this
 * ABI stay f32 (SURVEY.md 8b "numerics contract").  Default 0 = the reference's f32 arithmetic
 * (torch.setdefaulttensortype('torch.FloatTensor'), 002_train_baseline.lua:54). */
int nvqa_set_precision(nvqa_ctx *ctx, int bf16);
/* arch2 only: reproduce, on request, two things the reference's nn.Encoder does that are artefacts of its Lua code
 * (flags = OR of the bits; default 0 = the model as designed):
 *  NVQA_QUIRK_H0      misc/Encoder_lstm.lua:238-239 with :30-47,:164 -- backward aliases the table of initial-state
 *                     tensors and stores gradOutput in its last slot, so from the second iteration on (training and
 *                     validation forwards alike) the top layer starts from h0 = dL/d(encoder output) of the previous
 *                     backward, and the step-1 weight gradient of the top layer's W_h2h is taken against the CURRENT
 *                     step's dL/d(encoder output).
 *  NVQA_QUIRK_LOOKUP  misc/Encoder_lstm.lua:49-58 with 002_train_baseline.lua:186,273 -- getParameters() runs before
 *                     createClones(), whose lookup-table clones share the weight but not gradWeight with the flattened
 *                     module: the lookup slice of encoder_dw_q stays zero, W_lk only sees weightDecay.
 * Setting the flags also clears the carried h0 state. */
#define NVQA_QUIRK_H0 1
#define NVQA_QUIRK_LOOKUP 2
int nvqa_set_ref_quirks(nvqa_ctx *ctx, int flags);
/* torch.norm(cnn_w), torch.norm(encoder_w_q), torch.norm(multimodal_w) of 003_train_vqa_arch2/002_train_baseline.lua:402-404
 * (arch1: encoder_w_q, embedding_w_q, multimodal_w): the L2 norm of each of the three parameter segments in
 * nvqa_segments order, reduced on the device.  Synchronises the context's stream. */
int nvqa_param_norms(nvqa_ctx *ctx, float out[3]);
/* out[0] / out[1] = 1 when the next training step runs the LSTM forward unroll / the BPTT as ONE persistent launch
 * (R = 512, E in {200, 512}; BPTT: L <= 2), 0 when it takes the per-level kernels (other shapes, NVQA_PERSIST=0, or after a
 * reported time-out). */
int nvqa_persistent_state(const nvqa_ctx *ctx, int out[2]);
/* Per-segment gradient scale applied before the clamp: {lr_scale, lr_scale, 1} reproduces
 * -lr_scale of 003_train_ae_based_wp.lua:30,344. */
int nvqa_set_grad_scales(nvqa_ctx *ctx, const float scales[3]);

/* ---- HBM-resident dataset (the tensors of 002_train_baseline.lua:93-121) ---- */
/* questions [N x T] (aligned as nvqa_step expects), lengths [N], img_pos [N]
 * (1-based rows of feats), answers [N] (1-based), feats [N_img x I].
 * l2_normalize: 0 = keep, 1 = row L2 norm (002_train_baseline.lua:117-121), n > 1 = the two blocks
 * [0,n) and [n,I) normalised separately (early fusion, 003_train_ae_based_ef.lua:115-119, n = 2048). */
int nvqa_dataset_load(nvqa_ctx *ctx, int64_t n_q, const int32_t *questions, const int32_t *lengths,
                      const int32_t *img_pos, const int32_t *answers, int64_t n_img,
                      const float *feats, int l2_normalize);
/* JdJ with dataset:next_batch() done on the device: qinds [B], 0-based rows. */
int nvqa_step_indices(nvqa_ctx *ctx, const int64_t *qinds, const nvqa_dropout *dropout,
                      float *loss_out);

/* ---- data parallel (new functionality; the reference is single-GPU) --- */
#define NVQA_COMM_ID_BYTES 128
int nvqa_comm_unique_id(void *id_out /* NVQA_COMM_ID_BYTES */);
int nvqa_comm_init(nvqa_ctx *ctx, int rank, int world, const void *id);
/* Path of the collective library the nvqa_comm_* entries run on (loads it if that has not happened yet; NULL +
 * nvqa_last_error() when none can be loaded).  The rule: NVQA_RCCL_LIB if set; else the librccl that sits beside the HIP
 * runtime image mapped into this process, opened RTLD_LOCAL. */
const char *nvqa_comm_library(void);

/* ---- VGG-16 fc7 feature extractor (002_train_vqa_arch1/001_prepro_img_vgg.lua) ---- */
/* Forward only.  width_div = 1 and input_hw = 224 is the real 16-layer VGG (channels
 * 64..512, fc 4096); other values shrink it for tests.  Replaces loadcaffe.load +
 * net:forward + net.modules[38].output (:36-37,109-110). */
typedef struct nvqa_vgg nvqa_vgg;
int nvqa_vgg16_create(int device, int width_div, int input_hw, int max_batch, nvqa_vgg **out);
int nvqa_vgg16_destroy(nvqa_vgg *vgg);
size_t nvqa_vgg16_weight_count(const nvqa_vgg *vgg);
int nvqa_vgg16_feature_dim(const nvqa_vgg *vgg);
/* One flat vector in Caffe order and layout: per conv W [Cout][Cin][3][3], b [Cout];
 * fc6 W [F][C5*S*S] (CHW-flattened input), b [F]; fc7 W [F][F], b [F]. */
int nvqa_vgg16_set_weights(nvqa_vgg *vgg, const float *flat);
/* images: n x 3 x hw x hw as loadim returns them (BGR planes, mean-subtracted, :65-70);
 * feats_out: n x F = post-ReLU fc7. */
int nvqa_vgg16_fc7(nvqa_vgg *vgg, const float *images, int n, float *feats_out);
/* 1: the operands of every convolution / fc product are rounded to bf16 (bf16 matrix cores, f32 accumulate; weights,
 * activations and the features at the ABI stay f32), as nvqa_set_precision does for the training step.  The reference
 * extracts in fp32 (001_prepro_img_vgg.lua:36,109-110): 0 is the default. */
int nvqa_vgg16_set_precision(nvqa_vgg *vgg, int bf16);
/* loadim's arithmetic (:50,65-69): bilinear scale of RGB [0,1] planes n x 3 x H x W to hw x hw,
 * x255, RGB->BGR, mean subtraction.  out: n x 3 x hw x hw. */
int nvqa_vgg16_preprocess(nvqa_vgg *vgg, const float *rgb, int n, int H, int W, float *out);

/* The extractor run on the fly in front of the training step (BASELINE config "end-to-end"):
 * images [B x 3 x hw x hw] as loadim returns them -> fc7 -> row L2 norm (002_train_baseline.lua:117-121)
 * -> the JdJ body of nvqa_step.  The extractor's feature width must equal dims.I; the features
 * never leave the device. */
int nvqa_step_images(nvqa_ctx *ctx, nvqa_vgg *vgg, const float *images, const int32_t *tokens,
                     const int32_t *lengths, const int32_t *labels, const nvqa_dropout *dropout,
                     float *loss_out);

/* ---- measurement ------------------------------------------------------ */
/* HIP-event timing of kernel groups on the stream they are launched on.
 * enable=1 brackets every launch with events (slows the step; bench.py uses a
 * separate timed pass for it). Names: see nvqa_profile_name(). */
int nvqa_profile_enable(nvqa_ctx *ctx, int enable);
int nvqa_profile_reset(nvqa_ctx *ctx);
int nvqa_profile_count(const nvqa_ctx *ctx);
const char *nvqa_profile_name(const nvqa_ctx *ctx, int idx);
/* total_ms and launches accumulated since reset; flops = algorithmic FLOPs issued */
int nvqa_profile_get(nvqa_ctx *ctx, int idx, double *total_ms, int64_t *launches, double *flops,
                     double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* NVQA_H */
