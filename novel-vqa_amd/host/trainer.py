"""Host-side mirror of the reference's training-step interface.

The reference drives training through the optim closure protocol
(002_train_vqa_arch1/002_train_baseline.lua:272-335,394-414):

    for iter = 1, max_iters do
        optim.rmsprop(JdJ, winit, optimize, state)     -- JdJ(x) -> f, gradients
        optimize.learningRate = optimize.learningRate * decay_factor
    end

`VQATrainer` keeps the same names and argument meaning: `JdJ()` returns
(loss, clamped flat gradient) for one minibatch, `rmsprop()` is the fused
device-side update, `next_batch()` mirrors dataset:next_batch() (:195-222),
`validate()`/`predict()` mirror the eval-mode forward (:337-381,
004_eval_model.lua:202-233).  Same file serves arch1 and arch2.
"""
import numpy as np

from . import binding, t7

DECAY_FACTOR = 0.99997592083  # 002_train_baseline.lua:78


class VQATrainer:
    def __init__(self, dims, device=0, learning_rate=3e-4, alpha=0.99, epsilon=1e-8,
                 weight_decay=None, clamp=10.0, seed=123, dropout_p=0.5, dropout=True, rank=0, ref_quirks=0):
        self.dims = dims
        self.ctx = binding.Context(dims, device)
        self.learningRate = learning_rate
        self.alpha, self.epsilon, self.clamp = alpha, epsilon, clamp
        # arch2 baseline sets optimize.weightDecay = 1e-4 (003_.../002_train_baseline.lua:197)
        self.weightDecay = (1e-4 if dims.arch == 2 else 0.0) if weight_decay is None else weight_decay
        self.seed = seed
        # data parallel: parameters are initialised from `seed` on every rank (identical replicas), but each rank
        # draws its own dropout masks -- ranks hold different samples, identical masks would correlate them
        self.dropout_seed = seed + 1000 * rank
        self.rank = rank
        self.iter = 0
        self.dropout_p = dropout_p
        self.dropout_on = dropout
        self.running_avg = None
        self.rng = np.random.default_rng(seed + 1000 * rank)  # each rank draws its own sample ids
        self.n_questions = 0
        # arch2: 3 reproduces what 003_train_vqa_arch2/002_train_baseline.lua really trains (aliased top-layer h0 + lookup
        # table without gradient, include/nvqa.h nvqa_set_ref_quirks) -- lua/train_arch2.lua's default; 0 = the model as designed
        if dims.arch == 2 and ref_quirks:
            self.ctx.set_ref_quirks(ref_quirks)

    # -- parameters (join_vector({encoder_w_q, embedding_w_q, multimodal_w})) ---------------
    def init_params(self, lo=-0.08, hi=0.08):
        self.ctx.init_params(self.seed, lo, hi)

    def set_params(self, x):
        self.ctx.set_params(x)

    def get_params(self):
        return self.ctx.get_params()

    def _dropout(self):
        return binding.Dropout(1 if self.dropout_on else 0, self.dropout_p, self.dropout_seed, self.iter)

    # -- dataset:next_batch() -------------------------------------------------------------------
    def load_dataset(self, questions, lengths, img_pos, answers, feats, img_norm=True):
        """dataset[...] tensors (:93-121); right_align must already be applied for arch1.
        img_norm: False/True, or an int split n > 1 for the early-fusion two-block norm."""
        self.ctx.dataset_load(questions, lengths, img_pos, answers, feats, l2_normalize=int(img_norm))
        self.n_questions = int(np.asarray(questions).shape[0])

    def next_batch(self):
        """qinds[i] = torch.random(nqs): sampling with replacement on the host (:202-205)."""
        return self.rng.integers(0, self.n_questions, self.dims.B, dtype=np.int64)

    # -- JdJ -----------------------------------------------------------------------------------------
    def JdJ(self, batch=None, want_grads=True):
        """One forward+backward. batch = (tokens, lengths, img, labels) or None to draw
        from the loaded dataset.  Returns (f, gradients) like the reference closure."""
        if batch is None:
            f = self.ctx.step_indices(self.next_batch(), self._dropout())
        else:
            tokens, lengths, img, labels = batch
            f = self.ctx.step(tokens, lengths, img, labels, self._dropout())
        self.running_avg = f if self.running_avg is None else self.running_avg * 0.95 + f * 0.05
        return f, (self.ctx.get_grads(self.clamp) if want_grads else None)

    def rmsprop(self):
        """optim.rmsprop's update for the gradient left on the device by JdJ, then the
        per-iteration LR decay (:408-410)."""
        self.ctx.rmsprop_update(self.learningRate, self.alpha, self.epsilon, self.weightDecay, self.clamp)
        self.learningRate *= DECAY_FACTOR
        self.iter += 1

    def log_line(self):
        """The line the training scripts write to save/logFile.txt every 100 iterations: arch1
        (002_train_baseline.lua:413-416) the running loss, arch2 (003_train_vqa_arch2/002_train_baseline.lua:400-407) the
        running loss and torch.norm of the three parameter vectors (nvqa_param_norms: reduced on the device)."""
        if self.dims.arch == 2:
            n = self.ctx.param_norms()
            return "iter: %6d train loss: %.3f cnn_norm: %.3f enc_norm: %.3f mm_norm: %.3f" % (self.iter, self.running_avg, n[0], n[1], n[2])
        return "training loss: %s" % repr(float(self.running_avg))  # (the scripts append 'on iter: i/max_iters')

    def train_iteration(self, batch=None):
        f, _ = self.JdJ(batch, want_grads=False)
        self.rmsprop()
        return f

    # -- evaluation ------------------------------------------------------------------------------------
    def predict(self, tokens, lengths, img):
        """Eval-mode forward: (scores [n, A], argmax [n] 1-based), 004_eval_model.lua:202-233."""
        tokens = np.asarray(tokens)
        B = self.dims.B
        scores, preds = [], []
        for s in range(0, tokens.shape[0], B):
            sc, am = self.ctx.forward(tokens[s:s + B], None if lengths is None else lengths[s:s + B],
                                      img[s:s + B])
            scores.append(sc)
            preds.append(am)
        return np.concatenate(scores), np.concatenate(preds)

    def validate(self, tokens, lengths, img, labels):
        """validate() of the training scripts (002_train_baseline.lua:337-381; arch2 :335-378): evaluate-mode forward
        over the validation split in batches of B (the last one short), f_avg = mean of the per-batch mean
        cross-entropies, and the same 0.95 / 0.05 running average the scripts log."""
        tokens = np.asarray(tokens)
        B, n = self.dims.B, np.asarray(tokens).shape[0]
        f_sum, iters = 0.0, 0
        for s in range(0, n, B):
            r = self.ctx.evaluate(tokens[s:s + B], None if lengths is None else lengths[s:s + B], img[s:s + B],
                                  labels=labels[s:s + B])
            f = r["loss"]
            self.running_avg_val = f if getattr(self, "running_avg_val", None) is None else self.running_avg_val * 0.95 + f * 0.05
            f_sum += f
            iters += 1
        return f_sum / max(iters, 1)

    def predict_mc(self, tokens, lengths, img, mc_ans):
        """Open-ended argmax and multiple-choice answer ids (004_eval_model.lua:233,259-271), both taken on the device."""
        tokens = np.asarray(tokens)
        B = self.dims.B
        pred, mc = [], []
        for s in range(0, tokens.shape[0], B):
            r = self.ctx.evaluate(tokens[s:s + B], None if lengths is None else lengths[s:s + B], img[s:s + B],
                                  mc_ans=np.asarray(mc_ans)[s:s + B])
            pred.append(r["argmax"])
            mc.append(r["mc_argmax"])
        return np.concatenate(pred), np.concatenate(mc)

    # -- torch.save / torch.load of the reference's checkpoint table (:401-402, 004_eval_model.lua:154-163)
    # PARITY UNPINNED: the order of the LSTM tensors INSIDE encoder_w_q is this package's (include/nvqa_layout.h), not
    # nngraph's forward-node order, which cannot be established offline (no nngraph source, no reference .t7).  Files
    # written here carry layout = 'nvqa'; a file without the marker (a genuine Torch7 checkpoint) is refused unless
    # the caller supplies the permutation (host/t7.py: encoder_permutation).
    def save_checkpoint(self, path):
        t7.save_checkpoint(path, self.dims.arch, self.get_params(), self.ctx.segments())

    def load_checkpoint(self, path, encoder_perm=None):
        self.set_params(t7.load_checkpoint(path, self.dims.arch, self.ctx.segments(), encoder_perm=encoder_perm))

    # -- initialisation from an auto-encoder checkpoint (the *_ae_based* scripts) -------------------------------------------
    def init_from_autoencoder(self, path, encoder_perm=None, with_multimodal=False):
        """What the AE-based trainers do in place of `*_w:uniform(-0.08, 0.08)`.

        arch1 (002_train_vqa_arch1/003_train_ae_based.lua:65,175-186; _inc / _ef alike): `path` is the table written by
        001_train_autoencoder/002_convert_text_model_arch1.lua:27-39 -- lookup = the AE's LookupTable weight transposed,
        [E x (V+1)], encoder = the AE encoder's flat parameters.  embedding weight <- lookup[:, 1 .. V] (the last column,
        the START token of the AE, is dropped), embedding bias <- 0, encoder_w_q <- encoder, multimodal_w <-
        uniform(-0.08, 0.08).  with_multimodal (003_train_ae_based_wp.lua:153-160): the fusion projections W_q, b_q, W_v,
        b_v are copied from the table's 'multimodal' tensor as well and only the classifier Linear(C, A) is uniform.

        arch2 (003_train_vqa_arch2/003_train_ae_based.lua:74-75,150-152,186-194) clones modelT.ae.encoder and
        modelT.ae.lookup_table out of the AE checkpoint's nn modules; this reader takes tensors only (nothing in a file
        is executed), so `path` is the table lua/convert_ae_arch2.lua writes from that checkpoint: encoder = the LSTM's
        flat parameters, lookup_table = [(V+1) x E].  cnn_w and multimodal_w <- uniform(-0.08, 0.08) (:189,194).

        The uniform segments come from the library's counter-based stream (nvqa_init_params with this trainer's seed), so
        every rank of a data-parallel job builds the same vector.  As with load_checkpoint, the order of the LSTM tensors
        inside `encoder` is nngraph's: a table without layout = 'nvqa' needs encoder_perm (t7.encoder_permutation).
        PARITY UNPINNED: the reference ships no AE checkpoint."""
        d = self.dims
        t = t7.load(path)
        if t.get("layout") != t7.LAYOUT_MARK and encoder_perm is None:
            raise ValueError("auto-encoder table has no layout = 'nvqa' marker: the order of the LSTM tensors inside 'encoder' is "
                             "unknown (nngraph forward-node order); pass encoder_perm = t7.encoder_permutation(...)")
        self.ctx.init_params(self.seed, -0.08, 0.08)
        x = self.ctx.get_params()
        seg = self.ctx.segments()
        enc = np.asarray(t["encoder"], np.float32).ravel().copy()
        if encoder_perm is not None:
            enc[:len(encoder_perm)] = enc[np.asarray(encoder_perm)]
        if d.arch == 1:
            lookup = np.asarray(t["lookup"], np.float32)
            if lookup.ndim != 2 or lookup.shape != (d.E, d.V + 1):
                raise ValueError(f"lookup is {lookup.shape}, the model wants ({d.E}, {d.V + 1}) = [E x (V + 1)]")
            if enc.size != seg[0]:
                raise ValueError(f"encoder has {enc.size} values, the model wants {seg[0]}")
            x[:seg[0]] = enc
            x[seg[0]:seg[0] + d.E * d.V] = lookup[:, :d.V].ravel()          # Linear(V, E).weight [E x V]
            x[seg[0] + d.E * d.V:seg[0] + seg[1]] = 0.0                      # Linear bias <- 0 (:178)
            if with_multimodal:
                # 003_train_ae_based_wp.lua:151 builds netdef.AskipB; copying an AE's fusion projections into an AxB model
                # would train something the reference never does.  (The order of W_q, b_q / W_v, b_v INSIDE the
                # 'multimodal' tensor is nngraph's as well: PARITY UNPINNED, taken here as W_q, b_q, W_v, b_v.)
                if self.ctx.fusion != 1:
                    raise ValueError("with_multimodal=True is the -variant wp path: call ctx.set_fusion(1) (netdef.AskipB) "
                                     "right after creating the trainer, before any parameters are set")
                mm = np.asarray(t["multimodal"], np.float32).ravel()
                n_fuse = d.C * (2 * d.R * d.L) + d.C + d.C * d.I + d.C       # W_q, b_q, W_v, b_v
                if mm.size != n_fuse:
                    raise ValueError(f"multimodal has {mm.size} values, the fusion projections want {n_fuse}")
                x[seg[0] + seg[1]:seg[0] + seg[1] + n_fuse] = mm
        else:
            lk = np.asarray(t["lookup_table"], np.float32)
            if lk.shape != (d.V + 1, d.E):
                raise ValueError(f"lookup_table is {lk.shape}, the model wants ({d.V + 1}, {d.E})")
            n_lstm = seg[1] - lk.size
            if enc.size != n_lstm:
                raise ValueError(f"encoder has {enc.size} values, the model's LSTM wants {n_lstm}")
            x[seg[0]:seg[0] + n_lstm] = enc
            x[seg[0] + n_lstm:seg[0] + seg[1]] = lk.ravel()
        self.ctx.set_params(x)

    def close(self):
        self.ctx.close()


def late_fusion(scores_a, scores_b, weight_a=1.0, weight_b=1.0):
    """004_eval_model_lf.lua:109-132: weighted sum of two models' answer scores."""
    return weight_a * np.asarray(scores_a, np.float64) + weight_b * np.asarray(scores_b, np.float64)


def results_json(question_ids, answer_ids, ix_to_ans):
    """The reference's result file: [{question_id, answer}] with answers looked up by their 1-based
    id in json_file['ix_to_ans'] (004_eval_model.lua:249-255); feed to json.dump."""
    return [{"question_id": int(q), "answer": ix_to_ans[str(int(a))]} for q, a in zip(question_ids, answer_ids)]


def multiple_choice_argmax(scores, mc_ans):
    """Masked argmax over the non-zero candidate ids (004_eval_model.lua:259-271).
    scores [n, A]; mc_ans [n, 18] 1-based answer ids, 0 = empty slot. Returns 1-based ids."""
    scores = np.asarray(scores, np.float64)
    mc_ans = np.asarray(mc_ans)
    out = np.zeros(scores.shape[0], np.int64)
    for i in range(scores.shape[0]):
        cand = mc_ans[i][mc_ans[i] != 0]
        out[i] = cand[np.argmax(scores[i, cand - 1])]  # first maximal candidate, like torch.max
    return out
