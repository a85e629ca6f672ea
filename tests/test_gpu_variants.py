"""Model variants of the reference's other training scripts on the HIP path:
AskipB fusion, -lr_scale, early-fusion two-block feature normalisation."""
import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu
KW = dict(arch=1, B=9, T=7, V=50, E=12, R=16, L=2, I=32, C=24, A=12)


def test_askipb_fusion(pkg, orc):
    d = orc.make_dims(**KW)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    dr = orc.Dropout(1, 0.5, 123, 2)
    o = orc.Oracle(np.float64)
    o.set_fusion(1)
    try:
        ref = o.step(d, params, tok, lens, img, lab, dr)
    finally:
        o.set_fusion(0)
    base = o.step(d, params, tok, lens, img, lab, dr)
    assert abs(ref["loss"] - base["loss"]) > 1e-6  # the variant really changes the model
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    ctx.set_fusion(1)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
    assert abs(loss - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    bad = {k: e for k, e in segment_errors(orc, d, ctx.get_grads(), ref["grads"]).items() if e > 1e-3}
    assert not bad, bad
    ctx.close()


def test_a_b_join_fusion(pkg, orc):
    """netdef.A_B (misc/netdef.lua:27-35; defined, not called by any reference script): JoinTable(2)({qc, ic}) in front
    of a Linear(2C, A) classifier = nvqa_set_fusion(ctx, 2).  Another parameter layout (W_o [A x 2C]): accepted only on
    a fresh context; loss, logits and every gradient segment against the f64 oracle (pinned on the CPU against the
    independent autograd model, tests/test_oracle.py), through both head routes (split-K slabs + k_head_fuse for the
    full-size head, the fused epilogue for this small one)."""
    from util import assert_logits
    for kw in (KW, dict(arch=1, B=512, T=6, V=50, E=16, R=32, L=1, I=128, C=256, A=40)):
        d = orc.make_dims(**kw)
        lo = orc.layout(d, 2)
        params = orc.synth_params(d, fusion=2)
        tok, lens, img, lab = orc.synth_batch(d, full_length=False)
        dr = orc.Dropout(1, 0.5, 123, 2)
        o = orc.Oracle(np.float64)
        o.set_fusion(2)
        try:
            ref = o.step(d, params, tok, lens, img, lab, dr)
            ev = o.step(d, params, tok, lens, img, lab, None, train=False)
        finally:
            o.set_fusion(0)
        ctx = pkg.binding.Context(gdims(pkg, d), 0)
        n0 = ctx.param_count
        ctx.set_fusion(2)
        assert ctx.param_count == lo["_total"] == n0 + d.A * d.C
        ctx.set_params(params)
        with pytest.raises(pkg.binding.NvqaError):
            ctx.set_fusion(0)  # the layout is fixed once the context holds parameters
        loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
        assert abs(loss - ref["loss"]) <= 2e-6 * abs(ref["loss"])
        errs = segment_errors(orc, d, ctx.get_grads(), ref["grads"], fusion=2)
        assert max(errs.values()) < 2e-5, errs
        scores, argmax = ctx.forward(tok, lens, img)
        assert_logits(scores, ev["scores"])
        # the optimiser walks the longer vector too
        ctx.rmsprop_update(3e-4)
        p1 = ctx.get_params()
        assert p1.size == lo["_total"] and np.abs(p1[lo["w_o"][0]:lo["w_o"][0] + lo["w_o"][1]] - params[lo["w_o"][0]:lo["w_o"][0] + lo["w_o"][1]]).max() > 0
        ctx.close()


def test_lr_scale(pkg, orc):
    # gradients = join{enc*lr_scale, emb*lr_scale, mm}; clamp  (003_train_ae_based_wp.lua:344-345)
    d = orc.make_dims(**KW)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    o = orc.Oracle(np.float32)
    g = o.step(d, params, tok, lens, img, lab, None)["grads"]
    s0, s1, s2 = orc.layout(d)["_segments"]
    scale = np.concatenate([np.full(s0, 0.1, np.float32), np.full(s1, 0.1, np.float32), np.ones(s2, np.float32)])
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    ctx.set_grad_scales([0.1, 0.1, 1.0])
    ctx.step(tok, lens, img, lab, None)
    assert relmax(ctx.get_grads(), g * scale) < 1e-4
    ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, 10.0)
    x, m, gs = params.copy(), np.zeros_like(params), (g * scale).astype(np.float32)
    o.rmsprop(x, gs, m, 3e-4, 0.99, 1e-8, 0.0, 10.0)
    assert relmax(ctx.get_params(), x) < 1e-5
    ctx.close()


def test_two_block_feature_norm(pkg, orc):
    # early fusion: [0,n) and [n,I) normalised separately (003_train_ae_based_ef.lua:115-119)
    d = orc.make_dims(**{**KW, "I": 48})
    params = orc.synth_params(d)
    tok, lens, _, lab = orc.synth_batch(d, full_length=False)
    rng = np.random.default_rng(2)
    feats = np.abs(rng.standard_normal((20, d.I))).astype(np.float32)
    img_pos = rng.integers(1, 21, d.B).astype(np.int32)
    n = 16
    fn = feats.copy()
    fn[:, :n] /= np.sqrt((fn[:, :n] ** 2).sum(1, keepdims=True))
    fn[:, n:] /= np.sqrt((fn[:, n:] ** 2).sum(1, keepdims=True))
    tr = pkg.trainer.VQATrainer(gdims(pkg, d), 0, dropout=False)
    tr.set_params(params)
    tr.load_dataset(tok, lens, img_pos, lab, feats, img_norm=n)
    q = np.arange(d.B, dtype=np.int64)
    la = tr.ctx.step_indices(q, None)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, fn[img_pos - 1], lab, None)
    assert abs(la - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    tr.close()


def test_late_fusion_and_results_json(pkg):
    a = np.array([[0.1, 0.9], [0.6, 0.4]])
    b = np.array([[0.8, 0.1], [0.2, 0.3]])
    s = pkg.trainer.late_fusion(a, b, 1.0, 2.0)
    assert s.argmax(1).tolist() == [0, 0]
    r = pkg.trainer.results_json([11, 12], [2, 1], {"1": "yes", "2": "no"})
    assert r == [{"question_id": 11, "answer": "no"}, {"question_id": 12, "answer": "yes"}]


def test_evaluate_validation_loss_and_multiple_choice(pkg, orc):
    """nvqa_evaluate = what validate() (002_train_baseline.lua:337-381) and the test script (004_eval_model.lua:233,
    259-271) take from an evaluate-mode forward: mean cross-entropy of the rows, open-ended argmax, and the
    multiple-choice answer among the non-zero candidates (first on ties, as torch.max over the slot order)."""
    for kw in (dict(arch=1, B=24, T=7, V=40, E=16, R=16, L=2, I=32, C=24, A=32),
               dict(arch=2, B=24, T=6, V=40, E=16, R=16, L=1, I=32, C=4, A=32)):
        d = orc.make_dims(**kw)
        params = orc.synth_params(d) * np.float32(3.0)
        tok, lens, img, lab = orc.synth_batch(d, full_length=False)
        lens = lens if d.arch == 1 else None
        rng = np.random.default_rng(2)
        mc = rng.integers(0, d.A + 1, (d.B, 18)).astype(np.int32)   # 0 = empty slot
        mc[3] = 0                                                    # a row without candidates
        mc[5, :] = mc[5, 0] if mc[5, 0] else 7                       # all slots the same answer: a tie
        ctx = pkg.binding.Context(gdims(pkg, d), 0)
        ctx.set_params(params)
        for n in (d.B, 9):                                           # full batch and a short last batch
            sl = slice(0, n)
            dn = orc.make_dims(**{**kw, "B": n})
            ref = orc.Oracle(np.float64).step(dn, params, tok[sl], None if lens is None else lens[sl], img[sl], lab[sl], None, train=False)
            r = ctx.evaluate(tok[sl], None if lens is None else lens[sl], img[sl], labels=lab[sl], mc_ans=mc[sl])
            assert abs(r["loss"] - ref["loss"]) <= 2e-6 * abs(ref["loss"])
            assert np.array_equal(r["argmax"], ref["argmax"])
            exp = pkg.trainer.multiple_choice_argmax(ref["scores"][[i for i in range(n) if mc[i].any()]], mc[[i for i in range(n) if mc[i].any()]])
            got = r["mc_argmax"]
            assert np.array_equal(got[[i for i in range(n) if mc[i].any()]], exp)
            assert all(got[i] == 0 for i in range(n) if not mc[i].any())
        ctx.close()


@pytest.mark.parametrize("arch", [1, 2])
def test_embedding_gradient_skewed_vocabulary(pkg, orc, arch):
    """Real questions are Zipfian ("what", "is", "the" in most of them; arch2's START token in every row): the embedding /
    lookup gradient by token segments (kernels.h: k_tok_index + k_emb_bwd_tok) with words that occur once, a few dozen
    times (one wave), several hundred times and more than B times (chunked over waves, combined by the last to arrive)
    against the f64 oracle; bit-reproducible; and the same values as the scanning kernel (NVQA_EMB_SEG=0) to f32
    summation-order noise."""
    import os
    from util import assert_grads, gdims, gdrop
    kw = (dict(arch=1, B=512, T=26, V=14773, E=200, R=64, L=1, I=64, C=64, A=40) if arch == 1 else
          dict(arch=2, B=512, T=26, V=14773, E=512, R=64, L=1, I=64, C=4, A=40))
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=5, full_length=False, min_len=4)
    rng = np.random.default_rng(17)
    # Zipf over a 3000-word head + the uniform tail that synth_batch drew; word 7 twice per question where it fits
    zipf = np.minimum(rng.zipf(1.3, tok.shape), 3000).astype(np.int32)
    use = rng.random(tok.shape) < 0.7
    tok = np.where((tok > 0) & use, zipf, tok).astype(np.int32)
    live = tok > 0
    for b in range(d.B):
        idx = np.nonzero(live[b])[0]
        if len(idx) >= 6:
            tok[b, idx[1]] = 7
            tok[b, idx[4]] = 7
    counts = np.bincount(tok[tok > 0])
    assert counts.max() > d.B and (counts == 1).sum() > 100 and ((counts > 32) & (counts < 512)).sum() > 3
    lens_ = lens if arch == 1 else None
    dr = orc.Dropout(1, 0.5, 123, 3)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens_, img, lab, dr)
    got = {}
    for seg in ("1", "0"):
        old = os.environ.get("NVQA_EMB_SEG")
        os.environ["NVQA_EMB_SEG"] = seg
        try:
            ctx = pkg.binding.Context(gdims(pkg, d), 0)
        finally:
            if old is None:
                del os.environ["NVQA_EMB_SEG"]
            else:
                os.environ["NVQA_EMB_SEG"] = old
        ctx.set_params(params)
        loss = ctx.step(tok, lens_, img, lab, gdrop(pkg, dr))
        g = ctx.get_grads()
        assert abs(loss - ref["loss"]) <= 2e-6 * abs(ref["loss"])
        assert_grads(orc, d, g, ref["grads"], 2e-5, f"emb_skew_arch{arch}_seg{seg}")
        l2 = ctx.step(tok, lens_, img, lab, gdrop(pkg, dr))
        assert l2 == loss and np.array_equal(ctx.get_grads(), g)
        got[seg] = g
        ctx.close()
    lo = orc.layout(d)
    o, n = lo["w_e" if arch == 1 else "w_lk"]
    a, b = got["1"][o:o + n], got["0"][o:o + n]
    assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    assert np.count_nonzero(a) > 0


def test_init_from_autoencoder_through_the_library(pkg, orc, tmp_path):
    """VQATrainer.init_from_autoencoder (003_train_ae_based.lua:65,175-186) on a real context: the embedding weight crosses
    the ABI in Torch's nn.Linear layout [E x V] (the device keeps it transposed), the uniform segments come from
    nvqa_init_params, and a training step runs on the result."""
    import numpy as np
    d = orc.make_dims(arch=1, B=16, T=6, V=37, E=24, R=32, L=2, I=48, C=40, A=12)
    tr = pkg.trainer.VQATrainer(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), device=0, seed=5)
    seg = tr.ctx.segments()
    rng = np.random.default_rng(0)
    lookup = (0.1 * rng.standard_normal((d.E, d.V + 1))).astype(np.float32)
    enc = (0.05 * rng.standard_normal(seg[0])).astype(np.float32)
    path = str(tmp_path / "ae.t7")
    pkg.t7.save(path, {"lookup": lookup, "encoder": enc, "layout": "nvqa"})
    tr.init_params()
    uniform = tr.get_params()
    tr.init_from_autoencoder(path)
    x = tr.get_params()
    e, m = seg[0], seg[1]
    assert np.array_equal(x[:e], enc)
    assert np.array_equal(x[e:e + d.E * d.V].reshape(d.E, d.V), lookup[:, :d.V])
    assert not x[e + d.E * d.V:e + m].any()
    assert np.array_equal(x[e + m:], uniform[e + m:]) and np.abs(x[e + m:]).max() <= 0.08
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    f, g = tr.JdJ((tok, lens, img, lab))
    ref = orc.Oracle(np.float64).step(d, x, tok, lens, img, lab, orc.Dropout(1, 0.5, tr.dropout_seed, 0))
    assert abs(f - ref["loss"]) <= 2e-6 * abs(ref["loss"])
    tr.close()


def test_param_norms_and_arch2_log_line(pkg, orc):
    """nvqa_param_norms = torch.norm of the three parameter vectors (003_train_vqa_arch2/002_train_baseline.lua:401-403:
    cnn_w, encoder_w_q, multimodal_w; TH accumulates a FloatTensor's norm in double), reduced on the device, and the log
    line lua/train_arch2.lua and VQATrainer.log_line() build from it (:404)."""
    from util import gdims
    for kw in (dict(arch=2, B=8, T=6, V=300, E=64, R=64, L=2, I=128, C=4, A=40),
               dict(arch=1, B=8, T=6, V=300, E=40, R=64, L=2, I=128, C=48, A=40)):
        d = orc.make_dims(**kw)
        params = orc.synth_params(d)
        tr = pkg.trainer.VQATrainer(gdims(pkg, d), 0, seed=123)
        tr.set_params(params)
        seg = tr.ctx.segments()
        x = tr.get_params().astype(np.float64)
        exp, off = [], 0
        for n in seg:
            exp.append(np.sqrt((x[off:off + n] ** 2).sum()))
            off += n
        got = tr.ctx.param_norms()
        assert np.allclose(got, exp, rtol=3e-7, atol=0), (got, exp)
        tr.running_avg, tr.iter = 2.34567, 1200
        line = tr.log_line()
        if d.arch == 2:
            assert line == "iter: %6d train loss: %.3f cnn_norm: %.3f enc_norm: %.3f mm_norm: %.3f" % (1200, 2.34567, exp[0], exp[1], exp[2])
        tr.close()
