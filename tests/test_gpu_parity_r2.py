"""Round-2 parity cases (VERDICT r1 "What's weak", parity 2-5), all through the C ABI against the f64 oracle:

* the exact bench.py workload -- B = 512, ALL lengths 26, dropout on -- at full size;
* a saturated regime (parameters x10..15: gates pinned at 0 / 1, decisive logits) where the argmax must be
  bit-exact on every row, for arch1 and arch2;
* logits element by element (|a-b| <= 1e-4 |b| + 6e-6 max_row|b|: tests/util.py RTOL_LOGIT_ROW), gradients per tensor in
  max-norm AND L2-norm;
* two different batches in a row on ONE context (long questions, then short ones / another tmax): every stale
  activation of step A that step B must not see (inactive rows of Gt/Hs/Cs/U/X0, skipped row tiles, skipped
  K-tiles), under the default kernels and NVQA_FOLD_I2H=0;
* arch2's reference quirks (nvqa_set_ref_quirks) over three RMSprop iterations;
* (round 3) the persistent two-chain BPTT kernel as the DEFAULT path: against the oracle, bit-reproducible, against the
  per-level fallback (NVQA_PERSIST_BWD=0), and its bounded give-up path (a spin limit of a few polls).
Measured errors are appended to gpurun_out/parity_r04.jsonl (committed as profiles/r0N_parity_measured_errors.jsonl per round); the
tolerances below are ~10x those measurements."""
import os

import numpy as np
import pytest

from util import (assert_argmax_all_rows, assert_grads, assert_logits, gdims, gdrop, record, relmax)

pytestmark = pytest.mark.gpu

FULL1 = dict(arch=1, B=512, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000)
TOL_GRAD = 2e-5       # f32 MFMA chains vs f64: measured worst 1.6e-6 (init regime), see gpurun_out/parity_r02.jsonl
TOL_GRAD_SAT = 2e-4   # saturated regime: K = 512..4096 sums of O(1) terms


def _ctx(pkg, d, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return pkg.binding.Context(gdims(pkg, d), 0)  # the switches are read at nvqa_create
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _check_step(pkg, orc, d, ctx, params, batch, dr, tol, name, all_rows=False):
    tok, lens, img, lab = batch
    lens = lens if d.arch == 1 else None
    o64 = orc.Oracle(np.float64)
    ref = o64.step(d, params, tok, lens, img, lab, dr)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr) if dr is not None else None)
    grads = ctx.get_grads()
    el = abs(loss - ref["loss"]) / abs(ref["loss"])
    assert el <= 2e-6, (loss, ref["loss"])
    worst = assert_grads(orc, d, grads, ref["grads"], tol, name)
    ev = o64.step(d, params, tok, lens, img, lab, None, train=False)
    scores, argmax = ctx.forward(tok, lens, img)
    e = assert_logits(scores, ev["scores"])
    share = assert_argmax_all_rows(argmax, ev["scores"], ev["argmax"])
    if all_rows:
        assert share >= 0.99, f"only {share:.3f} of the rows are decisive: the case is not in the saturated regime"
    record(name, {"loss_rel": el, "grad_worst": worst, "logit_worst_x_tol": e, "decisive_rows": share,
                  "max_abs_logit": float(np.abs(ev["scores"]).max())})
    return ref


def test_headline_workload_all_lengths_26_dropout_on(pkg, orc):
    """What bench.py times: arch1, B=512, every question 26 tokens, dropout 0.5 on."""
    d = orc.make_dims(**FULL1)
    params = orc.synth_params(d)
    batch = orc.synth_batch(d, full_length=True)
    ctx = _ctx(pkg, d)
    ctx.set_params(params)
    _check_step(pkg, orc, d, ctx, params, batch, orc.Dropout(1, 0.5, 123, 7), TOL_GRAD, "headline_all26")
    ctx.close()


def _saturating_params(orc, d, seed=11):
    """Saturated gates WITHOUT chaotic dynamics: the weights stay at the init scale (so the recurrence contracts and
    f32 rounding does not amplify over time) while the gate biases are pushed to +-(2..6): sigmoids sit near 0 / 1,
    tanh near +-1; the embedding, fusion and classifier weights are scaled so that tanh saturates there too and the
    logits are far apart (decisive argmax)."""
    rng = np.random.default_rng(seed)
    p = orc.synth_params(d).copy()
    lo = orc.layout(d)
    for l in range(d.L):
        o, n = lo[f"b_i2h{l}"]
        p[o:o + n] = (rng.uniform(2.0, 6.0, n) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    for k, f in (("w_e", 25.0), ("w_lk", 25.0), ("w_q", 10.0), ("w_v", 40.0), ("w_p", 40.0), ("w_o", 40.0)):
        if k in lo:
            o, n = lo[k]
            p[o:o + n] *= np.float32(f)
    return p


SAT_CASES = {
    "satbias_arch1_mid": dict(arch=1, B=96, T=12, V=300, E=64, R=128, L=2, I=256, C=192, A=100),
    "satbias_arch1_full": FULL1,
    "satbias_arch2_mid": dict(arch=2, B=96, T=10, V=300, E=128, R=128, L=2, I=256, C=4, A=100),
    "satbias_arch2_full": dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000),
}


@pytest.mark.parametrize("name", list(SAT_CASES))
def test_saturated_gates_decisive_argmax(pkg, orc, name):
    """Gates pinned near 0 / 1 and logits far apart: loss, gradients and logits at the init-regime tolerances, argmax
    bit-exact on every decisive row (>= 99 % of the rows must be decisive)."""
    d = orc.make_dims(**SAT_CASES[name])
    params = _saturating_params(orc, d)
    batch = orc.synth_batch(d, full_length=False, min_len=2)
    ctx = _ctx(pkg, d)
    ctx.set_params(params)
    _check_step(pkg, orc, d, ctx, params, batch, orc.Dropout(1, 0.5, 123, 3), TOL_GRAD_SAT, name, all_rows=True)
    ctx.close()


@pytest.mark.parametrize("name,kw,scale", [
    ("chaos_arch1_mid", dict(arch=1, B=96, T=12, V=300, E=64, R=128, L=2, I=256, C=192, A=100), 12.0),
    ("chaos_arch1_full", FULL1, 3.0),   # x10 at full size: even the CPU f32 gradient is 140 % away from f64 (recorded round 2)
    ("chaos_arch2_mid", dict(arch=2, B=96, T=10, V=300, E=128, R=128, L=2, I=256, C=4, A=100), 12.0),
])
def test_large_weight_regime_is_as_accurate_as_cpu_f32(pkg, orc, name, kw, scale):
    """Every parameter x10..12: the LSTM becomes an expanding map and f32 rounding is amplified step after step --
    the f32 CPU oracle itself leaves the f64 oracle by far more than 1e-4 here, so no f32 evaluation (Torch7's
    included) can hold the init-regime tolerance.  What can be required: the HIP path stays within 10x of the
    distance between the f32 and the f64 CPU evaluations, and the argmax agrees on every row whose top-2 gap
    exceeds twice that distance."""
    d = orc.make_dims(**kw)
    params = orc.synth_params(d) * np.float32(scale)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False, min_len=2)
    lens = lens if d.arch == 1 else None
    dr = orc.Dropout(1, 0.5, 123, 3)
    o64, o32 = orc.Oracle(np.float64), orc.Oracle(np.float32)
    r64, r32 = o64.step(d, params, tok, lens, img, lab, dr), o32.step(d, params, tok, lens, img, lab, dr)
    e64 = o64.step(d, params, tok, lens, img, lab, None, train=False)
    e32 = o32.step(d, params, tok, lens, img, lab, None, train=False)
    ctx = _ctx(pkg, d)
    ctx.set_params(params)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
    grads = ctx.get_grads()
    scores, argmax = ctx.forward(tok, lens, img)
    ctx.close()
    cpu = {"loss": abs(r32["loss"] - r64["loss"]) / abs(r64["loss"]), "grad": relmax(r32["grads"], r64["grads"]),
           "logit": relmax(e32["scores"], e64["scores"])}
    gpu = {"loss": abs(loss - r64["loss"]) / abs(r64["loss"]), "grad": relmax(grads, r64["grads"]),
           "logit": relmax(scores, e64["scores"])}
    record(name, {"cpu_f32_vs_f64": cpu, "hip_vs_f64": gpu})
    for k in cpu:
        assert gpu[k] <= 10 * cpu[k] + 1e-6, (k, gpu, cpu)
    s = e64["scores"]
    top2 = np.sort(s, 1)[:, -2:]
    decisive = (top2[:, 1] - top2[:, 0]) > 2 * gpu["logit"] * np.abs(s).max()
    assert decisive.mean() > 0.9 and np.array_equal(argmax[decisive], e64["argmax"][decisive])


SEQ_CASES = {
    "arch1": dict(arch=1, B=96, T=9, V=80, E=32, R=64, L=2, I=64, C=48, A=24),
    "arch2": dict(arch=2, B=80, T=8, V=80, E=64, R=64, L=2, I=64, C=8, A=24),
}


@pytest.mark.parametrize("env", [{}, {"NVQA_FOLD_I2H": "0"}], ids=["default", "nofold"])
@pytest.mark.parametrize("arch", ["arch1", "arch2"])
def test_second_batch_on_a_used_context(pkg, orc, arch, env):
    """Step A (long questions) then step B (short, other lengths / another tmax) on the same context: step B must
    equal the oracle's single step on B -- nothing of A may leak through buffers B only partly rewrites."""
    d = orc.make_dims(**SEQ_CASES[arch])
    params = orc.synth_params(d)
    tokA, lensA, imgA, labA = orc.synth_batch(d, seed=5, full_length=True)
    tokB, lensB, imgB, labB = orc.synth_batch(d, seed=9, full_length=False)
    if d.arch == 1:   # B: at most 4 tokens, so most time columns have few or no active rows
        lensB = np.minimum(lensB, 1 + np.arange(d.B) % 4).astype(np.int32)
        left = np.zeros_like(tokB)
        rng = np.random.default_rng(1)
        for b in range(d.B):
            left[b, :lensB[b]] = rng.integers(1, d.V + 1, lensB[b])
        tokB = orc.right_align(left, lensB)
    else:             # B: every question ends after 3 tokens -> tmax = 5 of 10 steps
        tokB[:, 3:] = 0
    ctx = _ctx(pkg, d, env)
    ctx.set_params(params)
    for rep in range(2):  # A, B, A, B: also B's leftovers under A
        _check_step(pkg, orc, d, ctx, params, (tokA, lensA, imgA, labA), orc.Dropout(1, 0.5, 123, 2 * rep), TOL_GRAD,
                    f"seq_{arch}_A{rep}")
        _check_step(pkg, orc, d, ctx, params, (tokB, lensB, imgB, labB), orc.Dropout(1, 0.5, 123, 2 * rep + 1), TOL_GRAD,
                    f"seq_{arch}_B{rep}")
    ctx.close()


@pytest.mark.parametrize("flags", [1, 2, 3])
@pytest.mark.parametrize("L", [1, 2])
def test_arch2_reference_quirks_over_three_iterations(pkg, orc, flags, L):
    """nvqa_set_ref_quirks vs the oracle's switch (misc/Encoder_lstm.lua:238-239 aliased h0; :49-58 lookup table
    without gradient): three JdJ + rmsprop iterations, the oracle restarted from the device parameters each time."""
    d = orc.make_dims(arch=2, B=40, T=7, V=60, E=32, R=48, L=L, I=64, C=4, A=20)
    params = orc.synth_params(d) * np.float32(8.0)   # large enough for dL/dh (the carried h0) to matter: 1e-2 of the gradient
    tok, _, img, lab = orc.synth_batch(d, full_length=False)
    o = orc.Oracle(np.float64)
    ctx = _ctx(pkg, d)
    ctx.set_params(params)
    ctx.set_ref_quirks(flags)
    o.set_ref_quirks(flags)
    plain = orc.Oracle(np.float32)  # the f32 library holds the un-quirked result for the "it matters" check
    try:
        for it in range(3):
            x = ctx.get_params()
            dr = orc.Dropout(1, 0.5, 123, it)
            ref = o.step(d, x, tok, None, img, lab, dr)
            loss = ctx.step(tok, None, img, lab, gdrop(pkg, dr))
            g = ctx.get_grads()
            assert abs(loss - ref["loss"]) <= 2e-6 * abs(ref["loss"]), (it, loss, ref["loss"])
            assert_grads(orc, d, g, ref["grads"], TOL_GRAD_SAT, f"quirks{flags}_L{L}_it{it}")
            if flags & 1 and it > 0:
                nq = plain.step(d, x, tok, None, img, lab, dr)
                assert relmax(g, nq["grads"]) > 5 * TOL_GRAD_SAT, "the carried h0 must change the gradient measurably"
            if flags & 2:
                lo = orc.layout(d)["w_lk"]
                assert np.all(g[lo[0]:lo[0] + lo[1]] == 0)
            ctx.rmsprop_update(3e-3, 0.99, 1e-8, 1e-4, 10.0)
        ev = o.step(d, ctx.get_params(), tok, None, img, lab, None, train=False)
        scores, _ = ctx.forward(tok, None, img)     # validate(): evaluate mode reads the carried h0 too
        assert_logits(scores, ev["scores"])
        ctx.set_ref_quirks(0)                        # and it switches off: a clean h0 again
        o.set_ref_quirks(0)
        ev0 = o.step(d, ctx.get_params(), tok, None, img, lab, None, train=False)
        scores0, _ = ctx.forward(tok, None, img)
        assert_logits(scores0, ev0["scores"])
    finally:
        o.set_ref_quirks(0)
    ctx.close()


PERSIST_CASES = {   # (dims, batch A full-length?, batch B: uniform length or None = ragged [arch1: falls back to the level path])
    "arch1_all26": (FULL1, True, 9),
    "arch1_ragged_then_back": (FULL1, True, None),
    "arch1_ragged_first": (FULL1, False, None),     # a ragged batch on a fresh context (no rows left over from a full one), then another
    "arch2_L2": (dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000), False, None),
    "arch2_L1": (dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=1, I=4096, C=4, A=1000), False, None),
    "arch1_B200": (dict(FULL1, B=200), True, 5),    # 13 row tiles: partial row block, fewer workgroups than CUs
    "arch1_L1_ragged": (dict(FULL1, L=1), False, None),  # one layer: 4 row tiles per workgroup (MT = 4), ragged instance
    "arch1_E512_ragged": (dict(FULL1, E=512), False, None),  # an arch1 model as wide as arch2: the E = 512 instances with RAG
}


@pytest.mark.parametrize("name", list(PERSIST_CASES))
def test_persistent_forward_lstm(pkg, orc, name):
    """The forward unroll as ONE weight-stationary launch (csrc/lstm_persist.h; taken for arch2 and for arch1 batches of
    one question length) against the f64 oracle at the init-regime tolerances; two different batches in a row (counters
    re-zeroed, no stale state; a ragged arch1 batch in between goes through the per-level kernels on the same buffers)
    and the second one repeated (bit-reproducible)."""
    kw, full, lenB = PERSIST_CASES[name]
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    bA = orc.synth_batch(d, seed=123, full_length=full, min_len=3)
    bB = orc.synth_batch(d, seed=77, full_length=False, min_len=1)
    if d.arch == 2:
        bB[0][:, 9:] = 0   # another tmax
    elif lenB is not None:  # every question lenB tokens long: the row blocks start at step T - lenB
        tok, lens, img, lab = bB
        lens[:] = lenB
        left = np.zeros_like(tok)
        left[:, :lenB] = np.random.default_rng(5).integers(1, d.V + 1, (d.B, lenB))
        bB = (orc.right_align(left, lens), lens, img, lab)
    ctx = _ctx(pkg, d, {"NVQA_PERSIST": "1"})
    ctx.set_params(params)
    _check_step(pkg, orc, d, ctx, params, bA, orc.Dropout(1, 0.5, 123, 7), TOL_GRAD, f"persist_{name}_A")
    _check_step(pkg, orc, d, ctx, params, bB, orc.Dropout(1, 0.5, 123, 8), TOL_GRAD, f"persist_{name}_B")
    tok, lens, img, lab = bB
    lens = lens if d.arch == 1 else None
    dr = gdrop(pkg, orc.Dropout(1, 0.5, 123, 8))
    l1 = ctx.step(tok, lens, img, lab, dr)
    g1 = ctx.get_grads()
    l2 = ctx.step(tok, lens, img, lab, dr)
    g2 = ctx.get_grads()
    assert l1 == l2 and np.array_equal(g1, g2)
    ctx.close()


@pytest.mark.parametrize("name", ["arch1_all26", "arch2_L2", "arch2_L1", "arch1_ragged", "arch1_L1_ragged"])
def test_persistent_bptt(pkg, orc, name):
    """BPTT as one persistent launch with three workgroup roles and two independent row chains per workgroup
    (csrc/lstm_persist_bwd2.h) -- the DEFAULT path at R = 512 since round 3 -- against the f64 oracle at the init-regime
    tolerances, and bit-identical gradients from the per-level fallback route's point of view is NOT required (other
    summation order): what is required is bit-reproducibility of the route itself."""
    # (arch1_ragged: question lengths 3 .. 26 -- both persistent kernels run their RAG instances, which skip the row tiles
    # without active rows step by step)
    kw, full, _ = PERSIST_CASES[name] if name != "arch1_ragged" else (FULL1, False, None)
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    ctx = _ctx(pkg, d, {})
    ctx.set_params(params)
    for it, seed in enumerate((123, 77)):
        b = orc.synth_batch(d, seed=seed, full_length=full, min_len=3)
        if d.arch == 2 and it:
            b[0][:, 11:] = 0
        _check_step(pkg, orc, d, ctx, params, b, orc.Dropout(1, 0.5, 123, 20 + it), TOL_GRAD, f"persist_bwd_{name}_{it}")
    # bit-reproducible: the same step twice (fixed summation order: K order per output, wave order of the K-quarters)
    tok, lens, img, lab = b
    lens = lens if d.arch == 1 else None
    dr = gdrop(pkg, orc.Dropout(1, 0.5, 123, 30))
    l1 = ctx.step(tok, lens, img, lab, dr)
    g1 = ctx.get_grads()
    l2 = ctx.step(tok, lens, img, lab, dr)
    assert l1 == l2 and np.array_equal(g1, ctx.get_grads())
    ctx.close()
    # the per-level fallback (NVQA_PERSIST_BWD=0) stays parity-green on the same case
    ctx = _ctx(pkg, d, {"NVQA_PERSIST_BWD": "0"})
    ctx.set_params(params)
    _check_step(pkg, orc, d, ctx, params, b, orc.Dropout(1, 0.5, 123, 21), TOL_GRAD, f"levels_bwd_{name}")
    ctx.close()


def test_ride_along_jobs(pkg, orc):
    """Work that rides in the idle workgroups of the persistent BPTT launch (csrc/ride_jobs.h): the token index of the embedding
    gradient and the head weight gradients dW_o, dW_q (arch1) / dW_o (arch2).  Same K order per output as the kernels they
    replace, so the gradients must be BIT-IDENTICAL to a context with the jobs in their own launches (NVQA_RIDE_GEMM=0,
    NVQA_TOK_IN_BPTT=0), in f32 and in bf16 mode; against the oracle the default path is covered by every other test."""
    for kw, bf16 in ((FULL1, False), (FULL1, True), (PERSIST_CASES["arch2_L2"][0], False)):
        d = orc.make_dims(**kw)
        params = orc.synth_params(d)
        tok, lens, img, lab = orc.synth_batch(d, seed=5, full_length=d.arch == 2, min_len=3)
        lens = lens if d.arch == 1 else None
        got = []
        for env in ({}, {"NVQA_RIDE_GEMM": "0", "NVQA_TOK_IN_BPTT": "0"}):
            ctx = _ctx(pkg, d, env)
            ctx.set_params(params)
            if bf16:
                ctx.set_precision(1)
            for it in range(2):  # the second step reuses the device copy of the job list
                loss = ctx.step(tok, lens, img, lab, gdrop(pkg, orc.Dropout(1, 0.5, 123, 40 + it)))
            got.append((loss, ctx.get_grads()))
            ctx.close()
        assert got[0][0] == got[1][0]
        assert np.array_equal(got[0][1], got[1][1]), float(np.abs(got[0][1] - got[1][1]).max())


def test_image_projection_rides_in_the_forward_launch(pkg, orc):
    """arch1, f32: layer 0's workgroups of the persistent forward launch finish ~30 % before the launch does (K = E + R against
    2R per step, and nothing waits for them) and then multiply the head's image projection W_v Dropout(v) -- 128 tiles of
    64 x 64 x 4096 -- instead of the head's own split-K launch doing it on the critical path (lstm_persist.h:
    PersistFwdArgs::fr; NVQA_RIDE_FWD=0 switches it off).  The K order of a tile differs from the split-K form, so the two
    routes agree to f32 summation-order noise, not bit for bit; against the oracle the default route is held by every other
    arch1 test of this file.  Also at B = 500 (edge tiles) and on a forward-only call."""
    for kw, full in ((FULL1, True), ({**FULL1, "B": 500}, False)):
        d = orc.make_dims(**kw)
        params = orc.synth_params(d)
        tok, lens, img, lab = orc.synth_batch(d, seed=6, full_length=full, min_len=3)
        got = []
        for env in ({}, {"NVQA_RIDE_FWD": "0"}):
            ctx = _ctx(pkg, d, env)
            ctx.set_params(params)
            loss = ctx.step(tok, lens, img, lab, gdrop(pkg, orc.Dropout(1, 0.5, 123, 60)))
            grads = ctx.get_grads()
            scores, argmax = ctx.forward(tok, lens, img)
            l2 = ctx.step(tok, lens, img, lab, gdrop(pkg, orc.Dropout(1, 0.5, 123, 60)))
            assert l2 == loss and np.array_equal(ctx.get_grads(), grads)   # bit-reproducible on either route
            got.append((loss, grads, scores))
            ctx.close()
        assert abs(got[0][0] - got[1][0]) <= 1e-6 * abs(got[1][0])
        assert_logits(got[0][2], got[1][2], scale=0.2)
        assert_grads(orc, d, got[0][1], got[1][1], 2e-6, "fwd_ride_vs_launch")


def test_ride_along_jobs_without_a_free_slot(pkg, orc):
    """L = 1, B = 1024: 16 row blocks x 16 unit tiles fill all 256 slots of the BPTT grid -- no workgroup without a role.  The
    jobs must then run in their own launches behind the BPTT (ride_flush / emb_backward's fallback): against the oracle."""
    d = orc.make_dims(arch=1, B=1024, T=4, V=300, E=200, R=512, L=1, I=64, C=64, A=40)
    params = orc.synth_params(d)
    ctx = _ctx(pkg, d, {})
    ctx.set_params(params)
    b = orc.synth_batch(d, seed=9, full_length=False, min_len=2)
    _check_step(pkg, orc, d, ctx, params, b, orc.Dropout(1, 0.5, 123, 50), TOL_GRAD, "ride_no_free_slot")
    ctx.close()


def test_persistent_kernel_timeout_is_reported_and_survived(pkg, orc, monkeypatch):
    """The give-up path of the persistent kernels: with NVQA_PF_SPIN = 1 every cross-workgroup wait gives up at
    its second poll, so a full-size step MUST time out.  Required: the launch drains (no hang), the failure survives an
    asynchronous trainer loop -- nvqa_step(loss_out = NULL) + nvqa_rmsprop_update before any synchronisation: the sticky
    record makes k_rmsprop skip the update, the parameters stay bit-identical -- nvqa_sync returns an error that names the
    counter, and the following steps (per-level kernels: the persistent paths switch themselves off) match the oracle."""
    d = orc.make_dims(**FULL1)
    params = orc.synth_params(d)
    monkeypatch.setenv("NVQA_PF_SPIN", "1")
    ctx = pkg.binding.Context(gdims(pkg, d), 0)   # the limit is read at nvqa_create
    monkeypatch.delenv("NVQA_PF_SPIN")
    ctx.set_params(params)
    p0 = ctx.get_params()
    tok, lens, img, lab = orc.synth_batch(d, seed=123, full_length=True)
    ctx.step(tok, lens, img, lab, gdrop(pkg, orc.Dropout(1, 0.5, 123, 1)), want_loss=False)
    ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, 10.0)   # enqueued behind the failed step, before the host knows
    with pytest.raises(pkg.binding.NvqaError) as e:
        ctx.sync()
    msg = str(e.value)
    assert "timed out" in msg and "counter word" in msg and "not applied" in msg, msg
    assert np.array_equal(ctx.get_params(), p0), "the update of a failed step was applied"
    # the context lives on: the next steps run the per-level kernels and are right
    b = orc.synth_batch(d, seed=77, full_length=True)
    _check_step(pkg, orc, d, ctx, params, b, orc.Dropout(1, 0.5, 123, 2), TOL_GRAD, "after_timeout")
    ctx.close()
