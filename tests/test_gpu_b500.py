"""The reference's OWN default batch: `-batch_size 500` (002_train_vqa_arch1/002_train_baseline.lua:31; arch2
002_train_baseline.lua:34), kept as the default of lua/train_arch1.lua / train_arch2.lua (parity item 2).

500 = 31 row tiles of 16 + a last tile of 4 rows: the persistent forward kernel's last row block (MT = 8: 128 rows) holds
116 rows, the BPTT kernel's last block (7 row tiles: 112 rows) holds 52, with 12 dead rows in the last tile of both -- the
`iloc < nloc` masks, the zero-filled out-of-range loads and the ride-along products' M = 500 edge tiles, which no R = 512
case with B in {8, 16, 512, 1024} ever exercised.  All through the C ABI against the f64 oracle at the tolerances of the
B = 512 cases: arch1 all-26 and ragged, arch2 L = 2, f32 here and bf16 in test_gpu_bf16.py (FULL_BF16 "..._B500"); the
ride-along jobs bit-identical to their own launches; nvqa_evaluate with a short last batch."""
import numpy as np
import pytest

from test_gpu_parity_r2 import TOL_GRAD, _check_step, _ctx
from util import gdrop

pytestmark = pytest.mark.gpu

ARCH1_500 = dict(arch=1, B=500, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000)
ARCH2_500 = dict(arch=2, B=500, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000)
CASES = {
    "b500_arch1_all26": (ARCH1_500, True),
    "b500_arch1_ragged": (ARCH1_500, False),
    "b500_arch2_L2": (ARCH2_500, False),
    "b500_arch1_L1_ragged": ({**ARCH1_500, "L": 1}, False),   # the MT = 4 forward instance and the 2 + 2 BPTT instance
}


@pytest.mark.parametrize("name", list(CASES))
def test_reference_default_batch_500(pkg, orc, name):
    kw, full = CASES[name]
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    batch = orc.synth_batch(d, seed=21, full_length=full, min_len=3)
    ctx = _ctx(pkg, d)
    import os   # (the fallback runs of tools/gpu/r4_fallbacks.sh switch the persistent kernels off on purpose)
    fwd = os.environ.get("NVQA_PERSIST", "1") != "0"
    assert ctx.persistent_state() == {"fwd": fwd, "bwd": fwd and os.environ.get("NVQA_PERSIST_BWD", "1") != "0"}   # the fast path is what is being held
    ctx.set_params(params)
    _check_step(pkg, orc, d, ctx, params, batch, orc.Dropout(1, 0.5, 123, 31), TOL_GRAD, name)
    # a second, different batch on the same context (stale rows of the first must not leak into the dead rows' neighbours)
    b2 = orc.synth_batch(d, seed=22, full_length=False, min_len=1)
    _check_step(pkg, orc, d, ctx, params, b2, orc.Dropout(1, 0.5, 123, 32), TOL_GRAD, name + "_second_batch")
    ctx.close()


@pytest.mark.parametrize("bf16", [False, True])
def test_ride_along_jobs_at_batch_500(pkg, orc, bf16):
    """dW_o = dscores^T z and dW_q = dqc^T q have K = B = 500 (not a multiple of the 64-deep K tile) when they ride in the
    BPTT launch's idle workgroups: bit-identical to the same products as launches of their own."""
    d = orc.make_dims(**ARCH1_500)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=5, full_length=False, min_len=3)
    got = []
    for env in ({}, {"NVQA_RIDE_GEMM": "0", "NVQA_TOK_IN_BPTT": "0"}):
        ctx = _ctx(pkg, d, env)
        ctx.set_params(params)
        if bf16:
            ctx.set_precision(1)
        for it in range(2):
            loss = ctx.step(tok, lens, img, lab, gdrop(pkg, orc.Dropout(1, 0.5, 123, 40 + it)))
        got.append((loss, ctx.get_grads()))
        ctx.close()
    assert got[0][0] == got[1][0]
    assert np.array_equal(got[0][1], got[1][1]), float(np.abs(got[0][1] - got[1][1]).max())


@pytest.mark.parametrize("kw", [ARCH1_500, ARCH2_500])
def test_evaluate_short_last_batch_at_batch_500(pkg, orc, kw):
    """validate() walks the validation split in batches of 500 and the last one is short (002_train_baseline.lua:343-347):
    nvqa_evaluate with n = 137 < B on the persistent forward kernel -- loss, argmax and the multiple-choice answer."""
    d = orc.make_dims(**kw)
    params = orc.synth_params(d) * np.float32(2.0)
    tok, lens, img, lab = orc.synth_batch(d, seed=3, full_length=False, min_len=2)
    lens = lens if d.arch == 1 else None
    rng = np.random.default_rng(2)
    mc = rng.integers(0, d.A + 1, (d.B, 18)).astype(np.int32)
    mc[:, 0] = np.maximum(mc[:, 0], 1)
    ctx = _ctx(pkg, d)
    ctx.set_params(params)
    from util import assert_argmax_all_rows, assert_logits
    for n in (500, 137):
        sl = slice(0, n)
        dn = orc.make_dims(**{**kw, "B": n})
        ref = orc.Oracle(np.float64).step(dn, params, tok[sl], None if lens is None else lens[sl], img[sl], lab[sl], None, train=False)
        r = ctx.evaluate(tok[sl], None if lens is None else lens[sl], img[sl], labels=lab[sl], mc_ans=mc[sl])
        assert abs(r["loss"] - ref["loss"]) <= 2e-6 * abs(ref["loss"]), (n, r["loss"], ref["loss"])
        assert_logits(r["scores"], ref["scores"])
        assert assert_argmax_all_rows(r["argmax"], ref["scores"], ref["argmax"]) > 0.9
        exp = pkg.trainer.multiple_choice_argmax(ref["scores"], mc[sl])
        s = np.asarray(ref["scores"], np.float64)
        # (multiple choice: compare where the best and second-best candidate are further apart than f32 noise)
        ok = []
        for i in range(n):
            c = mc[i][mc[i] != 0] - 1
            v = np.sort(s[i, np.unique(c)])
            ok.append(len(v) < 2 or v[-1] - v[-2] > 1e-3 * np.abs(s[i]).max())
        ok = np.asarray(ok)
        assert ok.mean() > 0.9 and np.array_equal(r["mc_argmax"][ok], exp[ok])
    ctx.close()
