"""HIP path vs CPU oracle through the C ABI (arch2: image-as-first-token encoder).
Same tolerances as test_gpu_arch1.py."""
import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu

CASES = {
    "tiny": (dict(B=5, T=6, V=11, E=8, R=8, L=1, I=12, C=12, A=8), 0),
    "tiny_dropout_2layer": (dict(B=5, T=6, V=11, E=8, R=8, L=2, I=12, C=12, A=8), 1),
    "odd_sizes": (dict(B=37, T=9, V=301, E=20, R=36, L=3, I=52, C=4, A=28), 1),
    "mid": (dict(B=16, T=26, V=1000, E=512, R=512, L=1, I=4096, C=4, A=1000), 1),
    "mid_deeper_inception": (dict(B=16, T=26, V=1000, E=512, R=512, L=2, I=2048, C=4, A=1000), 1),
}


def _run(pkg, orc, kw, mode, short_to=None):
    d = orc.make_dims(arch=2, **kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    if short_to is not None:  # every question ends early: exercises the tmax cut (Encoder_lstm.lua:185-189)
        tok[:, short_to:] = 0
    dr = orc.Dropout(mode, 0.5, 123, 5)
    ref = orc.Oracle(np.float64).step(d, params, tok, None, img, lab, dr)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step(tok, None, img, lab, gdrop(pkg, dr))
    grads = ctx.get_grads()
    assert abs(loss - ref["loss"]) <= 1e-5 * abs(ref["loss"]), (loss, ref["loss"])
    bad = {k: e for k, e in segment_errors(orc, d, grads, ref["grads"]).items() if e > 1e-3}
    assert not bad, bad
    ev = orc.Oracle(np.float64).step(d, params, tok, None, img, lab, None, train=False)
    scores, argmax = ctx.forward(tok, None, img)
    assert relmax(scores, ev["scores"]) <= 1e-4
    top2 = np.sort(ev["scores"], 1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * np.abs(top2[:, 1])
    assert np.array_equal(argmax[clear], ev["argmax"][clear])
    ctx.close()


@pytest.mark.parametrize("name", list(CASES))
def test_step_matches_oracle(pkg, orc, name):
    kw, mode = CASES[name]
    _run(pkg, orc, kw, mode)


def test_early_termination(pkg, orc):
    _run(pkg, orc, CASES["tiny_dropout_2layer"][0], 1, short_to=3)
    _run(pkg, orc, CASES["odd_sizes"][0], 1, short_to=1)


def test_weight_decay_update(pkg, orc):
    # arch2 baseline: optimize.weightDecay = 1e-4, applied after the clamp (Q7)
    d = orc.make_dims(arch=2, B=8, T=6, V=40, E=12, R=16, L=1, I=32, C=4, A=12)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    ctx.step(tok, None, img, lab, None)
    g = ctx.get_grads() * 1.0
    x, m = params.copy(), np.zeros_like(params)
    ctx.rmsprop_update(3e-4, 0.99, 1e-8, 1e-4, 10.0)
    orc.Oracle(np.float32).rmsprop(x, g, m, 3e-4, 0.99, 1e-8, 1e-4, 10.0)
    assert relmax(ctx.get_params(), x) < 1e-6
    ctx.close()
