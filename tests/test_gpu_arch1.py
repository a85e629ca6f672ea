"""HIP path vs CPU oracle through the C ABI (arch1).  Tolerances: logits 1e-4 relative
(BASELINE.json north_star), argmax exact, gradients 1e-3 of the tensor's max (fp32
summation order differs: MFMA k-order vs the oracle's row order; quirk Q6)."""
import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu

CASES = {
    # name: (dims kwargs, full_length, dropout mode)
    "tiny_ragged": (dict(B=5, T=6, V=11, E=8, R=8, L=2, I=12, C=12, A=8), False, 0),
    "tiny_ragged_dropout": (dict(B=5, T=6, V=11, E=8, R=8, L=2, I=12, C=12, A=8), False, 1),
    "odd_sizes": (dict(B=37, T=9, V=301, E=20, R=36, L=3, I=52, C=44, A=28), False, 1),
    "one_layer": (dict(B=33, T=7, V=64, E=12, R=40, L=1, I=24, C=20, A=12), False, 1),
    "mid_full": (dict(B=16, T=26, V=1000, E=200, R=512, L=2, I=4096, C=1024, A=1000), True, 0),
    "mid_ragged_dropout": (dict(B=16, T=26, V=1000, E=200, R=512, L=2, I=4096, C=1024, A=1000), False, 1),
}


@pytest.mark.parametrize("name", list(CASES))
def test_step_matches_oracle(pkg, orc, name):
    kw, full, mode = CASES[name]
    d = orc.make_dims(arch=1, **kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=full)
    dr = orc.Dropout(mode, 0.5, 123, 5)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, dr)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
    grads = ctx.get_grads()
    assert abs(loss - ref["loss"]) <= 1e-5 * abs(ref["loss"]), (loss, ref["loss"])
    errs = segment_errors(orc, d, grads, ref["grads"])
    bad = {k: e for k, e in errs.items() if e > 1e-3}
    assert not bad, bad
    # eval-mode forward: logits and argmax
    ev = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, None, train=False)
    scores, argmax = ctx.forward(tok, lens, img)
    assert relmax(scores, ev["scores"]) <= 1e-4
    top2 = np.sort(ev["scores"], 1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * np.abs(top2[:, 1])  # ties below tolerance are unpinned
    assert np.array_equal(argmax[clear], ev["argmax"][clear])
    ctx.close()


def test_rmsprop_matches_oracle(pkg, orc):
    d = orc.make_dims(arch=1, B=8, T=6, V=40, E=12, R=16, L=2, I=32, C=24, A=12)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    o32 = orc.Oracle(np.float32)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    x = params.copy()
    m = np.zeros_like(x)
    lr = 3e-4
    for it in range(3):
        dr = orc.Dropout(1, 0.5, 123, it)
        g = o32.step(d, x, tok, lens, img, lab, dr)["grads"] * 50.0  # scaled so the clamp bites
        ctx.step(tok, lens, img, lab, gdrop(pkg, dr))
        # feed the oracle's own update with the DEVICE gradient so only the update is compared
        gdev = ctx.get_grads() * 50.0
        assert relmax(gdev, g) < 1e-3
        ctx.rmsprop_update(lr, 0.99, 1e-8, 1e-4, 10.0 / 50.0)
        gd = ctx.get_grads().copy()
        o32.rmsprop(x, gd, m, lr, 0.99, 1e-8, 1e-4, 10.0 / 50.0)
        assert relmax(ctx.get_params(), x) < 1e-6
        lr *= 0.99997592083
    ctx.close()


def test_bad_inputs_are_rejected(pkg, orc):
    d = orc.make_dims(arch=1, B=4, T=5, V=11, E=8, R=8, L=2, I=12, C=12, A=8)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    bad = tok.copy()
    bad[0, -1] = d.V + 1
    with pytest.raises(pkg.binding.NvqaError):
        ctx.step(bad, lens, img, lab)
    bl = lab.copy()
    bl[1] = 0
    with pytest.raises(pkg.binding.NvqaError):
        ctx.step(tok, lens, img, bl)
    with pytest.raises(pkg.binding.NvqaError):
        ctx.get_grads()  # no step has run
    ctx.close()
    with pytest.raises(pkg.binding.NvqaError):
        pkg.binding.Context(gdims(pkg, orc.make_dims(arch=1, E=6)), 0)  # E % 4 != 0
