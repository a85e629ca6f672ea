"""BASELINE.json configs[3] ("arch2 deeper LSTM + Inception-v3 feats, bf16"): nvqa_set_precision(ctx, 1) runs
every dense product on the bf16 matrix cores (operands rounded to bf16 as they are read, f32 accumulate); the
ABI stays f32.  The oracle has the same operand-rounding mode (pinned on the CPU against an independent
autograd statement, tests/test_oracle.py).

Tolerance: the HIP path and the f32 oracle round the same f32 values except where their f32 activations differ
in the last bits (summation order), which occasionally flips a bf16 rounding (a 2^-8 relative change of one
operand).  So: loss 1e-4 relative, logits 1e-3 of the largest logit, gradient segments 5e-3 of the segment's
largest entry -- two orders tighter than the distance between the bf16 and the f32 result, which the test also
checks so that the mode is known to be on."""
import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    # arch2 "deeper LSTM + Inception": L = 2, I = 2048 (001_prepro_img_inc.lua:82), scaled-down vocabulary / batch
    dict(arch=2, B=48, T=8, V=120, E=64, R=64, L=2, I=2048, C=8, A=40),
    dict(arch=2, B=33, T=5, V=60, E=32, R=32, L=1, I=40, C=8, A=12),
    dict(arch=1, B=48, T=7, V=80, E=40, R=64, L=2, I=96, C=72, A=32),
])
def test_bf16_step_matches_bf16_oracle(pkg, orc, kw):
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    lens = lens if d.arch == 1 else None
    odr = orc.Dropout(1, 0.5, 123, 4)
    o = orc.Oracle(np.float32)
    exact = o.step(d, params, tok, lens, img, lab, odr)
    o.set_precision(1)
    try:
        ref = o.step(d, params, tok, lens, img, lab, odr)
        ev = o.step(d, params, tok, lens, img, lab, None, train=False)
    finally:
        o.set_precision(0)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    ctx.set_precision(1)
    loss = ctx.step(tok, lens, img, lab, gdrop(pkg, odr))
    grads = ctx.get_grads()
    scores, argmax = ctx.forward(tok, lens, img)
    assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (loss, ref["loss"])
    errs = segment_errors(orc, d, grads, ref["grads"])
    assert max(errs.values()) < 5e-3, errs
    assert relmax(scores, ev["scores"]) < 1e-3
    # the mode is really on: every weight gradient sits at a bf16-sized distance from the f32 result, and is at
    # least twice as close to the bf16 oracle as to the f32 one
    dist = segment_errors(orc, d, grads, exact["grads"])
    for k in errs:
        if k.startswith("w_"):
            assert dist[k] > 5e-4 and errs[k] < 0.5 * dist[k], (k, errs[k], dist[k])
    # and it switches off again: back to the f32 path within the f32 tolerance
    ctx.set_precision(0)
    l32 = ctx.step(tok, lens, img, lab, gdrop(pkg, odr))
    assert abs(l32 - exact["loss"]) <= 1e-5 * abs(exact["loss"])
    ctx.close()
