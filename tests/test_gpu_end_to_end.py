"""BASELINE.json configs[4]: the VGG-16 fc7 extractor fused in front of the arch1 training step
(features never leave the device) against oracle(extractor) -> L2 norm -> oracle(step)."""
import numpy as np
import pytest

from util import gdims, gdrop, segment_errors

pytestmark = pytest.mark.gpu


def test_images_to_gradients(pkg, orc):
    div, hw = 16, 32                      # reduced-width extractor: fc7 width 4096/16 = 256
    vo = orc.VggOracle(div, hw)
    d = orc.make_dims(arch=1, B=6, T=7, V=40, E=12, R=16, L=2, I=vo.feature_dim, C=24, A=12)
    params = orc.synth_params(d)
    tok, lens, _, lab = orc.synth_batch(d, full_length=False)
    w = vo.synth_weights()
    rng = np.random.default_rng(8)
    images = rng.uniform(-100, 120, (d.B, 3, hw, hw)).astype(np.float32)
    feats = vo.fc7(w, images)
    assert (feats > 0).mean() > 0.05
    fn = feats / np.sqrt((feats.astype(np.float64) ** 2).sum(1, keepdims=True))
    dr = orc.Dropout(1, 0.5, 123, 4)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, fn, lab, dr)
    v = pkg.binding.Vgg16(0, div, hw, max_batch=d.B)
    v.set_weights(w)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step_images(v, images, tok, lens, lab, gdrop(pkg, dr))
    assert abs(loss - ref["loss"]) <= 2e-5 * abs(ref["loss"])
    bad = {k: e for k, e in segment_errors(orc, d, ctx.get_grads(), ref["grads"]).items() if e > 2e-3}
    assert not bad, bad
    with pytest.raises(pkg.binding.NvqaError):   # feature width mismatch is an error, not a silent reshape
        bad_ctx = pkg.binding.Context(gdims(pkg, orc.make_dims(arch=1, B=6, T=7, V=40, E=12, R=16, L=2, I=64, C=24, A=12)), 0)
        bad_ctx.step_images(v, images, tok, lens, lab, None)
    ctx.close()
    v.close()


def test_images_to_gradients_full_width(pkg, orc):
    """The same path at the reference's sizes: the FULL VGG-16 (width_div 1, 224 x 224 inputs, fc7 width
    4096 = the arch1 default -nhimage) in front of an arch1 step with the reference's E = 200, R = 512, L = 2, C = 1024,
    A = 1000 -- the persistent LSTM kernels run -- and B = 8 images: oracle(VGG-16) -> L2 norm -> oracle(step) in f64
    against nvqa_step_images."""
    div, hw = 1, 224
    vo = orc.VggOracle(div, hw)
    assert vo.feature_dim == 4096
    d = orc.make_dims(arch=1, B=8, T=6, V=400, E=200, R=512, L=2, I=vo.feature_dim, C=1024, A=1000)
    params = orc.synth_params(d)
    tok, lens, _, lab = orc.synth_batch(d, full_length=False)
    w = vo.synth_weights()
    rng = np.random.default_rng(9)
    images = rng.uniform(-100, 120, (d.B, 3, hw, hw)).astype(np.float32)
    feats = vo.fc7(w, images)
    assert (feats > 0).mean() > 0.05
    fn = feats / np.sqrt((feats.astype(np.float64) ** 2).sum(1, keepdims=True))
    dr = orc.Dropout(1, 0.5, 123, 4)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens, fn, lab, dr)
    v = pkg.binding.Vgg16(0, div, hw, max_batch=d.B)
    v.set_weights(w)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    loss = ctx.step_images(v, images, tok, lens, lab, gdrop(pkg, dr))
    assert abs(loss - ref["loss"]) <= 2e-5 * abs(ref["loss"])
    errs = segment_errors(orc, d, ctx.get_grads(), ref["grads"])
    from util import record
    record("e2e_full_width", {"loss_rel": abs(loss - ref["loss"]) / abs(ref["loss"]), "grad_relmax": {k: float(e) for k, e in errs.items()}})
    bad = {k: e for k, e in errs.items() if e > 2e-3}
    assert not bad, bad
    ctx.close()
    v.close()
