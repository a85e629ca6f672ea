"""VGG-16 fc7 extractor (HIP implicit-GEMM convolutions) vs the direct-convolution CPU oracle.
Synthetic He-scaled weights: the Caffe model is not available offline (PARITY UNPINNED w.r.t. the
real caffemodel / cudnn; the arithmetic definition is the public VGG-16 layer list)."""
import numpy as np
import pytest

from util import relmax

pytestmark = pytest.mark.gpu


def _images(n, hw, seed=5):
    rng = np.random.default_rng(seed)
    rgb = rng.uniform(0, 1, (n, 3, hw, hw)).astype(np.float32)
    mean = np.array([103.939, 116.779, 123.68], np.float32).reshape(1, 3, 1, 1)
    return rgb[:, ::-1] * 255.0 - mean  # loadim: BGR planes, mean-subtracted


# (1, 224, 4): the reference's network at full width and full resolution, four images
@pytest.mark.parametrize("div,hw,n", [(16, 32, 3), (16, 64, 2), (8, 96, 2), (1, 64, 2), (1, 224, 1), (1, 224, 4)])
def test_fc7_matches_oracle(pkg, orc, div, hw, n):
    o = orc.VggOracle(div, hw)
    w = o.synth_weights()
    x = np.ascontiguousarray(_images(n, hw))
    ref = o.fc7(w, x)
    v = pkg.binding.Vgg16(0, div, hw, max_batch=4)
    assert v.weight_count == o.weight_count and v.feature_dim == o.feature_dim
    v.set_weights(w)
    got = v.fc7(x)
    assert (ref > 0).mean() > 0.05, "degenerate test: almost every feature is clipped by the ReLU"
    assert relmax(got, ref) < 1e-4
    # batch rows are independent
    assert relmax(v.fc7(x[:1]), ref[:1]) < 1e-4
    v.close()


def test_fc7_chunked_host_images(pkg, orc, monkeypatch):
    """More host images than one chunk (NVQA_VGG_CHUNK, default 256): the batch travels chunk by chunk on the copy
    stream while the network runs on the chunks before it; a ragged last chunk, two calls in a
    row (the second call's copies wait for the first call's network), then a one-chunk call on the same handle."""
    monkeypatch.setenv("NVQA_VGG_CHUNK", "3")
    div, hw, n = 16, 32, 11
    o = orc.VggOracle(div, hw)
    w = o.synth_weights()
    x = np.ascontiguousarray(_images(n, hw, seed=11))
    y = np.ascontiguousarray(_images(n, hw, seed=12))
    v = pkg.binding.Vgg16(0, div, hw, max_batch=n)
    v.set_weights(w)
    gx, gy, g2 = v.fc7(x), v.fc7(y), v.fc7(x[:2])
    assert relmax(gx, o.fc7(w, x)) < 1e-4 and relmax(gy, o.fc7(w, y)) < 1e-4
    assert relmax(g2, gx[:2]) < 1e-5
    v.close()


def test_fc7_bf16_operands(pkg, orc):
    """nvqa_vgg16_set_precision(1): every product of the extractor with both operands rounded to bf16 (f32 accumulate)
    against the oracle's same mode; the result sits at a bf16-sized distance from the f32 features and closer to the
    bf16 oracle than 0.75 x that distance; switching back restores the f32 path."""
    div, hw, n = 8, 64, 5
    vo = orc.VggOracle(div, hw)
    w = vo.synth_weights()
    x = np.random.default_rng(3).uniform(-110, 130, (n, 3, hw, hw)).astype(np.float32)
    exact = vo.fc7(w, x)
    vo.set_precision(1)
    try:
        ref = vo.fc7(w, x)
    finally:
        vo.set_precision(0)
    v = pkg.binding.Vgg16(0, div, hw, max_batch=n)
    v.set_weights(w)
    v.set_precision(1)
    got = v.fc7(x)
    scale = np.abs(ref).max()
    err, dist = np.abs(got - ref).max() / scale, np.abs(got - exact).max() / scale
    # 15 layers of roundings: a flipped rounding upstream moves everything downstream (tests/test_oracle_bf16_cascade.py),
    # so two correct bf16 implementations sit a fixed fraction of the rounding error apart -- measured 0.56 in max norm
    assert err < 1e-2 and dist > 1e-3 and err < 0.75 * dist, (err, dist)
    v.set_precision(0)
    assert np.abs(v.fc7(x) - exact).max() / scale < 1e-4
    v.close()


def test_fc7_bf16_gfx950_form(pkg, orc, monkeypatch):
    """The full-width network in bf16 mode runs the gfx950 form (every C_in from conv1_2 on a multiple of 64): bf16 images of
    the weights, bf16 NHWC activations written by the convolution epilogues, bf16 LDS images, v_mfma_f32_16x16x32_bf16, a
    bf16 max pool, fc6 from bf16 operands.  The operand values are the ones the oracle's bf16 mode rounds (rounding when a
    value is stored = rounding when it is read; max commutes with a monotone rounding), only the summation order differs:
    same yardstick as test_fc7_bf16_operands, which covers the BF = 1 form at width / 8.  Also: rows are independent of the
    batch, a ragged batch of 5 spans two 128-row tiles at conv5, and f32 is restored afterwards."""
    div, hw, n = 1, 64, 5
    vo = orc.VggOracle(div, hw)
    w = vo.synth_weights()
    x = np.random.default_rng(4).uniform(-110, 130, (n, 3, hw, hw)).astype(np.float32)
    exact = vo.fc7(w, x)
    vo.set_precision(1)
    try:
        ref = vo.fc7(w, x)
    finally:
        vo.set_precision(0)
    v = pkg.binding.Vgg16(0, div, hw, max_batch=n)
    v.set_precision(1)           # before the weights: the bf16 images are made by set_weights
    v.set_weights(w)
    got = v.fc7(x)
    scale = np.abs(ref).max()
    err, dist = np.abs(got - ref).max() / scale, np.abs(got - exact).max() / scale
    from util import record
    # the yardstick: the BF = 1 form (f32 in memory, rounded on the fly) of the same network on the same input
    monkeypatch.setenv("NVQA_VGG_BF16_FORM", "1")
    v.set_precision(1)
    old = v.fc7(x)
    monkeypatch.delenv("NVQA_VGG_BF16_FORM")
    v.set_precision(1)
    err_old, between = np.abs(old - ref).max() / scale, np.abs(got - old).max() / scale
    record("vgg_bf16_gfx950_form", {"err_vs_bf16_oracle": float(err), "dist_bf16_vs_f32": float(dist),
                                    "bf1_form_err_vs_bf16_oracle": float(err_old), "between_forms": float(between)})
    assert err < 1e-2 and dist > 1e-3 and err < dist and err < 1.5 * err_old and between < 1.5 * err_old, (err, dist, err_old, between)
    assert np.array_equal(v.fc7(x), got)
    assert np.abs(v.fc7(x[:2]) - got[:2]).max() / scale < 1e-5
    v.set_precision(0)
    assert np.abs(v.fc7(x) - exact).max() / scale < 1e-4
    v.set_precision(1)           # after the weights: the images are made by set_precision
    assert np.array_equal(v.fc7(x), got)
    v.close()


@pytest.mark.parametrize("H,W,S", [(480, 640, 224), (100, 150, 224), (100, 300, 224), (50, 70, 64)])
def test_preprocess_matches_loadim(pkg, orc, H, W, S):
    """loadim on the device (k_vgg_preprocess) against the oracle, BIT-EXACT: Torch's image.scale in both of its branches
    (area average when a dimension shrinks -- 480 x 640 -> 224, the real pipeline; linear interpolation when it grows --
    100 x 150 -> 224; one of each -- 100 x 300 -> 224), x255, BGR, mean (001_prepro_img_vgg.lua:47-71).  Both sides follow
    the published algorithm of the `image` rock with unfused float operations in the same order."""
    rng = np.random.default_rng(1)
    rgb = rng.uniform(0, 1, (2, 3, H, W)).astype(np.float32)
    v = pkg.binding.Vgg16(0, 16, S, max_batch=2)
    got = v.preprocess(rgb)
    ref = orc.VggOracle(16, S).preprocess(rgb, S)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    v.close()


def test_preprocess_identity_size(pkg):
    # identity-size input: pure x255, BGR swap, mean subtraction (001_prepro_img_vgg.lua:65-69)
    rng = np.random.default_rng(1)
    v = pkg.binding.Vgg16(0, 16, 64, max_batch=2)
    same = rng.uniform(0, 1, (1, 3, 64, 64)).astype(np.float32)
    out = v.preprocess(same)
    assert np.allclose(out[0, 0], same[0, 2] * 255 - 103.939, atol=1e-3)
    assert np.allclose(out[0, 2], same[0, 0] * 255 - 123.68, atol=1e-3)
    v.close()


def test_rejects_bad_calls(pkg):
    with pytest.raises(pkg.binding.NvqaError):
        pkg.binding.Vgg16(0, 3, 224, 1)      # width_div must divide 64
    v = pkg.binding.Vgg16(0, 16, 32, max_batch=1)
    with pytest.raises(pkg.binding.NvqaError):
        v.fc7(np.zeros((1, 3, 32, 32), np.float32))  # no weights yet
    v.close()
