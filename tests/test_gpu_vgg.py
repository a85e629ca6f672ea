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


@pytest.mark.parametrize("div,hw,n", [(16, 32, 3), (16, 64, 2), (8, 96, 2), (1, 64, 2), (1, 224, 1)])
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


def test_preprocess_matches_loadim(pkg, orc):
    rng = np.random.default_rng(1)
    rgb = rng.uniform(0, 1, (2, 3, 50, 70)).astype(np.float32)
    v = pkg.binding.Vgg16(0, 16, 64, max_batch=2)
    got = v.preprocess(rgb)
    ref = orc.VggOracle(16, 64).preprocess(rgb, 64)
    assert np.abs(got - ref).max() < 1e-3
    # identity-size input: pure x255, BGR swap, mean subtraction (001_prepro_img_vgg.lua:65-69)
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
