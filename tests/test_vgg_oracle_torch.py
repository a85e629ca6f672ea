"""The VGG-16 fc7 oracle (oracle/vgg_oracle.c, direct NCHW convolution) against PyTorch's library operators:
13 x (conv2d 3x3 pad 1 + ReLU), 5 x max_pool2d 2x2, fc6 + ReLU over the CHW-flattened pool5, fc7 + ReLU
(001_prepro_img_vgg.lua:101-113, tap = module 38), weights in the flat Caffe order; and the loadim resize
against F.interpolate(bilinear, align_corners=True) -- the interpolation the oracle assumes for image.scale.
Third-party code as the stand-in for the cuDNN / Torch path that cannot run here."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import relmax

CH = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]
POOL_AFTER = {1, 3, 6, 9, 12}


@pytest.mark.parametrize("div,hw,n", [(16, 64, 2), (8, 32, 3)])
def test_vgg_oracle_matches_torch_functional(orc, div, hw, n):
    vo = orc.VggOracle(div, hw)
    w = vo.synth_weights(7)
    rng = np.random.default_rng(1)
    x = rng.uniform(-120, 130, (n, 3, hw, hw)).astype(np.float32)
    got = vo.fc7(w, x)
    t = torch.tensor(x, dtype=torch.float64)
    off, cin = 0, 3
    for i, c in enumerate(CH):
        co = max(1, c // div)
        wt = torch.tensor(w[off:off + co * cin * 9].reshape(co, cin, 3, 3), dtype=torch.float64)
        off += co * cin * 9
        b = torch.tensor(w[off:off + co], dtype=torch.float64)
        off += co
        t = F.relu(F.conv2d(t, wt, b, padding=1))
        if i in POOL_AFTER:
            t = F.max_pool2d(t, 2, 2)
        cin = co
    t = t.reshape(n, -1)
    for k in (t.shape[1], vo.feature_dim):
        Wm = torch.tensor(w[off:off + vo.feature_dim * k].reshape(vo.feature_dim, k), dtype=torch.float64)
        off += vo.feature_dim * k
        b = torch.tensor(w[off:off + vo.feature_dim], dtype=torch.float64)
        off += vo.feature_dim
        t = F.relu(F.linear(t, Wm, b))
    assert off == w.size
    assert relmax(got, t.numpy()) < 2e-5          # f32 direct convolution against f64


def test_loadim_resize_is_align_corners_bilinear(orc):
    vo = orc.VggOracle(16, 32)
    rng = np.random.default_rng(2)
    rgb = rng.uniform(0, 1, (2, 3, 41, 57)).astype(np.float32)
    got = vo.preprocess(rgb, 32)                  # scale -> x255 -> BGR -> minus mean (001_prepro_img_vgg.lua:50,65-69)
    r = F.interpolate(torch.tensor(rgb, dtype=torch.float64), size=(32, 32), mode="bilinear", align_corners=True) * 255.0
    mean = torch.tensor([103.939, 116.779, 123.68], dtype=torch.float64).view(1, 3, 1, 1)
    want = r[:, [2, 1, 0]] - mean
    assert np.abs(got - want.numpy()).max() < 2e-3
