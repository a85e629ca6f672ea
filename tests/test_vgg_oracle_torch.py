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


def _scale_matrix(src_len, dst_len):
    """Independent float64 statement of one 1-D pass of Torch's image.scale as a [dst x src] weight matrix: interpolation
    weights when enlarging (step (src - 1) / (dst - 1)), normalised interval-overlap weights (area average over
    [i s, (i + 1) s), s = src / dst) when shrinking."""
    Wm = np.zeros((dst_len, src_len))
    if dst_len > src_len:
        for i in range(dst_len):
            if i == dst_len - 1 or src_len == 1:
                Wm[i, src_len - 1 if i == dst_len - 1 else 0] = 1.0
                continue
            p = i * (src_len - 1) / (dst_len - 1)
            k = int(np.floor(p))
            Wm[i, k] += 1.0 - (p - k)
            Wm[i, k + 1] += p - k
    elif dst_len < src_len:
        sc = src_len / dst_len
        for i in range(dst_len):
            lo, hi = i * sc, (i + 1) * sc
            for k in range(int(np.floor(lo)), min(src_len, int(np.floor(hi)) + 1)):
                Wm[i, k] = max(0.0, min(k + 1, hi) - max(k, lo))
            Wm[i] /= Wm[i].sum()
    else:
        Wm = np.eye(src_len)
    return Wm


@pytest.mark.parametrize("H,W,S", [(480, 640, 224), (100, 150, 224), (100, 300, 224), (41, 57, 32), (32, 32, 32), (7, 300, 16)])
def test_loadim_resize_is_torch_image_scale(orc, H, W, S):
    """loadim's image.scale (001_prepro_img_vgg.lua:50): separable, rows first; linear interpolation when a dimension grows,
    AREA AVERAGE when it shrinks (the branch every real VQA image takes: 480 x 640 -> 224).  The C oracle against the
    weight-matrix statement above in float64; the `image` rock itself is absent (PARITY UNPINNED against Torch7)."""
    vo = orc.VggOracle(16, 32)
    rng = np.random.default_rng(H * 1000 + W)
    rgb = rng.uniform(0, 1, (2, 3, H, W)).astype(np.float32)
    got = vo.preprocess(rgb, S)                   # scale -> x255 -> BGR -> minus mean (001_prepro_img_vgg.lua:50,65-69)
    Wy, Wx = _scale_matrix(H, S), _scale_matrix(W, S)
    r = np.matmul(Wy, np.matmul(rgb.astype(np.float64), Wx.T)) * 255.0   # rows first, then columns
    want = r[:, [2, 1, 0]] - np.array([103.939, 116.779, 123.68]).reshape(1, 3, 1, 1)
    # float sample positions (di * scale in f32, as the source computes them) against f64 ones: 224 x 6e-8 of a pixel, times
    # neighbour differences of up to 255 -> a few 1e-3 on values of +-150
    assert np.abs(got - want).max() < 1e-2, float(np.abs(got - want).max())


def test_loadim_enlarging_is_align_corners_bilinear(orc):
    """Where both dimensions grow the separable linear pass is align-corners bilinear interpolation (torch.nn.functional)."""
    vo = orc.VggOracle(16, 32)
    rng = np.random.default_rng(2)
    rgb = rng.uniform(0, 1, (2, 3, 21, 27)).astype(np.float32)
    got = vo.preprocess(rgb, 32)
    r = F.interpolate(torch.tensor(rgb, dtype=torch.float64), size=(32, 32), mode="bilinear", align_corners=True) * 255.0
    mean = torch.tensor([103.939, 116.779, 123.68], dtype=torch.float64).view(1, 3, 1, 1)
    want = r[:, [2, 1, 0]] - mean
    assert np.abs(got - want.numpy()).max() < 2e-3
