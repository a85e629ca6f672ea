"""The library's data-parallel exchange with world > 1 on ONE GPU.

NVQA_RCCL_LIB points libnvqa at tests/shim/libnccl_shim.so, whose ncclAllReduce(sum) returns world x send: what
`world` ranks holding identical gradients produce.  Through the real nvqa_comm_init / nvqa_step* / nvqa_get_grads /
nvqa_rmsprop_update path (per-segment all-reduce on the communication stream, evSeg / evComm edges, 1/world in the
update) the gradients and the updated parameters must be BIT-IDENTICAL to a context without a communicator
(world x g x 1/world is exact for a power-of-two world): a slice reduced zero or two times, a missing event edge, or
a clamp before the mean would all show.  Reference anchor: 002_train_baseline.lua:323-329 (sum, then clamp)."""
import ctypes
import os

import numpy as np
import pytest

from util import gdims, gdrop

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim", "libnccl_shim.so")

CLAMP = 1e-4   # small enough to bite on > 1 % of the gradient entries of these tiny models
ARCH1 = dict(arch=1, B=24, T=9, V=61, E=20, R=32, I=48, C=40, A=16)
ARCH2 = dict(arch=2, B=24, T=7, V=61, E=32, R=32, I=48, C=4, A=16)


@pytest.fixture()
def shim_env(monkeypatch):
    assert os.path.exists(SHIM), "tests/shim/libnccl_shim.so missing: run __graft_entry__.build()"
    monkeypatch.setenv("NVQA_RCCL_LIB", SHIM)
    monkeypatch.setenv("NCCL_SHIM_DELAY_US", "300")  # every exchange outlasts the kernels around it
    yield ctypes.CDLL(SHIM)


def _run(pkg, orc, d, params, batch, world, steps=3, wd=0.0, scales=None):
    tok, lens, img, lab = batch
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    if scales is not None:
        ctx.set_grad_scales(scales)
    if world > 1:
        ctx.comm_init(0, world, ctx.comm_unique_id())
    out = []
    for it in range(steps):
        dr = orc.Dropout(1, 0.5, 123, it)
        loss = ctx.step(tok, lens if d.arch == 1 else None, img, lab, gdrop(pkg, dr))
        g = ctx.get_grads()                 # unclamped mean
        gc = ctx.get_grads(CLAMP)            # clamp on the way out acts on the mean
        ctx.rmsprop_update(3e-4, 0.99, 1e-8, wd, CLAMP)  # a clamp that bites: clamp-before-mean would differ
        out.append((loss, g, gc, ctx.get_params()))
    ctx.close()
    return out


@pytest.mark.parametrize("arch", [1, 2])
@pytest.mark.parametrize("L", [1, 2, 3])
def test_shim_world_equals_single_rank_bitwise(pkg, orc, shim_env, arch, L):
    kw = dict(ARCH1 if arch == 1 else ARCH2, L=L)
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    batch = orc.synth_batch(d, full_length=False)
    wd = 1e-4 if arch == 2 else 0.0
    base = _run(pkg, orc, d, params, batch, 1, wd=wd)
    assert float(np.mean(np.abs(base[0][1]) > CLAMP)) > 0.01, "the test clamp must bite"
    for world in (2, 4, 8):
        got = _run(pkg, orc, d, params, batch, world, wd=wd)
        for it, (a, b) in enumerate(zip(base, got)):
            assert a[0] == b[0], (world, it, "loss")
            assert np.array_equal(a[1], b[1]), (world, it, "mean gradient")
            assert np.array_equal(a[2], b[2]), (world, it, "clamped gradient")
            assert np.array_equal(a[3], b[3]), (world, it, "parameters after the update")


def test_shim_with_lr_scale_segments(pkg, orc, shim_env):
    """-lr_scale path: three k_rmsprop launches with per-segment scales x 1/world."""
    d = orc.make_dims(**dict(ARCH1, L=2))
    params = orc.synth_params(d)
    batch = orc.synth_batch(d, full_length=False)
    sc = np.array([0.5, 0.25, 1.0], np.float32)
    base = _run(pkg, orc, d, params, batch, 1, scales=sc)
    got = _run(pkg, orc, d, params, batch, 4, scales=sc)
    for a, b in zip(base, got):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])


def test_every_gradient_element_is_exchanged_exactly_once(pkg, orc, shim_env, monkeypatch):
    """world = 3 is not a power of two, so the identity is no longer exact, but the raw device sum must be 3 x g
    everywhere to rounding: checks slice coverage independently of the scale (zero or two exchanges give 1x / 9x)."""
    monkeypatch.setenv("NCCL_SHIM_DELAY_US", "0")
    for kw in (dict(ARCH1, L=2), dict(ARCH2, L=2)):
        d = orc.make_dims(**kw)
        params = orc.synth_params(d)
        batch = orc.synth_batch(d, full_length=False)
        base = _run(pkg, orc, d, params, batch, 1, steps=1)[0][1]
        got = _run(pkg, orc, d, params, batch, 3, steps=1)[0][1]   # = (3 g) / 3
        assert np.allclose(got, base, rtol=3e-7, atol=0), float(np.abs(got - base).max())
