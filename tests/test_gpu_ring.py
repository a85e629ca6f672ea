"""The opt-in LDS-DMA ring level kernels (csrc/gemm_ring.h, NVQA_RING=1) against the oracle and against the
default register-staged kernels, on a ragged batch (inactive row tiles, two-segment K, split-K BPTT levels)."""
import os

import numpy as np
import pytest

from util import gdims, gdrop, relmax, segment_errors

pytestmark = pytest.mark.gpu


def _run(pkg, d, params, batch, dr, ring):
    old = os.environ.get("NVQA_RING")
    os.environ["NVQA_RING"] = "1" if ring else "0"
    try:
        ctx = pkg.binding.Context(gdims(pkg, d), 0)   # the switch is read at nvqa_create
    finally:
        if old is None:
            del os.environ["NVQA_RING"]
        else:
            os.environ["NVQA_RING"] = old
    ctx.set_params(params)
    loss = ctx.step(*batch, dr)
    grads = ctx.get_grads()
    ctx.close()
    return loss, grads


@pytest.mark.parametrize("kw", [
    dict(arch=1, B=96, T=7, V=50, E=32, R=64, L=2, I=32, C=48, A=12),   # E, R multiples of 32: the ring kernel takes every level
    dict(arch=1, B=70, T=5, V=50, E=16, R=32, L=3, I=32, C=16, A=12),   # E = 16: forward levels fall back (K % 32), BPTT levels on the ring
    dict(arch=2, B=80, T=6, V=50, E=64, R=64, L=2, I=32, C=8, A=12),
])
def test_ring_levels_match_oracle_and_default_path(pkg, orc, kw):
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    odr = orc.Dropout(1, 0.5, 123, 5)
    batch = (tok, lens if d.arch == 1 else None, img, lab)
    ref = orc.Oracle(np.float64).step(d, params, tok, lens if d.arch == 1 else None, img, lab, odr)
    l1, g1 = _run(pkg, d, params, batch, gdrop(pkg, odr), ring=True)
    l0, g0 = _run(pkg, d, params, batch, gdrop(pkg, odr), ring=False)
    assert abs(l1 - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert max(segment_errors(orc, d, g1, ref["grads"]).values()) < 1e-3
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and relmax(g1, g0) < 1e-4
