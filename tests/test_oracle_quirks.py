"""The oracle's opt-in reference quirks for arch2 (oracle/nvqa_oracle.c, above oracle_arch2_step) against an
independent statement (tests/ref_autograd.py: autograd + the single correction the Lua aliasing implies):
Q1  aliased top-layer h0   misc/Encoder_lstm.lua:238-239 with :30-47,:164
Q11 untrained lookup table misc/Encoder_lstm.lua:49-58 with 002_train_baseline.lua:186,273
PARITY UNPINNED with respect to Torch7 itself (the reference cannot run here); both sides are readings of the Lua."""
import numpy as np

import ref_autograd as ra

KW = dict(arch=2, B=5, T=5, V=13, E=8, R=12, I=16, C=4, A=7)


def _iterate(orc, L, quirks, iters=3, scale=1.0):
    d = orc.make_dims(L=L, **KW)
    lo = orc.layout(d)
    params = orc.synth_params(d).astype(np.float64) * scale
    tok, _, img, lab = orc.synth_batch(d, full_length=False)
    o = orc.Oracle(np.float64)
    o.set_ref_quirks(quirks)
    alias = ra.AliasedH0(d.B, d.R) if quirks & 1 else None
    x = params.copy()
    out = []
    try:
        for it in range(iters):
            dr = orc.Dropout(1, 0.5, 123, it)
            got = o.step(d, x, tok, None, img, lab, dr)
            ref = ra.arch2(d, lo, x, tok, img, lab, dr, aliased_h0=alias, no_lookup_grad=bool(quirks & 2))
            out.append((got, ref))
            x = x - 0.05 * got["grads"]          # any update: the iterations must differ
        ev = o.step(d, x, tok, None, img, lab, None, train=False)   # validate(): evaluate mode reads the same h0
        evr = ra.arch2(d, lo, x, tok, img, lab, None, train=False, aliased_h0=alias)
    finally:
        o.set_ref_quirks(0)
    return d, lo, out, (ev, evr)


def test_q1_aliased_h0_matches_independent_statement():
    for L in (1, 2):
        d, lo, out, (ev, evr) = _iterate(__import__("oracle.oracle", fromlist=["x"]), L, 1, scale=4.0)
        for it, (got, ref) in enumerate(out):
            assert abs(got["loss"] - ref["loss"]) < 1e-12, (L, it)
            assert np.abs(got["grads"] - ref["grads"]).max() < 1e-12, (L, it)
        assert np.abs(ev["scores"] - evr["scores"]).max() < 1e-12


def test_q1_changes_later_iterations_only():
    orc = __import__("oracle.oracle", fromlist=["x"])
    _, lo, plain, _ = _iterate(orc, 2, 0, scale=4.0)
    _, _, quirk, _ = _iterate(orc, 2, 1, scale=4.0)
    o, n = lo["w_h2h1"]
    # iteration 1: forward identical (h0 = 0 both ways); only the top layer's dW_h2h differs (step-1 term)
    assert plain[0][0]["loss"] == quirk[0][0]["loss"]
    diff = np.abs(plain[0][0]["grads"] - quirk[0][0]["grads"])
    assert diff[o:o + n].max() > 0 and np.delete(diff, np.s_[o:o + n]).max() == 0
    # iteration 2 on: the forward pass itself starts from a non-zero h0
    assert plain[1][0]["loss"] != quirk[1][0]["loss"]


def test_q11_lookup_gradient_is_zero_and_nothing_else_changes():
    orc = __import__("oracle.oracle", fromlist=["x"])
    _, lo, plain, _ = _iterate(orc, 2, 0, iters=1)
    _, _, quirk, _ = _iterate(orc, 2, 2, iters=1)
    o, n = lo["w_lk"]
    g0, g1 = plain[0][0]["grads"], quirk[0][0]["grads"]
    assert np.abs(g0[o:o + n]).max() > 0 and np.all(g1[o:o + n] == 0)
    assert np.array_equal(np.delete(g0, np.s_[o:o + n]), np.delete(g1, np.s_[o:o + n]))
    assert np.abs(quirk[0][0]["grads"] - quirk[0][1]["grads"]).max() < 1e-12
