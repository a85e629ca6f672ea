"""Why the full-size bf16 parity test (tests/test_gpu_bf16.py::test_bf16_full_size) cannot be tighter than a fraction of
the bf16-to-f32 distance: the oracle in operand-rounding mode run twice, once with f32 and once with f64 arithmetic between
the roundings.  Without rounding the two agree to 1.5e-7; with it, an activation that sits within 1e-7 of a bf16 rounding
boundary goes to different neighbours in the two runs, the flip (2^-8 relative) moves every downstream pre-activation of
that row by ~1e-5, which flips a few per cent of ITS roundings, and over 28 steps x 2 layers the two runs end a fixed
fraction of the rounding error itself apart.  CPU only (the oracle is the thing under study here)."""
import numpy as np


def _l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


def test_bf16_roundings_flip_and_cascade(orc):
    d = orc.make_dims(arch=2, B=128, T=26, V=2000, E=512, R=512, L=2, I=2048, C=4, A=1000)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=11, full_length=True)
    sc = {}
    for dt in (np.float32, np.float64):
        o = orc.Oracle(dt)
        sc[dt, 0] = o.step(d, params, tok, None, img, lab, None, train=False)["scores"]
        o.set_precision(1)
        try:
            sc[dt, 1] = o.step(d, params, tok, None, img, lab, None, train=False)["scores"]
        finally:
            o.set_precision(0)
    arith = _l2(sc[np.float32, 0], sc[np.float64, 0])       # f32 vs f64 arithmetic, no rounding
    dist = _l2(sc[np.float32, 1], sc[np.float64, 0])        # the bf16 rounding error
    cascade = _l2(sc[np.float32, 1], sc[np.float64, 1])     # the same bf16 algorithm, arithmetic 1e-7 apart
    assert arith < 1e-6
    assert 5e-4 < dist < 1e-2
    assert 0.1 * dist < cascade < 0.6 * dist, (cascade, dist)  # measured 0.28 - 0.30 at B = 512 (arch1 and arch2)
