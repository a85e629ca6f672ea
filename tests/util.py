"""Shared helpers for the parity tests."""
import numpy as np


def gdims(pkg, d):
    return pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_])


def gdrop(pkg, dr):
    return pkg.binding.Dropout(dr.mode, dr.p, dr.seed, dr.step)


def relmax(a, b):
    """max |a-b| / max |b|  (the scale-relative error used for tensors)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def segment_errors(orc, d, got, ref):
    lo = orc.layout(d)
    out = {}
    for k, v in lo.items():
        if k.startswith("_"):
            continue
        o, n = v
        out[k] = relmax(got[o:o + n], ref[o:o + n])
    return out
