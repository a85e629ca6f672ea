"""Generates the committed golden vectors from the CPU oracle (float64 arithmetic, stored as
float32/float64 arrays).  The reference (Lua/Torch7) cannot run in the build image and holds
no fixtures of its own, so these vectors pin the ORACLE, not Torch7:
"parity unpinned" with respect to the real reference.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (dims, full_length, dropout mode, store_full_tensors)
    "arch1_tiny": (dict(arch=1, B=4, T=5, V=11, E=8, R=8, L=2, I=12, C=12, A=8), False, 1, True),
    "arch2_tiny": (dict(arch=2, B=4, T=5, V=11, E=8, R=8, L=2, I=12, C=12, A=8), False, 1, True),
    # BASELINE.json configs[0]: arch1, batch 16, seq 26, 4096-d feats (V reduced to 1000 to keep
    # the oracle fast); inputs are regenerated from the seed, outputs/checksums are stored
    "arch1_mid": (dict(arch=1, B=16, T=26, V=1000, E=200, R=512, L=2, I=4096, C=1024, A=1000), False, 0, False),
}


def run_case(name):
    kw, full, mode, store_full = CASES[name]
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=full)
    dr = orc.Dropout(mode, 0.5, 123, 11)
    o = orc.Oracle(np.float64)
    tr = o.step(d, params, tok, lens, img, lab, dr)
    ev = o.step(d, params, tok, lens, img, lab, None, train=False)
    x = params.astype(np.float64).copy()
    g = tr["grads"].copy()
    m = np.zeros_like(x)
    o.rmsprop(x, g, m, 3e-4, 0.99, 1e-8, 1e-4 if kw["arch"] == 2 else 0.0, 10.0)
    lo = orc.layout(d)
    seg_sum = {k: float(tr["grads"][v[0]:v[0] + v[1]].sum()) for k, v in lo.items() if not k.startswith("_")}
    seg_abs = {k: float(np.abs(tr["grads"][v[0]:v[0] + v[1]]).sum()) for k, v in lo.items() if not k.startswith("_")}
    out = dict(dims=np.array([getattr(d, n) for n, _ in d._fields_], np.int32),
               dropout=np.array([mode, 123, 11], np.int64),
               loss=np.float64(tr["loss"]), eval_scores=ev["scores"], eval_argmax=ev["argmax"],
               train_scores=tr["scores"],
               grad_seg_names=np.array(sorted(seg_sum)), grad_seg_sum=np.array([seg_sum[k] for k in sorted(seg_sum)]),
               grad_seg_abs=np.array([seg_abs[k] for k in sorted(seg_abs)]),
               grad_head=tr["grads"][:64].copy(), grad_tail=tr["grads"][-64:].copy(),
               params_after_head=x[:64].copy(), params_after_tail=x[-64:].copy())
    if store_full:
        out.update(params=params, tokens=tok, lengths=lens, img=img, labels=lab, grads=tr["grads"],
                   params_after=x)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", tr["loss"], "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    for n in CASES:
        run_case(n)
