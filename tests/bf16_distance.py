"""(script, not a test: python tests/bf16_distance.py on the GPU box) Calibration of the full-size bf16 parity tolerances (tests/test_gpu_bf16.py): distance of the HIP bf16 step to the
oracle in operand-rounding mode and to the f32 result, max-norm and L2, per parameter segment, at T = 26 and at 2-3 steps."""
import os, sys, numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
os.environ["NVQA_PERSIST"] = "1"

import conftest
from util import gdims, gdrop
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc
orc.build()
def l2(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-300))
def mx(a, b): return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-300))
for name, kw, ulen in [("arch1_T26", dict(arch=1, B=512, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000), None),
                 ("arch1_len3", dict(arch=1, B=512, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000), 3),
                 ("arch2_T26", dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000), None),
                 ("arch2_len2", dict(arch=2, B=512, T=26, V=14773, E=512, R=512, L=2, I=2048, C=4, A=1000), 2)]:
    d = orc.make_dims(**kw)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=11, full_length=True)
    if ulen is not None:
        if d.arch == 1:
            lens[:] = ulen
            left = np.zeros_like(tok); left[:, :ulen] = np.random.default_rng(5).integers(1, d.V + 1, (d.B, ulen))
            tok = orc.right_align(left, lens)
        else:
            tok[:, ulen:] = 0
    lens_ = lens if d.arch == 1 else None
    odr = orc.Dropout(1, 0.5, 123, 4)
    o = orc.Oracle(np.float32)
    exact = o.step(d, params, tok, lens_, img, lab, odr)
    exs = o.step(d, params, tok, lens_, img, lab, None, train=False)["scores"]
    o.set_precision(1)
    ref = o.step(d, params, tok, lens_, img, lab, odr)
    ev = o.step(d, params, tok, lens_, img, lab, None, train=False)
    o.set_precision(0)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params); ctx.set_precision(1)
    loss = ctx.step(tok, lens_, img, lab, gdrop(pkg, odr))
    g = ctx.get_grads()
    sc, _ = ctx.forward(tok, lens_, img)
    print(name, "scores: max", mx(sc, ev["scores"]), "dist", mx(sc, exs), "| l2", l2(sc, ev["scores"]), "dist", l2(sc, exs))
    lo = orc.layout(d)
    for k, v in lo.items():
        if k.startswith("_"): continue
        a, n = v
        print("   %-8s max %.5f / %.5f   l2 %.5f / %.5f" % (k, mx(g[a:a+n], ref["grads"][a:a+n]), mx(g[a:a+n], exact["grads"][a:a+n]),
              l2(g[a:a+n], ref["grads"][a:a+n]), l2(g[a:a+n], exact["grads"][a:a+n])))
    ctx.close()
