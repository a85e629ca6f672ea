# which rows of a B = 500 forward pass differ between the two persistent forward kernels (NVQA_FWD_KERNEL)
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if len(sys.argv) > 1:
    import __graft_entry__ as ge
    pkg = ge.load_package()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(ge.__file__)), "tests"))
    from oracle import oracle as orc
    orc.build()
    d = orc.make_dims(arch=1, B=500, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=123, full_length=True, min_len=3)
    from util import gdims
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    scores, _ = ctx.forward(tok, lens, img)
    np.save(sys.argv[1], np.asarray(scores))
    s2, _ = ctx.forward(tok, lens, img)
    print(sys.argv[1], "repeat identical:", np.array_equal(scores, s2), ctx.persistent_state())
    sys.exit(0)
out = {}
for k in ("1", "3"):
    f = f"/tmp/b500_{k}.npy"
    subprocess.run([sys.executable, __file__, f], env=dict(os.environ, NVQA_FWD_KERNEL=k), check=True)
    out[k] = np.load(f)
diff = np.abs(out["1"] - out["3"]).max(axis=1)
bad = np.nonzero(diff > 1e-3)[0]
print("rows differing > 1e-3:", len(bad), bad[:64], diff[bad][:16])
