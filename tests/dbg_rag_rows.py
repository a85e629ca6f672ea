# debugging aid (not a test): which rows of a ragged E = 200 forward pass differ between the two persistent forward kernels
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import oracle as orc
    orc.build()
    from util import gdims
    d = orc.make_dims(arch=1, B=512, T=26, V=14773, E=200, R=512, L=2, I=4096, C=1024, A=1000)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, seed=123, full_length=False, min_len=3)
    ctx = pkg.binding.Context(gdims(pkg, d), 0)
    ctx.set_params(params)
    scores, _ = ctx.forward(tok, lens, img)
    s2, _ = ctx.forward(tok, lens, img)
    np.save(sys.argv[1], np.asarray(scores)); np.save(sys.argv[1] + ".lens.npy", np.asarray(lens))
    print(sys.argv[1], "repeat identical:", np.array_equal(scores, s2), "max repeat diff", float(np.abs(np.asarray(scores) - np.asarray(s2)).max()))
    sys.exit(0)
out = {}
for k in ("1", "3"):
    f = f"/tmp/rag_{k}.npy"
    subprocess.run([sys.executable, __file__, f], env=dict(os.environ, NVQA_FWD_KERNEL=k, NVQA_FWD3_ALL="1"), check=True)
    out[k] = np.load(f)
lens = np.load("/tmp/rag_1.npy.lens.npy")
diff = np.abs(out["1"] - out["3"]).max(axis=1)
order = np.argsort(-lens, kind="stable")           # sorted position of each row (longest first)
pos = np.empty_like(order); pos[order] = np.arange(len(order))
bad = np.nonzero(diff > 1e-4)[0]
print("rows differing > 1e-4:", len(bad), "of", len(diff), "max diff", float(diff.max()))
for b in bad[:40]:
    print("row", int(b), "len", int(lens[b]), "sorted pos", int(pos[b]), "block", int(pos[b]) % 4, "local", int(pos[b]) // 4, "tile", int(pos[b]) // 4 // 16, "diff %.3g" % diff[b])
import collections
print("by length:", sorted(collections.Counter(int(lens[b]) for b in bad).items()))
print("by tile:", sorted(collections.Counter(int(pos[b]) // 4 // 16 for b in bad).items()))
