# which rows of a ragged E = 200 forward pass differ between the two persistent forward kernels, under each debugging switch of the
# direct-operand kernel (run by tests/test_gpu_fwd3.py; by hand: python tests/dbg_rag_rows.py)
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
# reference: round 2's ring kernel; then the direct-operand kernel under each debugging switch (csrc/lstm_persist_fwd3.h)
VARIANTS = [("ring", {"NVQA_FWD_KERNEL": "1"}),
            ("fwd3_rag_instance", {"NVQA_FWD_KERNEL": "3", "NVQA_FWD3_RAGGED": "2"}),
            ("fwd3_instance_without_skips", {"NVQA_FWD_KERNEL": "3", "NVQA_FWD3_RAGGED": "1"}),
            ("fwd3_rag_no_skips_at_run_time", {"NVQA_FWD_KERNEL": "3", "NVQA_FWD3_RAGGED": "2", "NVQA_PF_DBG": "1024"}),
            ("fwd3_rag_signal_now", {"NVQA_FWD_KERNEL": "3", "NVQA_FWD3_RAGGED": "2", "NVQA_PF_DBG": "512"}),
            ("fwd3_rag_slow_layer0", {"NVQA_FWD_KERNEL": "3", "NVQA_FWD3_RAGGED": "2", "NVQA_PF_DBG": "256"})]
import collections
out = {}
for name, env in VARIANTS:
    f = f"/tmp/rag_{name}.npy"
    r = subprocess.run([sys.executable, __file__, f], env=dict(os.environ, **env), capture_output=True, text=True)
    print("==", name, env, "rc", r.returncode, r.stdout.strip().splitlines()[-1:] , r.stderr.strip()[-300:] if r.returncode else "", flush=True)
    if r.returncode: continue
    out[name] = np.load(f)
    if name == "ring": lens = np.load(f + ".lens.npy"); continue
    diff = np.abs(out["ring"] - out[name]).max(axis=1)
    order = np.argsort(-lens, kind="stable")           # sorted position of each row (longest first)
    pos = np.empty_like(order); pos[order] = np.arange(len(order))
    bad = np.nonzero(diff > 1e-4)[0]
    print("   rows differing > 1e-4:", len(bad), "of", len(diff), "max diff %.3g" % float(diff.max()))
    print("   by tile:", sorted(collections.Counter(int(pos[b]) // 4 // 16 for b in bad).items()), " by row mod 4:", sorted(collections.Counter((int(pos[b]) // 4) % 4 for b in bad).items()), flush=True)
