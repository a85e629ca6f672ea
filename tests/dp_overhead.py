"""(script, not a test: python tests/dp_overhead.py on the GPU box) What the data-parallel plumbing costs a step on ONE GPU:
the headline workload with and without a communicator whose all-reduce is the stand-in of tests/shim (recv = world x
send, no delay) -- streams, events, the 1/world scale; the wire time of a real exchange is not in it."""
import os, sys, time
import numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
os.environ["NVQA_RCCL_LIB"] = os.path.join(os.getcwd(), "tests", "shim", "libnccl_shim.so")
os.environ["NCCL_SHIM_DELAY_US"] = "0"
import bench, __graft_entry__ as ge
pkg = ge.load_package()
w = bench.WORKLOAD
for world in (1, 2, 8):
    dims = pkg.binding.Dims(*[w[k] for k in ("arch", "B", "T", "V", "E", "R", "L", "I", "C", "A")])
    tr = pkg.trainer.VQATrainer(dims, device=0, seed=123)
    tr.init_params()
    q, lens, img_pos, ans, feats = bench.synth_dataset(w, 123)
    tr.load_dataset(q, lens, img_pos, ans, feats, img_norm=True)
    if world > 1:
        tr.ctx.comm_init(0, world, tr.ctx.comm_unique_id())
    def one():
        tr.ctx.step_indices(tr.next_batch(), tr._dropout(), want_loss=False)
        tr.rmsprop()
    for _ in range(5):
        one()
    tr.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        one()
    tr.ctx.sync()
    print("world", world, "ms/step %.4f" % ((time.perf_counter() - t0) / 20 * 1e3))
    tr.close()
