"""Data-parallel plumbing.

* The real librccl with world = 1: the communicator is created through the C ABI (dlopen of librccl, ncclUniqueId passed by
  value, ncclAllReduce on the library's stream) and the update must equal the single-process update bit for bit.
* TWO PROCESSES, two ranks with DIFFERENT half-batches, on the one GPU of the box: librccl refuses two ranks on one device,
  so the ranks exchange through the stand-in's shared-memory mode (tests/shim, NCCL_SHIM_SHM: a real stream-ordered
  all-reduce between processes).  Everything else is the library's own data-parallel path -- per-segment reductions on the
  communication stream, event edges, 1/world and the clamp in k_rmsprop, the status word -- and the result must be the
  oracle's GLOBAL-batch step: mean over ranks, THEN clamp (002_train_baseline.lua:323-329).
* The same two ranks on two devices through librccl itself (skipped on a one-GPU box)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from util import gdims

pytestmark = pytest.mark.gpu
KW = dict(arch=1, B=8, T=6, V=40, E=12, R=16, L=2, I=32, C=24, A=12)


def _world1_identity():
    """body of test_world1_allreduce_is_identity, in a process of its own"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    from oracle import oracle as orc
    pkg = ge.load_package()
    d = orc.make_dims(**KW)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    outs = []
    for use_comm in (False, True):
        ctx = pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), 0)
        ctx.set_params(params)
        if use_comm:
            ctx.comm_init(0, 1, ctx.comm_unique_id())
        ctx.step(tok, lens, img, lab, None)
        ctx.rmsprop_update(3e-4)
        outs.append(ctx.get_params())
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


def test_world1_allreduce_is_identity():
    """The real librccl, world = 1, through the C ABI, IN the test runner's process.  (Round 3 ran this in a process of its own:
    with librccl resident and child processes started afterwards the interpreter aborted at exit inside glibc, "double free
    or corruption".  The cause was the loader: librccl opened RTLD_GLOBAL and a second librccl image -- the torch wheel's,
    mapped by `import torch` in test_two_ranks_on_two_devices -- interposing its global C++ objects.  The library now opens
    it RTLD_LOCAL, tests/test_comm_loader.py holds both import orders on the CPU, and this test is back in-process.)"""
    _world1_identity()


CLAMP = 2e-3   # bites on the largest gradient entries of this model: clamp-before-mean would differ from clamp-after-mean


def _rank(rank, world, idq, resq, env=None, one_device=False, steps=1, clamp=10.0):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(env or {})
    try:
        import __graft_entry__ as ge
        from oracle import oracle as orc
        pkg = ge.load_package()
        kw = dict(KW)
        d = orc.make_dims(**kw)
        dg = orc.make_dims(**{**kw, "B": kw["B"] * world})
        params = orc.synth_params(d)
        tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
        sl = slice(rank * d.B, (rank + 1) * d.B)
        ctx = pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), 0 if one_device else rank)
        ctx.set_params(params)
        if rank == 0:
            cid = ctx.comm_unique_id()
            for _ in range(world - 1):
                idq.put(cid)
        else:
            cid = idq.get(timeout=60)
        ctx.comm_init(rank, world, cid)
        grads = None
        for it in range(steps):
            ctx.step(tok[sl], lens[sl], img[sl], lab[sl], None)
            grads = ctx.get_grads()                       # the mean over the ranks
            ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, clamp)
        resq.put((rank, "ok", (ctx.get_params(), grads)))
        ctx.close()
    except Exception as e:  # noqa: BLE001
        resq.put((rank, "err", repr(e)))


def test_two_ranks_on_two_devices(pkg, orc):
    import torch
    if torch.cuda.device_count() < 2:  # counting devices does not initialise the GPU in this process
        pytest.skip("needs 2 GPUs: RCCL refuses two ranks on one device ('invalid usage')")
    ctx = mp.get_context("spawn")
    idq, resq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, idq, resq)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r, st, val = resq.get(timeout=120)
            res[r] = (st, val)
    except Exception:
        for p in procs:
            p.kill()
        pytest.fail("2-rank RCCL run did not complete")
    for p in procs:
        p.join(30)
    assert all(st == "ok" for st, _ in res.values()), [v for s, v in res.values() if s != "ok"]
    _check_against_global_batch(orc, res, steps=1, clamp=10.0)


def _check_against_global_batch(orc, res, steps, clamp):
    # both ranks hold the same parameters and the same mean gradient, bit for bit ...
    assert np.array_equal(res[0][1][0], res[1][1][0]) and np.array_equal(res[0][1][1], res[1][1][1])
    # ... equal to the oracle's step on the GLOBAL batch (the mean of the ranks' means), clamped after the mean
    d = orc.make_dims(**KW)
    dg = orc.make_dims(**{**KW, "B": KW["B"] * 2})
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    o = orc.Oracle(np.float32)
    x, m = params.copy(), np.zeros_like(params)
    g = None
    for it in range(steps):
        g = o.step(dg, x, tok, lens, img, lab, None)["grads"].copy()
        o.rmsprop(x, g.copy(), m, 3e-4, 0.99, 1e-8, 0.0, clamp)   # (the oracle's update clamps its gradient argument in place)
    assert np.abs(res[0][1][1] - g).max() <= 2e-6 * np.abs(g).max()
    assert np.abs(res[0][1][0] - x).max() < 2e-6
    return g


def test_two_ranks_one_gpu_real_exchange(pkg, orc, tmp_path):
    """Two processes = two ranks on device 0, each with its own half of a global batch, exchanging through the stand-in's
    shared-memory all-reduce.  Three steps with a clamp that bites: parameters and mean gradients identical on both ranks
    and equal to the oracle's global-batch trajectory (clamp AFTER the mean: a rank-local clamp would not commute)."""
    import subprocess
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim", "libnccl_shim.so")
    assert os.path.exists(shim), "tests/shim/libnccl_shim.so missing: run __graft_entry__.build()"
    env = dict(os.environ, NVQA_RCCL_LIB=shim, NCCL_SHIM_SHM=f"/nvqa_dp_{os.getpid()}", NCCL_SHIM_SHM_MB="8",
               NCCL_SHIM_DELAY_US="0", NCCL_SHIM_CUS="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", str(tmp_path), "3", repr(CLAMP)], env=env)
             for r in range(2)]
    try:
        rcs = [p.wait(timeout=300) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("2-rank shared-memory run did not complete")
    assert rcs == [0, 0], rcs
    res = {}
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        res[r] = ("ok", (z["params"], z["grads"]))
    g = _check_against_global_batch(orc, res, steps=3, clamp=CLAMP)
    assert float(np.mean(np.abs(g) > CLAMP)) > 0.001, "the test clamp must bite"


KW_P = dict(arch=1, B=64, T=6, V=60, E=200, R=512, L=2, I=64, C=32, A=12)   # a shape the persistent kernels take


def test_two_ranks_timeout_on_one_rank_keeps_replicas_identical(pkg, orc, tmp_path):
    """the persistent kernel of ONE rank gives up in step 1 (NVQA_PF_SPIN = 1 for that rank's first step only),
    the trainer loop runs four steps without ever asking for the loss, so no host looks at its status in between.
    Required: step 1 is skipped on BOTH ranks (the exchanged status word), steps 2-4 are applied on BOTH ranks -- round 3
    consulted the rank-local sticky record too, which stays set until that rank's host synchronises, so the rank that timed
    out kept skipping while the other applied: replicas diverged silently -- the next synchronisation reports the failure
    on BOTH ranks, and the parameters are identical and equal to the oracle's global-batch trajectory of the three good
    steps."""
    import subprocess
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim", "libnccl_shim.so")
    base = dict(os.environ, NVQA_RCCL_LIB=shim, NCCL_SHIM_SHM=f"/nvqa_dpt_{os.getpid()}", NCCL_SHIM_SHM_MB="16",
                NCCL_SHIM_DELAY_US="0", NCCL_SHIM_CUS="0")
    envs = [dict(base), dict(base, NVQA_PF_SPIN="1", NVQA_PF_SPIN_STEPS="1")]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "timeout", str(r), "2", str(tmp_path)], env=envs[r])
             for r in range(2)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("2-rank time-out run did not complete")
    assert rcs == [0, 0], rcs
    z = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(2)]
    for r in range(2):
        assert bool(z[r]["reported"]), f"rank {r} never reported the failed step"
        assert not np.array_equal(z[r]["params"], z[r]["p0"]), f"rank {r} applied nothing"
    assert np.array_equal(z[0]["params"], z[1]["params"]), "replicas diverged after a one-rank time-out"
    assert np.array_equal(z[0]["params_after"], z[1]["params_after"])
    # three applied steps of the global batch (the skipped step leaves parameters AND the RMSprop state untouched)
    d = orc.make_dims(**KW_P)
    dg = orc.make_dims(**{**KW_P, "B": KW_P["B"] * 2})
    x = orc.synth_params(d)
    m = np.zeros_like(x)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    o = orc.Oracle(np.float32)
    for it in range(3):
        g = o.step(dg, x, tok, lens, img, lab, None)["grads"].copy()
        o.rmsprop(x, g, m, 3e-4, 0.99, 1e-8, 0.0, 10.0)
    assert np.abs(z[0]["params"] - x).max() < 2e-5, np.abs(z[0]["params"] - x).max()


def _main_rank_timeout(rank, world, out_dir):
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    from oracle import oracle as orc
    pkg = ge.load_package()
    d = orc.make_dims(**KW_P)
    dg = orc.make_dims(**{**KW_P, "B": KW_P["B"] * world})
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    sl = slice(rank * d.B, (rank + 1) * d.B)
    ctx = pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), 0)
    assert ctx.persistent_state() == {"fwd": True, "bwd": True}
    ctx.set_params(params)
    p0 = ctx.get_params()
    idf = os.path.join(out_dir, "comm_id.bin")
    if rank == 0:
        cid = ctx.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(cid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 120, "rank 0 never published the communicator id"
            time.sleep(0.05)
        cid = open(idf, "rb").read()
    ctx.comm_init(rank, world, cid)
    for it in range(4):                                  # an asynchronous trainer loop: nobody asks for the loss
        ctx.step(tok[sl], lens[sl], img[sl], lab[sl], None, want_loss=False)
        ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, 10.0)
    reported = False
    try:
        ctx.sync()
    except pkg.binding.NvqaError as e:
        reported = "timed out" in str(e)
    params_now = ctx.get_params()
    # both ranks left the persistent path together; one more (per-level) step keeps them identical
    ctx.step(tok[sl], lens[sl], img[sl], lab[sl], None, want_loss=False)
    ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, 10.0)
    ctx.sync()
    assert ctx.persistent_state() == {"fwd": False, "bwd": False}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=params_now, p0=p0, reported=reported, params_after=ctx.get_params())
    ctx.close()


def _main_rank(rank, world, out_dir, steps, clamp):
    """one rank of the test above as a process of its own (the communicator id travels through a file)"""
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    from oracle import oracle as orc
    pkg = ge.load_package()
    d = orc.make_dims(**KW)
    dg = orc.make_dims(**{**KW, "B": KW["B"] * world})
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    sl = slice(rank * d.B, (rank + 1) * d.B)
    ctx = pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), 0)
    ctx.set_params(params)
    idf = os.path.join(out_dir, "comm_id.bin")
    if rank == 0:
        cid = ctx.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(cid)
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 120, "rank 0 never published the communicator id"
            time.sleep(0.05)
        cid = open(idf, "rb").read()
    ctx.comm_init(rank, world, cid)
    grads = None
    for it in range(steps):
        ctx.step(tok[sl], lens[sl], img[sl], lab[sl], None)
        grads = ctx.get_grads()
        ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, clamp)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=ctx.get_params(), grads=grads)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "world1":
        _world1_identity()
        sys.exit(0)
    if sys.argv[1] == "timeout":
        _main_rank_timeout(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    _main_rank(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), float(sys.argv[5]))
