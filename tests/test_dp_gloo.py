"""Data-parallel semantics on CPU, world_size 2, gloo: each rank runs the step on its own
shard, gradients are summed and divided by the world size, and only THEN clamped
(002_train_baseline.lua:329 is non-linear) -- this must equal one step on the global batch.
The oracle stands in for the device step here (test infrastructure); on the GPU the same
reduction is one RCCL all-reduce inside nvqa_rmsprop_update."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(arch=1, T=6, V=23, E=8, R=12, L=2, I=16, C=12, A=8)
    dl = orc.make_dims(B=4, **kw)
    dg = orc.make_dims(B=4 * world, **kw)
    params = orc.synth_params(dg)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    sl = slice(4 * rank, 4 * rank + 4)
    o = orc.Oracle(np.float64)
    mine = o.step(dl, params, tok[sl], lens[sl], img[sl], lab[sl], None)
    g = torch.from_numpy(mine["grads"] * 300.0)  # scaled so that the clamp is active
    loss = torch.tensor([mine["loss"]], dtype=torch.float64)
    dist.all_reduce(g)
    dist.all_reduce(loss)
    g /= world
    x = params.astype(np.float64).copy()
    m = np.zeros_like(x)
    gl = g.numpy().copy()
    o.rmsprop(x, gl, m, 3e-4, 0.99, 1e-8, 0.0, 10.0)
    if rank == 0:
        ref = o.step(dg, params, tok, lens, img, lab, None)
        xr = params.astype(np.float64).copy()
        mr = np.zeros_like(xr)
        gr = ref["grads"] * 300.0
        clipped = float((np.abs(gr) > 10).mean())
        o.rmsprop(xr, gr, mr, 3e-4, 0.99, 1e-8, 0.0, 10.0)
        q.put((float(loss[0]) / world, ref["loss"], float(np.abs(x - xr).max()),
               float(np.abs(g.numpy() - ref["grads"] * 300.0).max()), clipped))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mean_then_clamp_equals_global_batch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss_dp, loss_ref, dx, dg, clipped = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert abs(loss_dp - loss_ref) < 1e-12
    assert dg < 1e-10 and dx < 1e-12
    assert clipped > 0.001  # the clamp really was exercised
