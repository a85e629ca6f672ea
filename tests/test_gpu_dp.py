"""Data-parallel plumbing with the real librccl: the communicator is created through the C ABI (dlopen of librccl,
ncclUniqueId passed by value, ncclAllReduce on the library's stream) with world = 1, and the update must equal the
single-process update bit for bit.  The 2-rank leg needs two devices (rank r runs on device r) and is skipped on a
one-GPU box; world > 1 through the library's own exchange code is covered on one GPU by tests/test_gpu_dp_shim.py."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from util import gdims

pytestmark = pytest.mark.gpu
KW = dict(arch=1, B=8, T=6, V=40, E=12, R=16, L=2, I=32, C=24, A=12)


def test_world1_allreduce_is_identity(pkg, orc):
    d = orc.make_dims(**KW)
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    outs = []
    for use_comm in (False, True):
        ctx = pkg.binding.Context(gdims(pkg, d), 0)
        ctx.set_params(params)
        if use_comm:
            ctx.comm_init(0, 1, ctx.comm_unique_id())
        ctx.step(tok, lens, img, lab, None)
        ctx.rmsprop_update(3e-4)
        outs.append(ctx.get_params())
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


def _rank(rank, world, idq, resq):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
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
        ctx = pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), rank)  # one device per rank
        ctx.set_params(params)
        if rank == 0:
            cid = ctx.comm_unique_id()
            for _ in range(world - 1):
                idq.put(cid)
        else:
            cid = idq.get(timeout=60)
        ctx.comm_init(rank, world, cid)
        ctx.step(tok[sl], lens[sl], img[sl], lab[sl], None)
        ctx.rmsprop_update(3e-4, 0.99, 1e-8, 0.0, 10.0)
        resq.put((rank, "ok", ctx.get_params()))
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
    # both ranks hold the same parameters, equal to the oracle's global-batch update
    assert np.array_equal(res[0][1], res[1][1])
    d = orc.make_dims(**KW)
    dg = orc.make_dims(**{**KW, "B": KW["B"] * 2})
    params = orc.synth_params(d)
    tok, lens, img, lab = orc.synth_batch(dg, full_length=False)
    o = orc.Oracle(np.float32)
    g = o.step(dg, params, tok, lens, img, lab, None)["grads"]
    x, m = params.copy(), np.zeros_like(params)
    o.rmsprop(x, g, m, 3e-4, 0.99, 1e-8, 0.0, 10.0)
    assert np.abs(res[0][1] - x).max() < 1e-6
