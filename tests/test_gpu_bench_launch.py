"""bench.py's multi-rank path END TO END on the one GPU of the box.

`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank, gloo rendezvous) -- id broadcast,
nvqa_comm_init, warm-up, timed blocks between barriers with the MAX over ranks, the profiled pass WITH a communicator,
rank 0's JSON line -- had never run past nvqa_create: the build container has no GPU and librccl refuses two ranks on one
device.  NVQA_BENCH_ONE_DEVICE=1 maps every rank to device 0 and NVQA_RCCL_LIB points the library's exchange at the
stand-in's shared-memory all-reduce (tests/shim, NCCL_SHIM_SHM: a real cross-process sum), so everything except librccl
itself is what an 8-GPU node will execute.  The same launch with real RCCL runs whenever the box has two devices."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim", "libnccl_shim.so")


def _launch(extra_env, gpus=2, extra_args=()):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "2",
           "--blocks", "2"] + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"rc {r.returncode}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-6000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line from rank 0, got {len(lines)}:\n{r.stdout[-3000:]}"
    return json.loads(lines[0]), r.stderr


def _check_line(out, world, persistent=True):
    assert out["n_gpus"] == world and out["config"]["global_batch"] == 512 * world and out["config"]["parallelism"] == f"dp{world}"
    assert out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 2
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 512 * world / (out["ms_per_step"] * 1e-3)) < 1e-3 * out["value"]
    tb = out["timed_blocks"]
    assert tb["blocks"] == 2 and tb["ms_per_step_min"] <= out["ms_per_step"] <= tb["ms_per_step_max"]
    assert out["final_loss"] == out["final_loss"] and 0 < out["final_loss"] < 20           # finite
    assert out["persistent"] == {"fwd": persistent, "bwd": persistent}
    # the profiled pass ran with the communicator: the exchange was timed on its own stream
    assert out["kernel_ms_per_step"]["allreduce"] > 0 and out["roofline"]["frac"] > 0
    assert "cpu_baseline" not in out and "secondary" not in out   # rank-0, N = 1 legs only


def test_gpus2_on_one_device_through_the_shared_memory_exchange():
    assert os.path.exists(SHIM), "tests/shim/libnccl_shim.so missing: run __graft_entry__.build()"
    out, err = _launch({"NVQA_BENCH_ONE_DEVICE": "1", "NVQA_RCCL_LIB": SHIM, "NCCL_SHIM_SHM": f"/nvqa_bench_{os.getpid()}",
                        "NCCL_SHIM_SHM_MB": "32", "NCCL_SHIM_DELAY_US": "0", "NCCL_SHIM_CUS": "0",
                        # TWO processes share the one GPU here.  Each rank's persistent forward launch wants all 256 CUs
                        # resident at once; dispatched concurrently from two processes' queues, each could hold half the chip
                        # and wait for the other half (bounded spins -> a reported time-out, not a hang).  That conflict
                        # cannot happen with one rank per GPU, so the rehearsal runs the per-level kernels; the persistent
                        # kernels beside a communicator are held by tests/test_gpu_dp_fullsize.py (one process, CU-holding
                        # all-reduce) and tests/test_gpu_dp.py (two processes, grids that fit side by side).
                        "NVQA_PERSIST": "0"})
    _check_line(out, 2, persistent=False)
    assert "rehearsal" in out and os.path.realpath(out["collective_library"]) == os.path.realpath(SHIM)
    assert "collective library" in err   # nvqa_comm_init names the library it resolved, once per process


def test_gpus2_real_rccl():
    import torch
    if torch.cuda.device_count() < 2:  # (counting devices does not initialise the GPU in this process)
        pytest.skip("needs 2 GPUs: RCCL refuses two ranks on one device")
    out, err = _launch({k: "" for k in ()})
    _check_line(out, 2)
    assert "librccl" in out["collective_library"]
