"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2` (no torch.distributed.run around it) must
start two ranks itself, get both through the gloo rendezvous and reach nvqa_create -- which fails here with
"no HIP device" because this container has no GPU (the library has no CPU fallback)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU dry-run: on a GPU box the ranks would really start")
def test_gpus2_self_spawns_and_reaches_nvqa_create():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=600)
    err = r.stderr
    assert r.returncode != 0
    for rank in (0, 1):
        assert f"bench.py rank {rank}/2: nvqa_create failed" in err, err[-2000:]
    assert "no HIP device available" in err
