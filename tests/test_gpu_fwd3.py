"""The direct-operand forward kernel (csrc/lstm_persist_fwd3.h, round 4) under conditions the default run never meets.

The kernel's switches are read once per process, so each case is a child pytest run of the headline parity case
(tests/test_gpu_parity_r2.py: B = 512, all lengths 26, dropout on, against the f64 oracle) with the switch set:

* NVQA_PF_DBG=256 -- layer 0 sleeps at the start of every chain-step, so the workgroups of layer 1 really WAIT at the
  counters of their input rows (alone on the device the producers are always early and the waits are register compares);
* NVQA_FWD_KERNEL=1 -- round 2's LDS-ring kernel, which ragged batches, B not a multiple of 128 and bf16 still use, on the
  shape the new kernel has taken over."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"NVQA_PF_DBG": "256"}, {"NVQA_FWD_KERNEL": "1"}], ids=["slow_layer0", "ring_kernel"])
def test_headline_case_in_a_child_process(env):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity_r2.py"), "-q", "-x", "-m", "gpu",
                        "-k", "headline or persistent_forward_lstm and arch1_all26"],
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
