"""The direct-operand forward kernel (csrc/lstm_persist_fwd3.h, round 4) under conditions the default run never meets.

The kernel's switches are read once per process, so each case is a child pytest run of the headline parity case
(tests/test_gpu_parity_r2.py: B = 512, all lengths 26, dropout on, against the f64 oracle) with the switch set:

* NVQA_PF_DBG=256 -- layer 0 sleeps at the start of every chain-step, so the workgroups of layer 1 really WAIT at the
  counters of their input rows (alone on the device the producers are always early and the waits are register compares);
* NVQA_FWD_KERNEL=1 -- round 2's LDS-ring kernel, which bf16 and the shapes without an instance still use, on the shape the new
  kernel has taken over.

And the RAGGED instance of the new kernel row by row against the ring kernel (tests/dbg_rag_rows.py: evaluate-mode scores of a
B = 512, lengths U{3..26} batch): round 4 had this instance wrong in the rows = 0, 1 (mod 4) of every row tile but the first of each
chain, from register copies hipcc placed at the control-flow joins of a C++ `if` around the inline-asm MFMAs (DESIGN.md section
4.6); the skip is a branch inside the asm statements now, and tests/test_mfma_hazard_lint.py scans the compiled code."""
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


def test_ragged_instance_row_by_row_against_the_ring_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dbg_rag_rows.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    out = r.stdout
    assert out.count("rc 0") == out.count("== ") >= 6, out[-3000:]
    assert "repeat identical: False" not in out, out[-3000:]                      # every variant bit-reproducible
    assert out.count("rows differing > 1e-4: 0 of 512") == out.count("== ") - 1, out[-3000:]   # every fwd3 variant = the ring kernel's rows
