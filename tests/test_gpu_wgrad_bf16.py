"""The bf16 weight-gradient kernel (csrc/wgrad_bf16.h: k-major operands, ds_read_b64_tr_b16 fragments,
v_mfma_f32_16x16x32_bf16) against a double-precision sum of the bf16-rounded operands (tools/kbench15's
check as a test).  The kernel is internal to libnvqa (no ABI entry takes a bare product), so the checker program
tests/shim/wgrad_bf16_check compiles the same header and runs the shapes of the step: dW[2048 x N] = dG^T X, K = 14336,
N = 512 and N = 200, f32 rows / bf16 images as operands, split-K 8 and 16; 24 rows x all columns per configuration."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "shim", "wgrad_bf16_check")


def test_wgrad_bf16_against_f64_sum_of_rounded_operands():
    assert os.path.exists(EXE), "tests/shim/wgrad_bf16_check missing: run __graft_entry__.build()"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if "max err" in l]
    assert r.returncode == 0 and len(lines) == 12 and all(l.endswith("OK") for l in lines), r.stdout + r.stderr
    try:
        from util import record
        record("wgrad_bf16_check", {"worst_relative": max(float(l.split("(")[1].split()[0]) for l in lines)})
    except Exception:
        pass
