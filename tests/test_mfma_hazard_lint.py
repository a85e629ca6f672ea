"""Static check of the compiled persistent kernels whose MFMAs are inline asm (csrc/lstm_persist_fwd3.h, lstm_persist_bwd3.h, lstm_persist_bwd2.h).

hipcc does not know that those asm statements are matrix instructions, so whatever IT places behind them -- the register copies
of a control-flow join, a spill -- gets no wait states.  Round 4 lost the ragged instances of the forward kernel to exactly that
(DESIGN.md section 4.6); novel-vqa_amd/tools/mfma_hazard_scan.py walks the assembly and reports every reader / overwriter of an
inline-asm MFMA's destination closer than the ISA's 11 wait states.  No GPU: hipcc -S cross-compiles (about 40 s per file)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


TUS = ["persist_fwd3", "persist_bwd", "persist_bwd_ring"]


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    """the three translation units compiled to assembly side by side (the longest takes about 1.5 min)"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("asm")
    procs = {tu: subprocess.Popen([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-c++20-extensions", "-I" + os.path.join(ROOT, "include"),
                                   "--cuda-device-only", "-S", "-o", str(d / (tu + ".s")), os.path.join(ROOT, "novel-vqa_amd", "csrc", tu + ".hip")],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE) for tu in TUS}
    for tu, pr in procs.items():
        out, err = pr.communicate(timeout=1200)
        assert pr.returncode == 0, err.decode()[-2000:]
    return d


@pytest.mark.parametrize("tu", TUS)
def test_no_reader_inside_an_inline_mfma_latency(tu, listings):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "novel-vqa_amd", "tools", "mfma_hazard_scan.py"), str(listings / (tu + ".s"))], capture_output=True, text=True)
    assert "findings" in r.stdout, r.stdout[-500:] + r.stderr[-500:]   # (kernels were found and walked)
    assert r.returncode == 0, r.stdout[-3000:]


def test_the_scanner_sees_a_planted_hazard(tmp_path):
    lst = tmp_path / "t.s"
    lst.write_text("_Zk:\n\t;;#ASMSTART\n\tv_mfma_f32_16x16x4_f32 v[0:3], v8, a0, v[0:3]\n\t;;#ASMEND\n\ts_nop 3\n\tv_mov_b64_e32 v[10:11], v[2:3]\n\ts_endpgm\n"
                   "_Zok:\n\t;;#ASMSTART\n\tv_mfma_f32_16x16x4_f32 v[0:3], v8, a0, v[0:3]\n\t;;#ASMEND\n\ts_nop 10\n\tv_mov_b64_e32 v[10:11], v[2:3]\n\ts_endpgm\n"
                   "_Zbuiltin:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v8, a0, v[0:3]\n\tv_mov_b64_e32 v[10:11], v[2:3]\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "novel-vqa_amd", "tools", "mfma_hazard_scan.py"), str(lst)], capture_output=True, text=True)
    assert r.returncode == 1 and "_Zk: 2 findings" in r.stdout and "_Zok: 0 findings" in r.stdout and "_Zbuiltin: 0 findings" in r.stdout, r.stdout
