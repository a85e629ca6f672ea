"""Which collective library nvqa_comm_* runs on, and that loading it never breaks the host process (CPU only).

Round 3 opened librccl RTLD_GLOBAL.  The torch wheel bundles its own librccl (and its own libamdhip64): with the system
copy mapped globally first and `import torch` afterwards, two librccl images interposed each other's global C++ objects and
the interpreter died at exit inside glibc ("double free or corruption", rc 134) -- reproduced without a GPU.  The loader now
opens RTLD_LOCAL and picks the librccl that sits BESIDE THE HIP RUNTIME IMAGE libnvqa itself is bound to (INTEGRATION.md
section 4), so a stream handed to ncclAllReduce always belongs to the runtime that library was built against.  Both import
orders must exit 0, and the resolved path must obey the rule."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "novel-vqa_amd", "libnvqa.so")

BODY = r"""
import ctypes, os, sys
order = sys.argv[1]
if order == "torch_first":
    import torch
lib = ctypes.CDLL(sys.argv[2])
lib.nvqa_comm_library.restype = ctypes.c_char_p
lib.nvqa_last_error.restype = ctypes.c_char_p
p = lib.nvqa_comm_library()
assert p is not None, lib.nvqa_last_error()
buf = ctypes.create_string_buffer(128)
lib.nvqa_comm_unique_id(buf)          # load_rccl + ncclGetUniqueId (may fail without a device: only the exit matters)
if order == "nvqa_first":
    import torch
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "/librccl" in l or "/libamdhip64" in l})
print("RCCL=" + p.decode())
print("MAPS=" + "|".join(maps))
"""


@pytest.mark.parametrize("order", ["torch_first", "nvqa_first"])
def test_collective_library_loads_locally_in_both_import_orders(order):
    assert os.path.exists(LIB), "build libnvqa.so first (__graft_entry__.build())"
    env = {k: v for k, v in os.environ.items() if k != "NVQA_RCCL_LIB"}
    r = subprocess.run([sys.executable, "-c", BODY, order, LIB], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout}\n{r.stderr}"   # 134 = the round-3 abort at exit
    out = dict(l.split("=", 1) for l in r.stdout.splitlines() if "=" in l)
    rccl = os.path.realpath(out["RCCL"])
    maps = [os.path.realpath(m) for m in out["MAPS"].split("|")]
    assert rccl in maps
    torch_lib = os.sep + os.path.join("torch", "lib") + os.sep
    if order == "torch_first":
        # libnvqa's NEEDED libamdhip64.so.7 binds to the image torch mapped first: the collective library is the one
        # torch ships beside it, and no second librccl is mapped
        assert torch_lib in rccl and sum("/librccl" in m for m in maps) == 1, out
    else:
        # libnvqa mapped the ROCm tree's HIP runtime: the collective library comes from the same directory, whatever
        # torch maps for itself afterwards
        assert torch_lib not in rccl, out
        hips = [m for m in maps if "/libamdhip64" in m and torch_lib not in m]
        assert hips and os.path.dirname(hips[0]) == os.path.dirname(rccl), out


def test_named_collective_library_wins():
    """NVQA_RCCL_LIB names the library outright (the test stand-in, or a site's own RCCL build)."""
    shim = os.path.join(ROOT, "tests", "shim", "libnccl_shim.so")
    assert os.path.exists(shim), "tests/shim/libnccl_shim.so missing: run __graft_entry__.build()"
    body = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); lib.nvqa_comm_library.restype=ctypes.c_char_p; "
            "print(lib.nvqa_comm_library().decode())")
    r = subprocess.run([sys.executable, "-c", body, LIB], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, NVQA_RCCL_LIB=shim))
    assert r.returncode == 0, r.stderr
    assert os.path.realpath(r.stdout.strip()) == os.path.realpath(shim)
