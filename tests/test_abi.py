"""The C-ABI library: every function include/nvqa.h declares is exported and bound, and the
product path fails loudly when no HIP device exists (no CPU fallback).  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "nvqa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nvqa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.binding.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nvqa.h but not exported"
    assert sorted(pkg.binding.SYMBOLS) == names, "binding.SYMBOLS must mirror include/nvqa.h"


def test_struct_sizes_match_header(pkg):
    assert ctypes.sizeof(pkg.binding.Dims) == 40
    assert ctypes.sizeof(pkg.binding.Dropout) == 24


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(pkg.binding.NvqaError):
        pkg.binding.load_library(str(tmp_path / "libnvqa.so"))


def test_no_device_is_an_error_not_a_fallback(pkg, orc):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = orc.make_dims(arch=1, B=4, T=5, V=11, E=8, R=8, L=2, I=12, C=12, A=8)
    with pytest.raises(pkg.binding.NvqaError) as e:
        pkg.binding.Context(pkg.binding.Dims(*[getattr(d, n) for n, _ in d._fields_]), 0)
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)


def test_product_package_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "novel-vqa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".lua")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower().replace("cpu oracle", "").replace("the oracle", ""), \
                    f"{f} references the oracle"


def test_lua_cdef_declares_every_header_function():
    """novel-vqa_amd/lua/nvqa_ffi.lua is the reference-side binding: its ffi.cdef must name
    every function include/nvqa.h declares, or a LuaJIT caller cannot reach it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "nvqa.h")).read()
    lua = open(os.path.join(root, "novel-vqa_amd", "lua", "nvqa_ffi.lua")).read()
    names = set(re.findall(r"\b(nvqa_[a-z0-9_]+)\s*\(", hdr))
    missing = sorted(n for n in names if not re.search(r"\b%s\s*\(" % n, lua))
    assert not missing, missing


def _prototypes(text):
    """{name: normalised prototype} of every nvqa_* function declared in a block of C declarations: comments stripped,
    white space collapsed, parameter NAMES dropped (the header and the cdef may call them differently), types kept."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(nvqa_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = re.sub(r"\b(extern|NVQA_API)\b", "", ret)

        def norm_type(t):
            t = re.sub(r"\s+", " ", t).strip()
            t = re.sub(r"\s*\*\s*", "*", t)
            return t

        params = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            arr = re.search(r"\[(\d*)\]\s*$", a)  # `size_t out[3]` is a pointer parameter
            a = re.sub(r"\[\d*\]\s*$", "", a).strip()
            mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a, flags=re.S)
            ty = mm.group(1) if mm and mm.group(1).strip() and mm.group(2) not in ("int", "float", "double", "void", "size_t", "int32_t", "int64_t", "uint64_t", "char") else a
            ty = norm_type(ty) + ("*" if arr else "")
            params.append(ty)
        out[name] = norm_type(ret) + " " + name + "(" + ", ".join(params) + ")"
    return out


def test_lua_cdef_prototypes_equal_the_header():
    """Every prototype of novel-vqa_amd/lua/nvqa_ffi.lua's ffi.cdef -- return type and every parameter type, in order -- is
    the one include/nvqa.h declares, and the two POD structs have the same fields.  (LuaJIT binds by this text alone: a
    drifted cdef would call the library with a wrong stack layout.)"""
    hdr = open(os.path.join(ROOT, "include", "nvqa.h")).read()
    lua = open(os.path.join(ROOT, "novel-vqa_amd", "lua", "nvqa_ffi.lua")).read()
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S).group(1)
    ph, pl = _prototypes(hdr), _prototypes(cdef)
    assert len(ph) >= 30 and set(ph) == set(pl), sorted(set(ph) ^ set(pl))
    diff = {n: (ph[n], pl[n]) for n in ph if ph[n] != pl[n]}
    assert not diff, diff

    def struct_fields(text, name):
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        body = re.search(r"struct\s+%s\s*\{(.*?)\}" % name, text, flags=re.S).group(1)
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            ty, names = re.match(r"^(.*?)\s+([A-Za-z0-9_,\s]+)$", decl, flags=re.S).groups()
            fields += [(re.sub(r"\s+", " ", ty), n.strip()) for n in names.split(",")]
        return fields
    for st in ("nvqa_dims", "nvqa_dropout"):
        assert struct_fields(hdr, st) == struct_fields(cdef, st), st


def _lua_code(path):
    """Lua source with comments and string literals blanked (enough for the static checks below)."""
    s = open(path).read()
    s = re.sub(r"--\[\[.*?\]\]--?", " ", s, flags=re.S)       # block comments
    s = re.sub(r"\[\[.*?\]\]", '""', s, flags=re.S)           # long strings (the cdef)
    s = re.sub(r"--[^\n]*", " ", s)                           # line comments
    s = re.sub(r"'(?:\\.|[^'\\\n])*'", "''", s)
    s = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', s)
    return s


def _call_args(code, start):
    """number of top-level arguments of the call whose '(' is at code[start]"""
    depth, n, i, empty = 0, 1, start, True
    while i < len(code):
        c = code[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return 0 if empty else n
        elif depth == 1 and c == ",":
            n += 1
        elif depth >= 1 and not c.isspace():
            empty = False
        i += 1
    raise AssertionError("unbalanced call")


def test_lua_scripts_call_the_abi_with_the_declared_arity_and_balance_their_blocks():
    """The Lua twins cannot be executed in the build image (no LuaJIT), so they are at least held to what can be checked
    statically: every nvqa.lib.nvqa_* call names a function of include/nvqa.h and passes as many arguments as its prototype
    takes, and every block opener (function / if / do / repeat) has its closer."""
    hdr = open(os.path.join(ROOT, "include", "nvqa.h")).read()
    protos = _prototypes(hdr)
    arity = {n: (0 if p.endswith("()") else p.count(",") + 1) for n, p in protos.items()}
    lua_dir = os.path.join(ROOT, "novel-vqa_amd", "lua")
    files = sorted(f for f in os.listdir(lua_dir) if f.endswith(".lua"))
    assert len(files) >= 9
    for fn in files:
        code = _lua_code(os.path.join(lua_dir, fn))
        for m in re.finditer(r"nvqa\.lib\.(nvqa_[a-z0-9_]+)\s*\(", code):
            name = m.group(1)
            assert name in arity, f"{fn}: {name} is not declared in include/nvqa.h"
            got = _call_args(code, m.end() - 1)
            assert got == arity[name], f"{fn}: {name} called with {got} arguments, the header declares {arity[name]}"
        opens = len(re.findall(r"\bfunction\b", code)) + len(re.findall(r"(?<![A-Za-z_])if\b", code)) + len(re.findall(r"\bdo\b", code))
        opens -= len(re.findall(r"\belseif\b", code)) * 0     # 'elseif' is one token: the regex above does not match inside it
        closes = len(re.findall(r"\bend\b", code))
        assert opens == closes, f"{fn}: {opens} block openers, {closes} 'end'"
        assert len(re.findall(r"\brepeat\b", code)) == len(re.findall(r"\buntil\b", code))
