#!/usr/bin/env python3
"""Static scan of a gfx950 assembly listing (hipcc -S) for a hazard the assembler cannot see: the persistent kernels issue their
MFMAs as inline asm, so the compiler's hazard recogniser does not know that the destination registers are written by the matrix
pipe several passes later (MFMAs from __builtin_amdgcn_mfma_* are the compiler's own business and are not tracked), and any instruction IT places behind such an MFMA that reads those registers (a copy at a control-flow
join, a spill, a ds_write) sees stale data unless the wait states are there.

The scan walks every kernel in listing order and, at every branch, also walks the first instructions of the branch target with
the state at the branch (a jump from the last MFMA of a block straight to a reader).  Time is counted in WAIT STATES, the unit
of the ISA manual's hazard tables (one issue slot of the SIMD = 4 cycles): 1 per instruction, k + 1 per s_nop k; an 8-pass MFMA
(v_mfma_f32_16x16x4_f32) issues when the matrix pipe is free, 8 slots after the MFMA before it.  CDNA3 ISA, "XDL write VGPR ->
VALU / VMEM / LDS read" and "-> VALU write", 8 passes: 11 wait states between the two; anything closer is reported.

usage: mfma_hazard_scan.py listing.s [kernel-name-substring] [min-wait-states=11]      exit code 1 when something is found"""
import re, sys

def regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m: return [(m.group(1), int(m.group(2)))]
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return []

NODST = ("ds_write", "buffer_store", "global_store", "scratch_store", "ds_add", "global_atomic", "buffer_atomic")
NOVALU = ("buffer_load", "global_load", "ds_read", "scratch_load", "s_")

def scan(lines, labels, start, state, need, path, out, budget=None, via=""):
    """walk from lines[start]; state = [clock, pipe_free, {reg: (slot, line)}]; budget: stop after that many wait states (branch targets)"""
    clock, pipe_free, last = state
    t0 = clock
    i = start
    while i < len(lines):
        ln, s = lines[i]; i += 1
        if s.endswith(":") or s.startswith("."): continue
        parts = s.split(None, 1); op = parts[0]; args = parts[1].split(",") if len(parts) > 1 else []
        if op == "s_endpgm": break
        if budget is not None and clock - t0 > budget: break
        if op == "s_nop": clock += int(args[0]) + 1; continue
        if op.startswith("v_mfma"):
            clock = max(clock, pipe_free)
            for r in regs(args[0]): last[r] = (clock, ln)
            pipe_free = clock + 8; clock += 1
            continue
        clock += 1
        nodst = op.startswith(NODST)
        for a in (args if nodst else args[1:]):
            for r in regs(a):
                if r in last and clock - 2 - last[r][0] < need:
                    out.add(f"{path}:{ln}: {s}   <- reads {r[0]}{r[1]}, written by the MFMA at line {last[r][1]}, {clock - 2 - last[r][0]} wait states between{via}")
        if not nodst and args:
            if not op.startswith(NOVALU):
                for r in regs(args[0]):
                    if r in last and clock - 2 - last[r][0] < need:
                        out.add(f"{path}:{ln}: {s}   <- OVERWRITES {r[0]}{r[1]}, destination of the MFMA at line {last[r][1]}, {clock - 2 - last[r][0]} wait states between{via}")
            for r in regs(args[0]): last.pop(r, None)
        if budget is None and (op.startswith("s_cbranch") or op == "s_branch"):
            tgt = args[0].strip()
            if tgt in labels and any(clock - v[0] < need + 2 for v in last.values()):
                scan(lines, labels, labels[tgt], [clock, pipe_free, dict(last)], need, path, out, budget=need + 2, via=f" (via the branch at line {ln})")
        if op == "s_branch" and budget is not None: # follow an unconditional jump inside a target walk
            tgt = args[0].strip()
            if tgt in labels: i = labels[tgt]
    return

def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""; need = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    kernels = {}; cur = None
    in_asm = False
    for ln, line in enumerate(open(path), 1):
        if "#ASMSTART" in line: in_asm = True
        if "#ASMEND" in line: in_asm = False
        s = line.split(";")[0].strip()
        if not s: continue
        if s.startswith("v_mfma") and not in_asm: s = "compiler_" + s   # an MFMA the compiler emitted itself: its hazards are the compiler's
        m = re.match(r"^(_Z\w+):", s)
        if m: cur = m.group(1); kernels[cur] = []; continue
        if cur is not None: kernels[cur].append((ln, s))
        if s == "s_endpgm": cur = None
    total = 0
    for name, lines in kernels.items():
        if want not in name: continue
        labels = {s[:-1]: i for i, (ln, s) in enumerate(lines) if s.endswith(":")}
        out = set()
        scan(lines, labels, 0, [0, 0, {}], need, path, out)
        for o in sorted(out, key=lambda x: int(x.split(":")[1])): print(o)
        print(f"{name}: {len(out)} findings"); total += len(out)
    sys.exit(1 if total else 0)

main()
