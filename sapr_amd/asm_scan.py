"""Scan of gfx950 assembly for the hand-written scalar-load idiom of csrc/emission.h.

`sload8` / `sload2` issue `s_load_dwordx8` in one `asm volatile` and the matching `s_waitcnt lgkmcnt(0)` lives in a
SEPARATE asm statement (so that the fp64 work of the previous group runs under the load).  That is only correct while
the compiler puts nothing between the two that READS the destination SGPRs — a copy such as s_mov_b32 / s_mov_b64 /
v_writelane_b32 / v_mov_b32 v, s would read stale data.  `check()` walks every hand-written load (between
`;;#ASMSTART` / `;;#ASMEND` markers) forward to the next hand-written wait and reports such reads.

`sapr_amd.build` runs it on the assembly of every translation unit that includes emission.h (`-save-temps`), fails the
build on a violation and records the counts in `csrc/sload_scan.json`; `tests/test_build_guards_cpu.py` checks that
record; `scripts/verify/check_sload_hazard.py` is the stand-alone form.
"""
from __future__ import annotations

import os
import re

SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path, verbose=True):
    """(hand-written scalar loads seen, violations) for one device assembly file."""
    bad = n_loads = 0
    in_asm = False
    pending = []  # [(dest regs, line no)]
    with open(path, encoding="utf-8", errors="replace") as fh:
        for no, ln in enumerate(fh, 1):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            code = t.split(";")[0].strip()
            if in_asm and code.startswith("s_load_dwordx"):
                dest = code.split()[1].rstrip(",")
                pending.append((sregs(dest), no))
                n_loads += 1
                continue
            if in_asm and code.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
                pending = []
                continue
            if code.startswith("s_endpgm"):
                pending = []
                continue
            if not pending:
                continue
            parts = code.split(None, 1)
            if len(parts) < 2:
                continue
            ops = parts[1].split(",")
            # operand 0 is the destination for ALU / move instructions; reads are the rest
            reads = sregs(",".join(ops[1:])) if not parts[0].startswith(("s_cbranch", "s_branch")) else set()
            for dest, at in pending:
                hit = dest & reads
                if hit:
                    if verbose:
                        print(f"{os.path.basename(path)}:{no}: `{code}` reads s{sorted(hit)} loaded at line {at} "
                              f"before its wait")
                    bad += 1
    return n_loads, bad
