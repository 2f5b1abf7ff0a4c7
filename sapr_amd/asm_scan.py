"""Scan of gfx950 assembly for the hand-written scalar-load idiom of csrc/emission.h.

`sload8` / `sload2` issue `s_load_dwordx8` in one `asm volatile` and the matching `s_waitcnt lgkmcnt(0)` lives in a
SEPARATE asm statement (so that the fp64 work of the previous group runs under the load).  That is only correct while
the compiler puts nothing between the two that READS the destination SGPRs — a copy such as s_mov_b32 / s_mov_b64 /
v_writelane_b32 / v_mov_b32 v, s would read stale data — or WRITES them (the late-arriving load would overwrite the
new value), and while the path from load to wait is straight-line code.  `check()` walks every hand-written load
(between `;;#ASMSTART` / `;;#ASMEND` markers) forward to the next hand-written wait and reports reads and writes of
the pending registers on EVERY path of the control-flow graph from the load to a hand-written wait (the compiler
moves if-blocks out of line and back: branches are followed, conditional ones on both sides); a path that reaches the
end of the program, an indirect jump or the load itself again without passing a hand-written wait is reported too.

`sapr_amd.build` runs it on the assembly of every translation unit that includes emission.h (`-save-temps`), fails the
build on a violation and records the counts in `csrc/sload_scan.json`; `tests/test_build_guards_cpu.py` checks that
record; `scripts/verify/check_sload_hazard.py` is the stand-alone form.
"""
from __future__ import annotations

import os
import re

FAR = re.compile(r"\(([.\w$]+)-\.Lpost_getpc\d+\)")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


_BRANCH = ("s_cbranch", "s_branch")
_INDIRECT = ("s_setpc", "s_swappc", "s_call")
_NO_DEST = ("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop", "v_cmpx", "s_sleep", "s_barrier")
_STORES = ("global_store", "buffer_store", "ds_write", "scratch_store", "flat_store", "global_atomic", "ds_add")


def _parse(path):
    """[(line no, opcode, reads, writes, kind, target)] for every instruction, and {label: index of the instruction
    that follows it}.  kind: 'load' / 'wait' = hand-written s_load_dwordx* / s_waitcnt lgkmcnt(0), 'branch',
    'cbranch', 'indirect', 'end', '' = anything else."""
    ins, label_at, in_asm, far_target = [], {}, False, None
    with open(path, encoding="utf-8", errors="replace") as fh:
        for no, ln in enumerate(fh, 1):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            code = t.split(";")[0].strip()
            if not code or code.startswith(("//", ".")) and not code.endswith(":"):
                continue
            if code.endswith(":"):
                label_at[code[:-1]] = len(ins)
                continue
            parts = code.split(None, 1)
            op = parts[0]
            ops = parts[1].split(",") if len(parts) > 1 else []
            kind, target, reads, writes = "", None, set(), set()
            m = FAR.search(code) if op == "s_add_u32" else None
            if m:
                far_target = m.group(1)
            if in_asm and op.startswith("s_load_dwordx"):
                kind, writes = "load", sregs(ops[0])
                reads = sregs(",".join(ops[1:]))
            elif in_asm and op.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
                kind = "wait"
            elif op.startswith("s_endpgm"):
                kind = "end"
            elif op.startswith("s_setpc") and far_target is not None:
                # long-branch relaxation: s_getpc_b64 / s_add_u32 (LABEL - .Lpost_getpcN) / s_addc_u32 / s_setpc_b64
                # is a jump to LABEL
                kind, target, far_target = "branch", far_target, None
            elif op.startswith(_INDIRECT):
                kind = "indirect"
            elif op.startswith(_BRANCH):
                kind, target = ("branch" if op.startswith("s_branch") else "cbranch"), (ops[0].strip() if ops else "")
            else:
                # operand 0 is the destination for ALU / move / load instructions, the rest are reads; compares and
                # stores only read
                reads = sregs(",".join(ops[1:]))
                if op.startswith(_NO_DEST) or op.startswith(_STORES):
                    reads |= sregs(ops[0]) if ops else set()
                elif ops:
                    writes = sregs(ops[0])
            ins.append((no, code, reads, writes, kind, target))
    return ins, label_at


def check(path, verbose=True):
    """(hand-written scalar loads seen, violations) for one device assembly file."""
    ins, label_at = _parse(path)
    name = os.path.basename(path)
    n_loads = 0
    found = set()  # (line, message): a shared stretch of code is reported once however many loads cross it

    def report(no, msg):
        if (no, msg) not in found:
            found.add((no, msg))
            if verbose:
                print(f"{name}:{no}: {msg}")

    for i, (at, _, _, dest, kind, _) in enumerate(ins):
        if kind != "load":
            continue
        n_loads += 1
        todo, seen = [i + 1], set()
        while todo:
            k = todo.pop()
            while True:
                if k in seen:
                    break
                seen.add(k)
                if k >= len(ins):
                    report(at, "a path from this load runs off the end of the file without a hand-written wait")
                    break
                no, code, reads, writes, kd, target = ins[k]
                if kd == "wait":
                    break
                if k == i:
                    report(at, "a path from this load comes back to it without a hand-written wait")
                    break
                if kd == "end":
                    break   # nothing reads the registers any more
                if kd == "indirect":
                    report(no, f"`{code}` with the load of line {at} pending (indirect jump: not followed)")
                    break
                hit_r, hit_w = dest & reads, (dest & writes if kd != "load" else set())
                if hit_r:
                    report(no, f"`{code}` reads s{sorted(hit_r)} loaded at line {at} before its wait")
                if hit_w:
                    report(no, f"`{code}` writes s{sorted(hit_w)} loaded at line {at} before its wait")
                if kd in ("branch", "cbranch"):
                    if target not in label_at:
                        report(no, f"`{code}`: unknown label with the load of line {at} pending")
                        break
                    if kd == "cbranch":
                        todo.append(k + 1)
                    k = label_at[target]
                    continue
                k += 1
    return n_loads, len(found)
