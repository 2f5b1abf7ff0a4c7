"""Build-time check for the inline-asm scalar-load idiom of emission.h ("sload8 issues s_load_dwordx8 in
one asm volatile and the matching s_waitcnt lives in a separate asm; a compiler-inserted copy of the destination SGPRs
in between would read stale data").

Compiles viterbi.hip and estep.hip to gfx950 assembly and, for every hand-written `s_load_dwordx8 s[a:b]`, walks
forward to the next hand-written `s_waitcnt lgkmcnt(0)`: no instruction in between may READ any of s[a:b] (a copy
such as s_mov_b32 / s_mov_b64 / v_writelane_b32 / v_mov_b32 v, s would).  Compiler-generated s_loads into OTHER
registers and their waits are fine.  Run after a ROCm upgrade or any change to emission.h:

    python scripts/verify/check_sload_hazard.py        (2-3 minutes; exits non-zero on a violation)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["-O3", "-std=c++17", "-ftemplate-depth=2048", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wno-unused-function", "-S", "--cuda-device-only"]
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    lines = open(path).read().split("\n")
    bad = n_loads = 0
    in_asm = False
    pending = []  # [(dest regs, line no)]
    for no, ln in enumerate(lines, 1):
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
                print(f"{os.path.basename(path)}:{no}: `{code}` reads s{sorted(hit)} loaded at line {at} before its wait")
                bad += 1
    return n_loads, bad


def main():
    total = bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("viterbi.hip", "estep.hip"):
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", os.path.join(ROOT, "sapr_amd", "csrc"),
                                   os.path.join(ROOT, "sapr_amd", "csrc", src), "-o", out],
                                  stderr=subprocess.DEVNULL)
            n, b = check(out)
            print(f"{src}: {n} hand-written scalar loads checked, {b} violations")
            total += n
            bad += b
    sys.exit(1 if bad or not total else 0)


if __name__ == "__main__":
    main()
