"""Build recipe for libsapr_hip.so (hipcc, gfx950 only, in-tree).

``python -m sapr_amd.build`` or ``sapr_amd.build.build()``.  No torch headers are
involved: the library is a plain C-ABI shared object (include/sapr_hip.h).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsapr_hip.so")
SOURCES = ["common.hip", "viterbi.hip", "estep.hip", "custom.hip", "resample.hip", "mfcc.hip"]
# -ffp-contract=off: the trellis kernels must perform the individually rounded IEEE
# operations numpy performs (bit-identical Viterbi scores); kernels that want FMAs
# call fma()/__builtin_fmaf explicitly.
FLAGS = ["-O3", "-std=c++17", "-ftemplate-depth=2048", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _extra_flags() -> list:
    """Developer switch: extra compiler flags (e.g. -DSAPR_Q_EARLY=8) without editing the recipe."""
    return os.environ.get("SAPR_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libsapr_hip.so cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(path: str, seen=None) -> list:
    """The file and every header it includes with quotes, transitively (a header edit rebuilds only its users)."""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return []
    seen.add(path)
    out = [path]
    with open(path, encoding="utf-8") as fh:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
            out += _deps(os.path.join(os.path.dirname(path), inc), seen)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, _deps(s) + [__file__]):
            jobs.append([hipcc, *FLAGS, *_extra_flags(), "-c", s, "-o", o])
        objs.append(o)
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[sapr_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 6)) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[sapr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
