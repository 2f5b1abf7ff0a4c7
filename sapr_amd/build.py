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
# the exact Viterbi kernels are one translation unit per (D, S) shape: the heavy ones first, so that a cold build on
# 6-8 cores takes about as long as the slowest of them (~3 minutes) instead of one nine-minute compile
SOURCES = ["viterbi_exact_39_18_t1s1.hip", "viterbi_exact_39_18_t1s0.hip", "viterbi_exact_39_18_t0s1.hip",
           "viterbi_exact_39_18_t0s0.hip", "estep.hip", "viterbi_exact_39_10_t1s1.hip", "viterbi_exact_39_10_t1s0.hip",
           "viterbi_exact_39_10_t0s1.hip", "viterbi_exact_39_10_t0s0.hip", "viterbi_exact_13_18.hip",
           "viterbi_bound.hip", "mfcc.hip", "viterbi_exact_13_10.hip", "custom.hip", "viterbi.hip", "common.hip",
           "resample.hip"]
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


SCAN_RECORD = os.path.join(CSRC, "sload_scan.json")

# the sources behind each launch sequence of the headline bench: profiles/pmc_traffic.json is stamped with their hash
# (scripts/make_profile_artifacts.py) and bench.py drops counter-derived figures whose stamp no longer matches
KERNEL_SOURCES = {
    "mfcc": ["mfcc.hip", "mfcc_wave.h", "mfcc_wave_pack.h", "sapr_common.h"],
    "decode": ["viterbi.hip", "viterbi_bound.hip", "viterbi_exact.inc", "viterbi_exact_13_10.hip", "viterbi_shared.h",
               "emission.h", "sapr_common.h"],
}


def source_hash(group: str) -> str:
    """sha256 (16 hex digits) over the sources of one launch sequence, in the order listed above."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[group]:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def uses_sload_idiom(src: str) -> bool:
    """Translation units that include emission.h carry its hand-written s_load / s_waitcnt pairs (asm_scan.py)."""
    return any(os.path.basename(d) == "emission.h" for d in _deps(os.path.join(CSRC, src)))


def _compile(hipcc: str, src: str, obj: str, verbose: bool):
    """One translation unit.  Users of the scalar-load idiom are compiled with -save-temps into a scratch directory so
    that the very assembly the object was made from is scanned (asm_scan.check); a violation fails the build."""
    s = os.path.join(CSRC, src)
    if not uses_sload_idiom(src):
        cmd = [hipcc, *FLAGS, *_extra_flags(), "-c", s, "-o", obj]
        if verbose:
            print("[sapr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return None
    import glob
    import tempfile
    from .asm_scan import check
    with tempfile.TemporaryDirectory(prefix="sapr_build_") as tmp:
        tmp_obj = os.path.join(tmp, os.path.basename(obj))
        cmd = [hipcc, *FLAGS, *_extra_flags(), "-save-temps=obj", "-c", s, "-o", tmp_obj]
        if verbose:
            print("[sapr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL if not verbose else None)
        asm = glob.glob(os.path.join(tmp, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
        if len(asm) != 1:
            raise RuntimeError(f"{src}: expected one gfx950 assembly file from -save-temps, found {asm}")
        loads, bad = check(asm[0])
        if bad:
            raise RuntimeError(f"{src}: {bad} hazard(s) between hand-written scalar loads and their s_waitcnt "
                               "(sapr_amd/asm_scan.py); the object was NOT installed")
        shutil.move(tmp_obj, obj)
    return {"loads": loads, "violations": bad}


def build(force: bool = False, verbose: bool = True) -> str:
    import json
    hipcc = _hipcc()
    objs, jobs = [], []
    try:
        with open(SCAN_RECORD) as fh:
            scanned = set(json.load(fh))
    except (OSError, ValueError):
        scanned = set()
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        # an object whose assembly was never scanned (record lost) is rebuilt: the record must cover what is linked
        if force or _stale(o, _deps(s) + [__file__]) or (uses_sload_idiom(src) and src not in scanned):
            jobs.append((src, o))
        objs.append(o)
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 8)) as pool:
            scans = list(pool.map(lambda j: _compile(hipcc, j[0], j[1], verbose), jobs))
        record = {}
        if os.path.exists(SCAN_RECORD):
            try:
                with open(SCAN_RECORD) as fh:
                    record = json.load(fh)
            except ValueError:
                record = {}
        for (src, _), r in zip(jobs, scans):
            if r is not None:
                record[src] = r
        record = {k: v for k, v in record.items() if k in SOURCES}
        with open(SCAN_RECORD, "w") as fh:
            json.dump(record, fh, indent=1, sort_keys=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[sapr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
