"""Stand-alone form of the build's scalar-load hazard scan (sapr_amd/asm_scan.py has the why and the walk):
compiles every translation unit that includes emission.h to gfx950 assembly and scans it.  `python -m sapr_amd.build`
already does this on every (re)compile and records the result in sapr_amd/csrc/sload_scan.json; run this after a ROCm
upgrade to re-check without rebuilding the library:

    python scripts/verify/check_sload_hazard.py [source.hip ...]      (minutes; exits non-zero on a violation)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sapr_amd import build  # noqa: E402
from sapr_amd.asm_scan import check  # noqa: E402


def main():
    sources = sys.argv[1:] or [s for s in build.SOURCES if build.uses_sload_idiom(s)]
    total = bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sources:
            out = os.path.join(tmp, src.replace(".hip", ".s"))
            subprocess.check_call([build._hipcc(), *build.FLAGS, "-S", "--cuda-device-only",
                                   os.path.join(build.CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
            n, b = check(out)
            print(f"{src}: {n} hand-written scalar loads checked, {b} violations")
            total += n
            bad += b
    sys.exit(1 if bad or not total else 0)


if __name__ == "__main__":
    main()
