"""profiles/<tag>_*: condensed rocprofv3 evidence + profiles/pmc_traffic.json (HBM bytes per launch of the
bench's kernels from the FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE x2 for wide coalesced reads on gfx950, WRITE_SIZE as is)."""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict

src, tag, utts = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100000
os.makedirs("profiles", exist_ok=True)
txt = subprocess.run([sys.executable, "scripts/prof_summary.py", src], capture_output=True, text=True).stdout
keep = [l for l in txt.split("\n") if "at::native" not in l and "rocclr" not in l]
# drop counter lines that belonged to the removed torch kernels
out, skip = [], False
for l in txt.split("\n"):
    if l.startswith("  ") and not l.startswith("     "):
        skip = ("at::native" in l) or ("rocclr" in l) or ("elementwise" in l)
    if l.startswith("==") or not l.startswith(" "):
        skip = False
    if l.startswith("{'Name'") and "sapr::" not in l:
        continue
    if not skip:
        out.append(l)
open(f"profiles/{tag}_rocprofv3_summary.txt", "w").write(
    f"# rocprofv3 (--kernel-trace --stats; separate --pmc passes) of: python bench.py --steps 5 --warmup 2 --no-cpu-baseline\n"
    f"# {utts} utterances x 1 s per step on one MI355X; sapr kernels only.  SQ_* cycle counters are quad-cycles summed over waves.\n"
    + "\n".join(out))


def pmc(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "sapr" in row["Kernel_Name"]:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
res = {}
for key, pat in (("mfcc", "mfcc_kernel"), ("viterbi", "viterbi_bidiag_kernel"), ("backtrace", "viterbi_backtrace_kernel")):
    f = [v for k, vs in fetch.items() if pat in k for v in vs]
    w = [v for k, vs in write.items() if pat in k for v in vs]
    if not f or not w:
        continue
    fkb, wkb = sum(f) / len(f), sum(w) / len(w)
    res[key] = {"utts": utts, "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
                "hbm_bytes_per_launch": fkb * 1024 * 2 + wkb * 1024,
                "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced reads) + WRITE_SIZE"}
json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
