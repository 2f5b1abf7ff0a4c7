"""profiles/<tag>_rocprofv3_summary.txt: condensed rocprofv3 evidence of one profiled command (sapr kernels only) and,
for the bench command, profiles/pmc_traffic.json (HBM bytes per launch of the bench's kernels from the FETCH_SIZE /
WRITE_SIZE passes, corrected as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE x2 for wide coalesced reads on
gfx950, WRITE_SIZE as is).
    python scripts/make_profile_artifacts.py <rocprof out dir> <tag> "<command that was profiled>" [utts]"""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sapr_amd.build import source_hash  # noqa: E402

src, tag = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else "bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
utts = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
os.makedirs("profiles", exist_ok=True)
txt = subprocess.run([sys.executable, "scripts/prof_summary.py", src], capture_output=True, text=True).stdout
out, skip = [], False
for l in txt.split("\n"):
    if l.startswith("  ") and not l.startswith("     "):
        skip = "sapr" not in l
    if l.startswith("==") or not l.startswith(" "):
        skip = False
    if l.startswith("{'Name'") and "sapr::" not in l:
        continue
    if not skip:
        out.append(l)


def pmc(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "sapr" in row["Kernel_Name"]:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
valu = pmc("pmc_sq1", "SQ_INSTS_VALU")
traffic = ["== HBM bytes per launch = FETCH_SIZE[KB] x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE[KB] x 1024"]
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
    fkb, wkb = sum(f) / len(f), sum(w) / len(w)
    traffic.append(f"  {k[:100]}\n     fetch {fkb * 2048 / 1e6:10.1f} MB   write {wkb * 1024 / 1e6:10.1f} MB   total {(fkb * 2048 + wkb * 1024) / 1e6:10.1f} MB")
open(f"profiles/{tag}_rocprofv3_summary.txt", "w").write(
    f"# rocprofv3 (--kernel-trace --stats; separate --pmc passes) of: python {cmd}\n"
    f"# one MI355X; sapr kernels only.  SQ_* cycle counters are quad-cycles summed over waves.\n"
    + "\n".join(out) + "\n" + "\n".join(traffic) + "\n")

if "bench.py" in cmd and "--mode" not in cmd:
    res = {}
    groups = {"mfcc": ("mfcc_wave_kernel", "mfcc_wave_finish_kernel", "mfcc_kernel", "mfcc_finish_kernel"),
              "decode": ("viterbi_bound_lds_kernel", "viterbi_approx_kernel", "viterbi_select_kernel", "viterbi_bidiag_kernel", "viterbi_backtrace")}
    for key, pats in groups.items():
        tot_f = tot_w = 0.0
        found = False
        per = {}
        for pat in pats:
            f = [v for k, vs in fetch.items() if pat in k for v in vs]
            w = [v for k, vs in write.items() if pat in k for v in vs]
            if not f and not w:
                continue
            found = True
            fkb = sum(f) / len(f) if f else 0.0
            wkb = sum(w) / len(w) if w else 0.0
            per[pat] = fkb * 2048 + wkb * 1024
            tot_f += fkb
            tot_w += wkb
        if found:
            vi = {pat: sum(v) / len(v) for pat in pats for k, vs in valu.items() if pat + "<" in k or pat + "(" in k
                  for v in [vs]}
            res[key] = {"utts": utts, "source_hash": source_hash(key), "fetch_size_kb_raw": tot_f, "write_size_kb_raw": tot_w,
                        "hbm_bytes_per_launch": tot_f * 1024 * 2 + tot_w * 1024, "per_kernel_bytes": per,
                        "valu_insts_per_launch": sum(vi.values()), "per_kernel_valu_insts": vi,
                        "source": f"profiles/{tag}_rocprofv3_summary.txt (rocprofv3 --pmc passes of: python {cmd})",
                        "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced reads) + "
                                "WRITE_SIZE; 'decode' sums the pruned decoder's kernels (one launch sequence)"}
    json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
print("wrote", f"profiles/{tag}_rocprofv3_summary.txt")
