"""Condense rocprofv3 csv outputs of scripts/prof.sh into a small text summary."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


for f in find("trace", "*kernel_trace.csv"):
    print("== kernel trace (sapr kernels):", os.path.relpath(f, out))
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "sapr" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], row["Grid_Size_X"], row["VGPR_Count"], row["LDS_Block_Size"])].append(
                int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for (name, grid, vgpr, lds), v in acc.items():
        print(f"  {name}  grid={grid} vgpr={vgpr} lds={lds}  calls={len(v)} avg={sum(v)/len(v)/1e6:.4f} ms "
              f"min={min(v)/1e6:.4f} max={max(v)/1e6:.4f}")
for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters:", sub)
        for name, cs in acc.items():
            short = name[:90]
            print(" ", short)
            for c, v in cs.items():
                print(f"     {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
