#!/bin/bash
# Dev tool (GPU box): per-kernel times (rocprofv3 kernel trace) of a python script:  scripts/kern_times.sh script.py [args]
REPO=$(pwd); export TMPDIR=/tmp; export PYTHONPATH=$REPO
cd /tmp; rm -rf /tmp/kt_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_out -- python3 $REPO/"$@" > /tmp/kt.log 2>&1
f=$(find /tmp/kt_out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sapr" in r["Name"] or "kernel" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    n = r["Name"].replace("sapr::(anonymous namespace)::", "").split("(")[0]
    print(f"  {n[:70]:70s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e6:8.4f} min {float(r['MinNs']) / 1e6:8.4f} max {float(r['MaxNs']) / 1e6:8.4f} ms")
PY
