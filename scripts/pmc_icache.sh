#!/bin/bash
# instruction-cache counters of one kernel:  scripts/pmc_icache.sh <tag> <kernel substring> script.py [args]
set -e
TAG=$1; PAT=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_ic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$REPO:$PYTHONPATH
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $OUT/ic -- python3 $REPO/"$@" > $OUT/ic.log 2>&1 || tail -5 $OUT/ic.log
cd $REPO
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
