#!/bin/bash
# kernel-trace statistics of an arbitrary python command, sapr kernels only:  scripts/prof_cmd.sh <tag> script.py [args]
set -e
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/kt_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$REPO:$PYTHONPATH
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/"$@" > $OUT/run.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sapr" in r["Name"]:
            print(f'{float(r["AverageNs"])/1e6:9.4f} ms x{r["Calls"]:>4}  {r["Name"][:110]}')
PY
