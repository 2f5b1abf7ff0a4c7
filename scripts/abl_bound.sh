#!/bin/bash
# Dev tool (GPU box): kernel time of the bounding pass for each variant library given, from a rocprofv3 kernel trace
#   scripts/abl_bound.sh "<args of time_bound.py>" name1 name2 ...   (name "" = the default library)
REPO=$(pwd); export TMPDIR=/tmp; export PYTHONPATH=$REPO
ARGS=$1; shift
cd /tmp
for n in "$@"; do
  if [ "$n" = "default" ]; then unset SAPR_LIB; else export SAPR_LIB=$REPO/sapr_amd/libsapr_hip_$n.so; fi
  rm -rf /tmp/abl_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$n -- python3 $REPO/scripts/time_bound.py $ARGS > /tmp/abl_$n.log 2>&1
  f=$(find /tmp/abl_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("viterbi_bound", "viterbi_approx", "viterbi_bidiag")):
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']} avg {float(r['AverageNs']) / 1e6:.3f} ms")
PY
done
