#!/bin/bash
# kernel-trace of scripts/time_decode_only.py, pruned-decoder kernels only (dev tool; run on the GPU box)
#   scripts/prof_dec.sh <tag> [env assignments ...]
TAG=$1; shift
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/pd_$TAG
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pd_$TAG -- python3 $R/scripts/time_decode_only.py > $R/gpurun_out/pd_$TAG.log 2>&1 < /dev/null
cd $R
echo "== $TAG: $(grep candidates gpurun_out/pd_$TAG.log)"
for f in gpurun_out/pd_$TAG/*/*kernel_stats.csv; do
  grep -E "approx|bidiag|select|backtrace" "$f" | sed -E "s/\(anonymous namespace\):://; s/\(.*\)\"//" | cut -d, -f1,2,4 | cut -c1-110
done
