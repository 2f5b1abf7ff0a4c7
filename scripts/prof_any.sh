#!/bin/bash
# rocprofv3 passes (kernel trace + SQ / FETCH / WRITE counter passes, each its own run) of an arbitrary python
# command, condensed into profiles/<tag>_rocprofv3_summary.txt (sapr kernels only).  Run on the GPU box:
#   scripts/prof_any.sh <tag> script.py [args...]
set -e
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT
mkdir -p $OUT $REPO/profiles
export TMPDIR=/tmp
export PYTHONPATH=$REPO:$PYTHONPATH
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/"$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/"$@" > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/"$@" > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/"$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/"$@" > $OUT/pmc_write.log 2>&1
cd $REPO
python3 scripts/make_profile_artifacts.py $OUT $TAG "$*" > $OUT/artifacts.log 2>&1 || true
tail -5 $OUT/artifacts.log
