#!/bin/bash
# Ablation builds of the MFCC kernel (developer tool): libsapr_hip_ablN.so skips the phases in bit mask N
# (see SAPR_ABLATE in mfcc.hip).  Build here, then on the GPU box:
#   for n in 0 1 2 4 8 16; do SAPR_LIB=$PWD/sapr_amd/libsapr_hip_abl$n.so python scripts/time_mfcc_only.py; done
set -e
cd "$(dirname "$0")/../.."
FLAGS="-O3 -std=c++17 -ftemplate-depth=2048 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DSAPR_ABLATE=$n -c sapr_amd/csrc/mfcc.hip -o /tmp/mfcc_abl$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC sapr_amd/csrc/common.o sapr_amd/csrc/viterbi.o sapr_amd/csrc/estep.o \
     sapr_amd/csrc/custom.o sapr_amd/csrc/resample.o /tmp/mfcc_abl$n.o -o sapr_amd/libsapr_hip_abl$n.so
done
ls -la sapr_amd/libsapr_hip_abl*.so
