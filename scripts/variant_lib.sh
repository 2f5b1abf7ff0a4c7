#!/bin/bash
# Developer tool: libsapr_hip_<name>.so = the in-tree objects with the named translation units rebuilt under extra
# flags (timing variants side by side on one GPU box: SAPR_LIB=$PWD/sapr_amd/libsapr_hip_<name>.so python scripts/...).
#   scripts/variant_lib.sh <name> "<extra flags>" mfcc.hip [more.hip ...]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
FLAGS="-O3 -std=c++17 -ftemplate-depth=2048 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
objs=$(ls sapr_amd/csrc/*.o)
for src in "$@"; do
  o=/tmp/variant_${name}_${src%.hip}.o
  /opt/rocm/bin/hipcc $FLAGS $flags -c sapr_amd/csrc/$src -o $o &
  objs=$(echo "$objs" | grep -v "/${src%.hip}.o$")
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o sapr_amd/libsapr_hip_$name.so
ls -la sapr_amd/libsapr_hip_$name.so
