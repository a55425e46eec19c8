#!/bin/bash
# Build tools/.ab/libssn_prev.so from the csrc/ of a git revision (default HEAD): the PREV arm of tools/gpu_ab_lib.sh.
# Only the listed translation units are recompiled from the old revision; the others link from the in-tree objects.
set -e
REV=${1:-HEAD}; shift || true
UNITS=${@:-conv_pl.hip}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
D=$(mktemp -d)
mkdir -p $HERE/tools/.ab
git -C $HERE archive $REV action-detection_amd/csrc | tar -x -C $D
OBJS=()
for f in $HERE/action-detection_amd/csrc/*.hip; do
  b=$(basename $f)
  if [[ " $UNITS " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $D/action-detection_amd/csrc/$b -o $D/$b.o &
    OBJS+=($D/$b.o)
  else
    OBJS+=($f.o)
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/tools/.ab/libssn_prev.so "${OBJS[@]}"
rm -rf $D
echo "built tools/.ab/libssn_prev.so from $REV ($UNITS)"
