#!/bin/bash
# Tooling: the product library with conv_pl.hip / wgrad_pl.hip compiled under -DPL_ABLATE (runtime ablation switches of the
# planes kernels, tools/ablate_conv_pl.py), linked against the already built objects of the other sources.
# Output: tools/.trace/libssn_hip_ablate.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
python -c "import sys; sys.path.insert(0, '$R'); import action_detection_amd as p; p.build()" 2>&1 | grep -v "occupancy\|warnings gen" || true
mkdir -p "$R/tools/.trace"
C="$R/action-detection_amd/csrc"
EXCL=""
for f in conv_pl wgrad_pl; do
  [ -f "$C/$f.hip" ] || continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPL_ABLATE -c "$C/$f.hip" -o "$R/tools/.trace/$f.ablate.o"
  EXCL="$EXCL|/$f.hip.o"
done
OBJS=$(ls "$C"/*.hip.o | grep -v -E "${EXCL#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/.trace/libssn_hip_ablate.so" $OBJS "$R"/tools/.trace/*.ablate.o
echo built
