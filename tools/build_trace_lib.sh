#!/bin/bash
# Tooling: the product library with conv_x6.hip compiled under -DX6_PHASE_TRACE (per-phase cycle counters),
# linked against the already built objects of the other sources.  Output: tools/.trace/libssn_hip_trace.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
python -c "import sys; sys.path.insert(0, '$R'); import action_detection_amd as p; p.build()" 2>&1 | grep -v "occupancy\|warnings gen" || true
mkdir -p "$R/tools/.trace"
C="$R/action-detection_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX6_PHASE_TRACE -c "$C/conv_x6.hip" -o "$R/tools/.trace/conv_x6.o" 2>&1 | grep -v "occupancy\|warnings gen" || true
OBJS=$(ls "$C"/*.hip.o | grep -v "/conv_x6.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/.trace/libssn_hip_trace.so" $OBJS "$R/tools/.trace/conv_x6.o"
echo built
