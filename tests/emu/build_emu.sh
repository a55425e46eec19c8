#!/bin/bash
# TEST INFRASTRUCTURE ONLY: build the product's .hip sources for the host through the HIP emulator
# header so CPU-only tests can drive the very same C ABI (libssn_emu.so) on host memory.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../action-detection_amd/csrc"
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT="${EMU_OUT:-$HERE/libssn_emu.so}"      # EMU_OUT / EMU_OBJ: build into another place (a second build while tests run)
OBJ="${EMU_OBJ:-$HERE}"
OBJS=()
for f in "$SRC"/*.hip "$HERE/emu_globals.cpp"; do
  o="$OBJ/.obj_$(basename "$f").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$o" ] || [ "$SRC/ssn_common.h" -nt "$o" ] || [ "$SRC/conv_epilogue.h" -nt "$o" ] || [ "$SRC/conv_x6_kernel.h" -nt "$o" ] || [ "$SRC/planes.h" -nt "$o" ] || [ "$SRC/conv_pl_epilogue.inc" -nt "$o" ]; then
    rm -f "$o"      # (a failed compile must not leave the previous object to be linked)
    "$CXX" -x c++ -std=c++17 -O1 -fPIC -w -I "$HERE" -I "$SRC" -c "$f" -o "$o" &
  fi
  OBJS+=("$o")
done
wait
for o in "${OBJS[@]}"; do [ -f "$o" ] || { echo "emulator build FAILED: $o missing" >&2; exit 1; }; done
"$CXX" -shared -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
