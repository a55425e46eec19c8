#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
L=action-detection_amd/libssn_hip.so
echo "== new lib: model"; SSN_BRANCH_LANES=0 timeout 200 python tools/diag_model_fault.py 288 2>&1 | tail -8
cp $L /tmp/new.so; cp tools/.ab/libssn_prev.so $L
echo "== prev lib: tiles"; timeout 200 python tools/diag_epilogue.py 288 2>&1 | grep -E "tile  7|tile 11"
cp /tmp/new.so $L
