#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2e; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -q -m gpu -k "bn_modes or binary or rgbdiff or reorg or frame_diff" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
timeout 600 python tools/trace_x6.py > $O/trace_x6.log 2>&1; echo "trace rc=$?"; grep -v "^\[" $O/trace_x6.log | tail -40
