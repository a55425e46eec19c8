#!/bin/bash
# round-2: full GPU test tier (bn_mode partial / full, RGBDiff, config-2-size parity), phase trace of the split conv kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2d; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -8 $O/gpu_tests.log
timeout 600 python tools/trace_x6.py > $O/trace_x6.log 2>&1; echo "trace rc=$?"; grep -v "^\[" $O/trace_x6.log | tail -40
timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
