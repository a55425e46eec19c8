#!/bin/bash
# x6 kernels: parity on the GPU, phase trace, then tile autotune of the fwd6/dgrad6 kinds
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "x6" > gpurun_out/x6_tests.log 2>&1; echo "rc=$?" >> gpurun_out/x6_tests.log; tail -5 gpurun_out/x6_tests.log
timeout 600 python tools/trace_x6.py > gpurun_out/trace_x6.log 2>&1; grep -v distinct gpurun_out/trace_x6.log | cut -c1-250
timeout 900 python tools/autotune.py 288 fwd6,dgrad6 > gpurun_out/autotune_x6.log 2>&1; echo "rc=$?" >> gpurun_out/autotune_x6.log; tail -3 gpurun_out/autotune_x6.log
