#!/bin/bash
# x6 kernels: parity on the GPU, then tile autotune of the fwd6/dgrad6 kinds next to the f32 kinds
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "x6" > gpurun_out/x6_tests.log 2>&1; echo "rc=$?" >> gpurun_out/x6_tests.log; tail -5 gpurun_out/x6_tests.log
timeout 900 python tools/autotune.py 288 fwd,dgrad,fwd6,dgrad6 > gpurun_out/autotune_x6.log 2>&1; echo "rc=$?" >> gpurun_out/autotune_x6.log; tail -3 gpurun_out/autotune_x6.log
