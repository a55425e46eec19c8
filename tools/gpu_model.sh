#!/bin/bash
# model-level check: SSN GPU tests (+ selected kernel tests via KTESTS) + bench (no autotune)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/md
if [ -n "$KTESTS" ]; then timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "$KTESTS" > gpurun_out/md/ktests.log 2>&1; echo "rc=$?" >> gpurun_out/md/ktests.log; tail -3 gpurun_out/md/ktests.log; fi
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/md/tests.log 2>&1; echo "rc=$?" >> gpurun_out/md/tests.log; tail -3 gpurun_out/md/tests.log
timeout 600 python bench.py --cpu-baseline-videos 0 > gpurun_out/md/bench.log 2>&1; echo "rc=$?" >> gpurun_out/md/bench.log; tail -2 gpurun_out/md/bench.log | cut -c1-400
grep -o '"conv_dgrad_[a-z0-9]*": {[^}]*}' gpurun_out/md/bench.log
