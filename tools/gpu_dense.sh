#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/dt
timeout 900 python -m pytest tests/test_dense_test.py -x -q -m gpu > gpurun_out/dt/tests.log 2>&1; echo "rc=$?" >> gpurun_out/dt/tests.log; tail -3 gpurun_out/dt/tests.log
timeout 600 python tools/bench_dense_test.py > gpurun_out/dt/bench.log 2>&1; echo "rc=$?" >> gpurun_out/dt/bench.log; tail -2 gpurun_out/dt/bench.log | cut -c1-1200
