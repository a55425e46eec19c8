#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/v3
timeout 900 python -m pytest tests/test_inceptionv3.py tests/test_kernels.py -x -q -m gpu -k "inceptionv3 or rect" > gpurun_out/v3/tests.log 2>&1; echo "rc=$?" >> gpurun_out/v3/tests.log; tail -5 gpurun_out/v3/tests.log
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 --cpu-ticks 4 > gpurun_out/v3/bench.log 2>&1; echo "rc=$?" >> gpurun_out/v3/bench.log; tail -2 gpurun_out/v3/bench.log | cut -c1-1200
