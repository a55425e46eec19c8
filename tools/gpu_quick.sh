#!/bin/bash
# quick GPU iteration: kernel parity tests + one bench line (no CPU baseline, no profile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q --tb=short --no-header -p no:cacheprovider -x > gpurun_out/pytest_quick.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_quick.log; tail -6 gpurun_out/pytest_quick.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-videos 0 ${BENCH_ARGS} > gpurun_out/bench_quick.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_quick.log; tail -2 gpurun_out/bench_quick.log | cut -c1-2500
