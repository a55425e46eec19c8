#!/bin/bash
# full autotune (f32 + x6 kinds, with times) -> GPU test tier -> bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python tools/autotune.py 288 > gpurun_out/autotune_all.log 2>&1; echo "rc=$?" >> gpurun_out/autotune_all.log; tail -2 gpurun_out/autotune_all.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-videos 0 > gpurun_out/bench_quick.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_quick.log; tail -2 gpurun_out/bench_quick.log | cut -c1-1800
