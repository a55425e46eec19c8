#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 900 python tools/autotune.py > gpurun_out/autotune.log 2>&1; echo "rc=$?" >> gpurun_out/autotune.log; tail -3 gpurun_out/autotune.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-videos 0 > gpurun_out/bench_quick.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_quick.log; tail -2 gpurun_out/bench_quick.log | cut -c1-2500
