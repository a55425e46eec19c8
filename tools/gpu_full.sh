#!/bin/bash
# full GPU parity suite + bench (no rocprof)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-videos 0 > gpurun_out/bench_quick.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_quick.log; tail -2 gpurun_out/bench_quick.log | cut -c1-1200
