#!/bin/bash
# kernel parity tests -> autotune -> bench (quick iteration loop on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q --tb=short --no-header -p no:cacheprovider -x > gpurun_out/pytest_quick.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_quick.log; tail -4 gpurun_out/pytest_quick.log
if [ "${AUTOTUNE:-1}" = "1" ]; then
  timeout 900 python tools/autotune.py > gpurun_out/autotune.log 2>&1; echo "rc=$?" >> gpurun_out/autotune.log; tail -2 gpurun_out/autotune.log
fi
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-videos 0 > gpurun_out/bench_quick.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_quick.log; tail -2 gpurun_out/bench_quick.log | cut -c1-2500
timeout 300 python tools/layer_table.py > gpurun_out/layers.txt 2>&1
