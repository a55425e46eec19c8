#!/bin/bash
# what the driver runs at round end, in its order: GPU test tier, smoke(), default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/drv
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/drv/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/drv/gpu_tests.log; tail -3 gpurun_out/drv/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/drv/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/drv/smoke.log; tail -2 gpurun_out/drv/smoke.log
timeout 900 python bench.py > gpurun_out/drv/bench.log 2>&1; echo "rc=$?" >> gpurun_out/drv/bench.log; tail -2 gpurun_out/drv/bench.log | cut -c1-250
