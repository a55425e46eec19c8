#!/bin/bash
# first GPU contact: tests (not fail-fast), smoke, bench, rocprof stats
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/hw.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
