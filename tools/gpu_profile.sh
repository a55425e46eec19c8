#!/bin/bash
# GPU-box script: parity tests, bench line, rocprofv3 kernel-trace stats of the same bench command.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log
fi
timeout 600 python bench.py ${BENCH_ARGS:---steps 10 --warmup 3} > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
if [ "${SKIP_PROF:-0}" != "1" ]; then
  rm -rf gpurun_out/prof; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o run -- python "$R/bench.py" --steps 5 --warmup 2 --cpu-baseline-videos 0 > "$R/gpurun_out/prof.log" 2>&1
  echo "rocprof rc=$?" >> "$R/gpurun_out/prof.log"; cd "$R"
  find gpurun_out/prof -name "*stats*.csv" | head; tail -3 gpurun_out/prof.log
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
  # keep the merge small: drop the raw trace if it is large
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
