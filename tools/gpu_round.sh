#!/bin/bash
# one GPU call: autotune (f32 + x6 kinds, with times) -> GPU test tier -> bench line -> per-layer table -> rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/rd; export TMPDIR=/tmp
if [ "${AUTOTUNE:-1}" = "1" ]; then
  timeout 700 python tools/autotune.py 288 > gpurun_out/rd/autotune.log 2>&1; echo "rc=$?" >> gpurun_out/rd/autotune.log; tail -2 gpurun_out/rd/autotune.log
fi
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/rd/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/rd/gpu_tests.log; tail -4 gpurun_out/rd/gpu_tests.log
timeout 600 python bench.py > gpurun_out/rd/bench.log 2>&1; echo "rc=$?" >> gpurun_out/rd/bench.log; tail -2 gpurun_out/rd/bench.log | cut -c1-3000
timeout 300 python tools/layer_table.py > gpurun_out/rd/layers.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/rd/prof" -o run -- python "$R/bench.py" --cpu-baseline-videos 0 > "$R/gpurun_out/rd/prof.log" 2>&1
cd "$R"; find gpurun_out/rd -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/rd -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-160
du -sh gpurun_out/rd
