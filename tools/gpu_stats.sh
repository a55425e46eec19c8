#!/bin/bash
# kernel-level time distribution of the default bench (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/st; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/st/prof" -o run -- python "$R/bench.py" --steps 10 --warmup 3 --cpu-baseline-videos 0 --no-kernel-events > "$R/gpurun_out/st/prof.log" 2>&1
cd "$R"; find gpurun_out/st -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/st -name "*kernel_stats.csv" | head -1); echo $f; head -40 "$f" | cut -c1-200
tail -1 gpurun_out/st/prof.log | cut -c1-300
