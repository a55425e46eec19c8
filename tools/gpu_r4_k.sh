#!/bin/bash
# Round 4, call K: rocprofv3 kernel stats of the eager single-stream step + the PMC passes (tools/gpu_pmc.sh)
O=gpurun_out/r4k; mkdir -p $O
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 10 --warmup 3 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_eager.csv; head -14 "$f" | cut -c1-150
bash tools/gpu_pmc.sh; cp gpurun_out/pmc/summary.json $O/pmc_summary_planes.json
