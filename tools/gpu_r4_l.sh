#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
timeout 600 python -m pytest tests/test_planes.py -m gpu -q -k pools > $O/tests.log 2>&1; tail -2 $O/tests.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 60 --warmup 3 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_eager.csv; grep -i "pool\|gap" "$f" | cut -c1-140
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | cut -c1-200
