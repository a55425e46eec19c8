#!/bin/bash
# closing evidence after the last kernel change (16-byte loads on odd-sized planes): full GPU tier, smoke, the default bench
# line, rocprofv3 kernel stats (graph replay + eager single stream), layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2u; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events > $R/$O/prof_graph.log 2>&1; echo "prof graph rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events --no-graph --single-stream > $R/$O/prof_eager.log 2>&1; echo "prof eager rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; tail -2 $O/layer_table.txt
