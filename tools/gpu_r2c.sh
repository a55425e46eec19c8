#!/bin/bash
# round-2, f16-split kernels: GPU test tier, K-sweep table, autotune of the split kernels' tiles, bench (graph) with the
# old and the new tile table, rocprofv3 kernel stats of the graph-replay and the eager single-stream run, layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2c; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "error_growth" -s > $O/ksweep.log 2>&1; grep "^K=" $O/ksweep.log | tail -12
timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench_oldtiles.json 2> $O/bench_oldtiles.err; echo "bench(old tiles) rc=$?"; cut -c1-260 $O/bench_oldtiles.json
timeout 900 python tools/autotune.py 288 fwd6,dgrad6,wgrad6 > $O/autotune.log 2>&1; echo "autotune rc=$?"; tail -3 $O/autotune.log
cp action-detection_amd/tuned_tiles.json $O/tuned_tiles.json
timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events > $R/$O/prof_graph.log 2>&1; echo "prof graph rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events --no-graph --single-stream > $R/$O/prof_eager.log 2>&1; echo "prof eager rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; tail -3 $O/layer_table.txt
du -sh $O
