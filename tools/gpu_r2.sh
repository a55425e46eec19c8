#!/bin/bash
# round-2 measurement call: GPU test tier, K-sweep table, bench (graph, pool order A/B, eager single-stream), two-rank
# self-launch control flow, rocprofv3 kernel stats of the graph-replay AND the eager single-stream run, layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k error_growth -s > $O/ksweep.log 2>&1; grep "^K=" $O/ksweep.log | tail -12
timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
SSN_POOL_ORDER=manifest timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_pool_manifest.json 2> $O/bench_pool_manifest.err; cut -c1-200 $O/bench_pool_manifest.json
SSN_BENCH_ONE_DEVICE=1 SSN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-graph --cpu-baseline-videos 0 --no-kernel-events > $O/bench_selflaunch2.json 2> $O/bench_selflaunch2.err; echo "selflaunch rc=$?"; cut -c1-300 $O/bench_selflaunch2.json; tail -3 $O/bench_selflaunch2.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events > $R/$O/prof_graph.log 2>&1; echo "prof graph rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events --no-graph --single-stream > $R/$O/prof_eager.log 2>&1; echo "prof eager rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; tail -3 $O/layer_table.txt
du -sh $O
