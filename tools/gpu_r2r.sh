#!/bin/bash
# round-2 closing evidence: full GPU tier, bench (default flags, as the driver runs it), self-launch with 2 ranks on one device,
# rocprofv3 kernel stats (graph replay + eager single stream), PMC passes, layer table, Inception-v3 training / dense-test lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2r; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
SSN_BENCH_ONE_DEVICE=1 SSN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-graph --cpu-baseline-videos 0 --no-kernel-events > $O/bench_selflaunch2.json 2> $O/bench_selflaunch2.err; echo "selflaunch rc=$?"; cut -c1-200 $O/bench_selflaunch2.json
timeout 600 python tools/bench_train_v3.py --families --layers 40 > $O/bench_train_v3.json 2> $O/v3_layers.txt; echo "v3 train rc=$?"; cut -c1-300 $O/bench_train_v3.json
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; echo "v3 dense rc=$?"; cut -c1-300 $O/dense_test_v3.json
timeout 600 python tools/bench_dense_test.py > $O/dense_test.json 2> $O/dense_test.err; echo "dense rc=$?"; cut -c1-300 $O/dense_test.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events > $R/$O/prof_graph.log 2>&1; echo "prof graph rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events --no-graph --single-stream > $R/$O/prof_eager.log 2>&1; echo "prof eager rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v3 -o p -- python $R/tools/bench_train_v3.py > $R/$O/prof_v3.log 2>&1; echo "prof v3 rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log; cp gpurun_out/pmc/summary.json $O/pmc_summary.json 2>/dev/null
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; tail -2 $O/layer_table.txt
du -sh $O
