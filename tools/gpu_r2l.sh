#!/bin/bash
# round-2 closing run, in the driver's order: full GPU tier (-x), smoke(), default bench; then the other configurations
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2l; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-250 $O/bench.json
timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 --no-kernel-events > $O/bench_flow.json 2> $O/bench_flow.err; echo "flow rc=$?"; cut -c1-200 $O/bench_flow.json
timeout 600 python bench.py --precision f32 --cpu-baseline-videos 0 --no-kernel-events > $O/bench_f32.json 2> $O/bench_f32.err; echo "f32 rc=$?"; cut -c1-200 $O/bench_f32.json
timeout 600 python tools/bench_dense_test.py > $O/dense_test.json 2> $O/dense_test.err; echo "dense rc=$?"; cut -c1-300 $O/dense_test.json
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; echo "dense v3 rc=$?"; cut -c1-300 $O/dense_test_v3.json
