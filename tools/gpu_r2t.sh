#!/bin/bash
# 16-byte activation loads on planes that are not a multiple of 4 pixels (padded enumeration): conv tests, determinism,
# Inception-v3 lines, headline bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2t; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_inceptionv3.py -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 600 python tools/bench_train_v3.py --families --layers 30 > $O/bench_train_v3.json 2> $O/v3_layers.txt; echo "v3 bench rc=$?"; cut -c1-900 $O/bench_train_v3.json
timeout 600 python tools/bench_train_v3.py --videos 4 > $O/bench_train_v3_v4.json 2>> $O/err.txt; echo "v3 bench V=4 rc=$?"; cut -c1-200 $O/bench_train_v3_v4.json
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; echo "v3 dense rc=$?"; cut -c1-200 $O/dense_test_v3.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
timeout 600 python tools/diag_determinism.py 12 > $O/determinism.log 2>&1; tail -2 $O/determinism.log
