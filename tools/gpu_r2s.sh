#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2s; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_inceptionv3.py -q -m gpu -k "pack_rect or inceptionv3 or rect_backward" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 600 python tools/bench_train_v3.py --families > $O/bench_train_v3.json 2> $O/err.txt; echo "v3 bench rc=$?"; cut -c1-900 $O/bench_train_v3.json
timeout 600 python tools/bench_train_v3.py --videos 4 > $O/bench_train_v3_v4.json 2>> $O/err.txt; echo "v3 bench V=4 rc=$?"; cut -c1-300 $O/bench_train_v3_v4.json
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; echo "v3 dense rc=$?"; cut -c1-200 $O/dense_test_v3.json
