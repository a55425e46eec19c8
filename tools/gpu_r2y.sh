#!/bin/bash
# tile table for the rectangular-tap launches of the Inception-v3 plan, then its bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2y; mkdir -p $O; export TMPDIR=/tmp
AUTOTUNE_RECT_ONLY=1 timeout 60 python tools/autotune.py 144 fwd6 InceptionV3 > $O/autotune_v3_rect.log 2>&1; echo "autotune rc=$?"; tail -2 $O/autotune_v3_rect.log
timeout 40 python tools/bench_train_v3.py > $O/bench_train_v3.json 2> $O/err.txt; echo "v3 bench rc=$?"; cut -c1-200 $O/bench_train_v3.json
timeout 40 python tools/bench_train_v3.py --videos 4 > $O/bench_train_v3_v4.json 2>> $O/err.txt; cut -c1-200 $O/bench_train_v3_v4.json
timeout 40 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; cut -c1-200 $O/dense_test_v3.json
