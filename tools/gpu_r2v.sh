#!/bin/bash
# tile table for the square-tap launches of the Inception-v3 plan (added to tuned_tiles.json), then its bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2v; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python tools/autotune.py 144 fwd6,dgrad6,wgrad6 InceptionV3 > $O/autotune_v3.log 2>&1; echo "autotune rc=$?"; tail -3 $O/autotune_v3.log
timeout 200 python tools/bench_train_v3.py --families > $O/bench_train_v3.json 2> $O/err.txt; echo "v3 bench rc=$?"; cut -c1-900 $O/bench_train_v3.json
timeout 200 python tools/bench_train_v3.py --videos 4 > $O/bench_train_v3_v4.json 2>> $O/err.txt; echo "v3 bench V=4 rc=$?"; cut -c1-200 $O/bench_train_v3_v4.json
timeout 200 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 > $O/dense_test_v3.json 2> $O/dense_test_v3.err; echo "v3 dense rc=$?"; cut -c1-200 $O/dense_test_v3.json
