#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2o; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/bench_train_v3.py --families --layers 70 > $O/bench_train_v3.json 2> $O/layers.txt; echo "v3 bench rc=$?"; cut -c1-300 $O/bench_train_v3.json
timeout 900 python -m pytest tests/test_inceptionv3.py -q -m gpu -s > $O/gpu_tests_v3.log 2>&1; echo "rc=$?" >> $O/gpu_tests_v3.log; grep -n "median\|passed\|failed\|rc=" $O/gpu_tests_v3.log | head
