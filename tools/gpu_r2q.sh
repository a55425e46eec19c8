#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2q; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_inceptionv3.py -q -m gpu -s -k backbone_backward > $O/gpu_tests_v3.log 2>&1; echo "rc=$?" >> $O/gpu_tests_v3.log; grep -n "median\|passed\|failed\|rc=" $O/gpu_tests_v3.log | head
timeout 600 python tools/bench_train_v3.py > $O/bench_train_v3.json 2> $O/err.txt; echo "v3 bench rc=$?"; cut -c1-400 $O/bench_train_v3.json
