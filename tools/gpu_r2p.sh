#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2p; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -k "rect_backward or pad0 or conv_x6 or wgrad" > $O/gpu_tests_k.log 2>&1; echo "rc=$?" >> $O/gpu_tests_k.log; tail -3 $O/gpu_tests_k.log
timeout 900 python -m pytest tests/test_inceptionv3.py -q -m gpu -s > $O/gpu_tests_v3.log 2>&1; echo "rc=$?" >> $O/gpu_tests_v3.log; grep -n "median\|passed\|failed\|rc=" $O/gpu_tests_v3.log | head
timeout 600 python tools/bench_train_v3.py --families --layers 40 > $O/bench_train_v3.json 2> $O/layers.txt; echo "v3 bench rc=$?"; cut -c1-1200 $O/bench_train_v3.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
