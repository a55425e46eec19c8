#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2j; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -q -m gpu -k "not fwd_bwd_matches_oracle" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
timeout 900 python tools/autotune.py 288 fwd6s2d,wgrad6s2d > $O/autotune.log 2>&1; echo "autotune rc=$?"; grep "conv1" $O/autotune.log
cp action-detection_amd/tuned_tiles.json $O/tuned_tiles.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
SSN_STEM_S2D=0 timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_nos2d.json 2> $O/bench_nos2d.err; echo "bench(no s2d) rc=$?"; cut -c1-200 $O/bench_nos2d.json
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; head -4 $O/layer_table.txt | cut -c1-36,58-140; tail -1 $O/layer_table.txt
