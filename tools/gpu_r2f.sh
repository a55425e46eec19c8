#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2f; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/diag_bn_full.py > $O/diag_bn_full.log 2>&1; grep "bn_mode" $O/diag_bn_full.log
timeout 900 python -m pytest tests/ -q -m gpu -k "not fwd_bwd_matches_oracle" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
timeout 900 python tools/autotune.py 288 fwd6,dgrad6,wgrad6 > $O/autotune.log 2>&1; echo "autotune rc=$?"; tail -2 $O/autotune.log
cp action-detection_amd/tuned_tiles.json $O/tuned_tiles.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_merge.json 2> $O/bench_merge.err; echo "bench(merge) rc=$?"; cut -c1-200 $O/bench_merge.json
SSN_MERGE_PROJ=0 timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_nomerge.json 2> $O/bench_nomerge.err; echo "bench(no merge) rc=$?"; cut -c1-200 $O/bench_nomerge.json
