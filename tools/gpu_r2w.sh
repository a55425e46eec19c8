#!/bin/bash
# the 7x7-stage layers take the 16-byte-load path now: re-tune their tiles, bench before / after
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2w; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_before.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-170 $O/bench_before.json
cp action-detection_amd/tuned_tiles.json $O/tuned_before.json
AUTOTUNE_HIN=7 timeout 200 python tools/autotune.py 288 fwd6,dgrad6 > $O/autotune7.log 2>&1; echo "autotune rc=$?"; tail -2 $O/autotune7.log
timeout 120 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_after.json 2>> $O/bench.err; echo "bench rc=$?"; cut -c1-170 $O/bench_after.json
timeout 120 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_after2.json 2>> $O/bench.err; cut -c1-170 $O/bench_after2.json
