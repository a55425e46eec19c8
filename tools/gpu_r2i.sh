#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2i; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/autotune.py 288 fwd6,dgrad6,wgrad6 > $O/autotune.log 2>&1; echo "autotune rc=$?"; grep -c "1x1+" $O/autotune.log
cp action-detection_amd/tuned_tiles.json $O/tuned_tiles.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
timeout 600 python tools/layer_table.py > $O/layer_table.txt 2>&1; tail -2 $O/layer_table.txt
