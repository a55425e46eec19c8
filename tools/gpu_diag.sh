#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/diag; mkdir -p $O
timeout 300 python tools/diag_wgrad.py > $O/wgrad.log 2>&1; tail -12 $O/wgrad.log
timeout 600 python tools/diag_grad.py > $O/grad.log 2>&1; tail -50 $O/grad.log
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
