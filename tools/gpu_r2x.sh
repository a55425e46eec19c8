#!/bin/bash
# which of the stream-level overlaps pay under graph replay: wgrad side stream x forward branch lanes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2x; mkdir -p $O; export TMPDIR=/tmp
for cfg in "1 1" "1 0" "0 1" "0 0" "1 1"; do
  set -- $cfg
  SSN_OVERLAP_WGRAD=$1 SSN_BRANCH_STREAMS=$2 timeout 120 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_w$1_b$2.json 2>> $O/bench.err
  echo "wgrad-stream=$1 branch-lanes=$2: $(cut -c60-140 $O/bench_w$1_b$2.json)"
done
