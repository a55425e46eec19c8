#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -q -m gpu -k "not fwd_bwd_matches_oracle" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
