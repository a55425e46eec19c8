#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2m; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/diag_determinism.py 24 > $O/determinism.log 2>&1; tail -4 $O/determinism.log
timeout 900 python -m pytest tests/ -x -q -m gpu -k "not fwd_bwd_matches_oracle" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
