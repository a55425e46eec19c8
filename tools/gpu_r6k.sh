#!/bin/bash
# round 6, call K: the driver's sequence on the closing tree -- GPU tier, smoke(), the bench command.
O=gpurun_out/r6k; mkdir -p $O
timeout 3000 python -m pytest tests/ -m gpu -q -x --durations=10 > $O/gpu_tests.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; tail -2 $O/bench.err
echo "K: done at ${SECONDS}s"
