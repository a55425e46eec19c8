#!/bin/bash
# round 6, call H: fma_mix low-plane split + one-launch objective + label select + mask audit + non-finite passes: GPU parity tier
# (planes, model incl. mask audit, golden, scale guard), smoke(), then A/B against the previous build (alternating runs).
O=gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/test_planes.py tests/test_model_gpu.py tests/test_golden.py tests/test_scale_guard.py -x -q -m gpu -s > $O/h_tests.txt 2>&1; grep -E "mask audit|passed|failed|Error" $O/h_tests.txt | tail -8
timeout 300 python __graft_entry__.py smoke > $O/h_smoke.txt 2>&1; tail -2 $O/h_smoke.txt
O=$O REPS="1 2 3" BENCH_ARGS="--no-secondary" bash tools/gpu_ab_lib.sh 2>&1 | tee $O/h_ab.txt
echo "H: done at ${SECONDS}s"
