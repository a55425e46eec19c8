#!/bin/bash
# round 6, call B: the 16-byte epilogue (v_permlane32_swap pairs, conv_pl_epilogue.inc) -- parity of the conv kernels on the GPU, the
# new two-lane / per-layer-wgrad test, then A/B against the previous build of conv_pl.hip on this box (alternating runs).
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_planes.py tests/test_model_gpu.py -x -q -m gpu -k "conv_pl or per_layer_wgrad or fwd_bwd_matches" > $O/b_tests.txt 2>&1; tail -5 $O/b_tests.txt
O=$O REPS="1 2 3" bash tools/gpu_ab_lib.sh 2>&1 | tee $O/b_ab.txt
echo "B: done at ${SECONDS}s"
