#!/bin/bash
# round 6, call D: 16-byte epilogue with the store-data hazard closed (planes.h: pl_store_b128): sporadic-corruption check at the bench
# batch (every tile, zero-filled destination: a lost low-plane dword shows as 1e-4), GPU parity tests, then the A/B against round 5's conv_pl.
O=gpurun_out/r6; mkdir -p $O
for rep in 1 2 3; do timeout 200 python tools/diag_epilogue.py 288 2>&1 | grep -v "e-07" | grep -v amdgpu.ids; done > $O/d_diag.txt; echo "diag lines with errors above 1e-6: $(grep -c tile $O/d_diag.txt)"; head -5 $O/d_diag.txt
timeout 1200 python -m pytest tests/test_planes.py tests/test_model_gpu.py tests/test_kernels.py -x -q -m gpu > $O/d_tests.txt 2>&1; tail -4 $O/d_tests.txt
O=$O REPS="1 2 3" bash tools/gpu_ab_lib.sh 2>&1 | tee $O/d_ab.txt
echo "D: done at ${SECONDS}s"
