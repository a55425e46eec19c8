#!/bin/bash
# x6 conv iteration: parity of the x6 kernels -> re-tune the x6 kinds (others keep their entries) -> bench -> layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/it
KINDS=${KINDS:-fwd6,dgrad6}
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "x6 or fused" > gpurun_out/it/tests.log 2>&1; echo "rc=$?" >> gpurun_out/it/tests.log; tail -3 gpurun_out/it/tests.log
timeout 700 python tools/autotune.py 288 $KINDS > gpurun_out/it/autotune.log 2>&1; echo "rc=$?" >> gpurun_out/it/autotune.log; tail -2 gpurun_out/it/autotune.log
timeout 600 python bench.py --cpu-baseline-videos 0 > gpurun_out/it/bench.log 2>&1; echo "rc=$?" >> gpurun_out/it/bench.log; tail -2 gpurun_out/it/bench.log | cut -c1-3500
timeout 300 python tools/layer_table.py > gpurun_out/it/layers.txt 2>&1; tail -1 gpurun_out/it/layers.txt
