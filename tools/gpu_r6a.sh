#!/bin/bash
# round 6, call A: (1) the default bench line of the tree as round 5 left it (this box's baseline), (2) cold-vs-hot operands for the
# forward launches that run 30-35 % longer inside the step than in the autotuner's loop (tools/bench_conv_cold.py).
O=gpurun_out/r6; mkdir -p $O
timeout 300 python bench.py --cpu-baseline-videos 0 > $O/a_bench.json 2> $O/a_bench.err; cut -c1-300 $O/a_bench.json
timeout 600 python tools/bench_conv_cold.py 288 > $O/a_cold.txt 2> $O/a_cold.err; cat $O/a_cold.txt | cut -c1-400
echo "A: done at ${SECONDS}s"
