#!/bin/bash
# Round 5, call E: (1) pool lab: what bounds the stem pools' backward (tools/pool_lab); (2) two-lane branch schedule A/B + parity.
O=gpurun_out/r5; mkdir -p $O
R=$(pwd)
STAGES=${STAGES:-lab,lanes}
stage_lab() { timeout 300 tools/pool_lab/pool_lab > $O/e_pool_lab.txt 2>&1; cat $O/e_pool_lab.txt; }
stage_lanes() {
  SSN_BRANCH_LANES=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "fwd_bwd_matches_oracle and RGB or full_batch" > $O/e_lanes_tests.log 2>&1; tail -4 $O/e_lanes_tests.log
  for rep in 1 2 3; do
    for l in 0 1; do
      SSN_BRANCH_LANES=$l timeout 300 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/e_bench_lanes${l}_$rep.json 2> $O/e_bench_lanes${l}_$rep.err
      echo "lanes=$l #$rep $(cut -c1-175 $O/e_bench_lanes${l}_$rep.json)"; grep -i "error\|Traceback" $O/e_bench_lanes${l}_$rep.err | head -3
    done
  done
}
for st in ${STAGES//,/ }; do echo "== $st $(date +%T)"; cd $R; stage_$st; done
