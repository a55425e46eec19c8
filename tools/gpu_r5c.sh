#!/bin/bash
# Round 5, call C: pool kernels with 32-bit fast-division indexing (was: 64-bit div / mod per element) -- parity, then the step and the
# per-kernel times of an eager run (rocprofv3 --kernel-trace --stats).
O=gpurun_out/r5; mkdir -p $O
R=$(pwd)
STAGES=${STAGES:-test,bench,prof}
stage_test() { timeout 900 python -m pytest tests/test_planes.py tests/test_planes_bn_exec.py -m gpu -q -x --durations=5 > $O/c_planes_tests.log 2>&1; tail -6 $O/c_planes_tests.log; }
stage_bench() { for rep in 1 2; do timeout 300 python bench.py --cpu-baseline-videos 0 > $O/c_bench_$rep.json 2> $O/c_bench_$rep.err; cut -c1-200 $O/c_bench_$rep.json; done; }
stage_prof() {
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/c_prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 30 --warmup 3 > $R/$O/c_prof.log 2>&1
  cd $R; find $O/c_prof -name "*kernel_trace.csv" -delete; find $O/c_prof -name "*.db" -delete; f=$(find $O/c_prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/c_kernel_stats_eager.csv; head -40 "$f" | cut -c1-150
}
for st in ${STAGES//,/ }; do echo "== $st $(date +%T)"; cd $R; stage_$st; done
