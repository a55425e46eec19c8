#!/bin/bash
# Round 5, call A: measure what round 4 left written-but-OFF (VERDICT r4 "next" #2), on ONE box, before / after:
#   base    bench.py with the committed tile table (this box's baseline)
#   exp     the experimental planes tests (haloed per-image 3x3 tiles) on the GPU
#   tune    SPLITS=1 tools/autotune_pl.py (fwd, dgrad; candidates incl. the haloed 32+c / per-image 48+c tiles and bulk+tail splits)
#   after   bench.py with the re-tuned table
O=gpurun_out/r5; mkdir -p $O
R=$(pwd)
STAGES=${STAGES:-base,exp,tune,after}
stage_base() { timeout 400 python bench.py --cpu-baseline-videos 0 > $O/a_bench_base.json 2> $O/a_bench_base.err; cut -c1-400 $O/a_bench_base.json; tail -2 $O/a_bench_base.err; }
stage_exp() { SSN_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_planes.py -m gpu -q -x --durations=8 > $O/a_exp_tests.log 2>&1; tail -15 $O/a_exp_tests.log; }
stage_tune() {
  cp action-detection_amd/tuned_tiles_pl.json $O/a_tiles_before.json
  SPLITS=1 KINDS=${KINDS:-fwd,dgrad} timeout 700 python tools/autotune_pl.py 288 BNInception > $O/a_autotune_splits.txt 2>&1
  tail -70 $O/a_autotune_splits.txt | cut -c1-230
  cp action-detection_amd/tuned_tiles_pl.json $O/a_tiles_after.json
}
stage_after() { timeout 400 python bench.py --cpu-baseline-videos 0 > $O/a_bench_after.json 2> $O/a_bench_after.err; cut -c1-400 $O/a_bench_after.json; tail -2 $O/a_bench_after.err; }
for st in ${STAGES//,/ }; do echo "== $st $(date +%T)"; cd $R; stage_$st; done
