#!/bin/bash
# Round 5, call B: grouped weight gradients (ssn_conv_wgrad_pl_group) -- parity on the GPU, then the step with per-layer launches
# (SSN_GROUP_WGRAD=0) against grouped ones with several planner constants, same box, alternating.
O=gpurun_out/r5; mkdir -p $O
R=$(pwd)
STAGES=${STAGES:-test,ab}
stage_test() { timeout 900 python -m pytest tests/test_planes.py -m gpu -q -x -k "wgrad" --durations=5 > $O/b_wgrad_tests.log 2>&1; tail -12 $O/b_wgrad_tests.log; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    print("   %.3f ms/step  %.1f proposals/s  wgrad: %s" % (d["ms_per_step"], d["value"], {k: v for k, v in rd.items() if "wgrad" in k}))
except Exception as e:
    print("   (no line: %r)" % e)
PY
}
stage_ab() {
  for rep in 1 2; do
    for cfg in off default "900,300" "225,75" "1800,600" "450,600" "450,40"; do
      f=$O/b_bench_${cfg//,/_}_$rep.json
      if [ $cfg = off ]; then SSN_GROUP_WGRAD=0 timeout 300 python bench.py --cpu-baseline-videos 0 > $f 2> $f.err
      elif [ $cfg = default ]; then timeout 300 python bench.py --cpu-baseline-videos 0 > $f 2> $f.err
      else SSN_GROUP_TUNING=$cfg timeout 300 python bench.py --cpu-baseline-videos 0 > $f 2> $f.err; fi
      echo "$cfg #$rep"; line $f; grep -v Warning $f.err | grep -i "error\|Traceback" | head -3
    done
  done
}
for st in ${STAGES//,/ }; do echo "== $st $(date +%T)"; cd $R; stage_$st; done
