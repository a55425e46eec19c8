#!/bin/bash
# call T: planner constants of the grouped weight gradients (SSN_GROUP_TUNING = fixed9,fixed1,min9,min1) inside the step, two passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O
for rep in 1 2; do for t in 450,150,4,32 300,150,4,32 700,150,4,32 1000,150,4,32 450,100,4,32 450,250,4,32 450,400,4,32 450,150,4,64 450,150,8,32 700,250,8,32; do
  SSN_GROUP_TUNING=$t timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary > $O/t_${t}_$rep.json 2> $O/t_err.txt
  python - $O/t_${t}_$rep.json $t $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("tuning %-16s #%s  %.3f ms/step  wgrad %.3f ms" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline_detail"]["conv_wgrad_all"]["ms_per_step"]))
except Exception as e:
    print("tuning %s #%s no line (%r)" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/t_ab.txt
echo "T: done at ${SECONDS}s"
