#!/bin/bash
# call V: parameter-only head of a pass on the side stream (SSN_HEAD_LANES) + gap_fwd with eight loads in flight: model tests, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_scale_guard.py tests/test_planes.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2 3; do for mode in 0 1; do
  SSN_HEAD_LANES=$mode timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary > $O/v_${mode}_$rep.json 2> $O/v_err.txt
  python - $O/v_${mode}_$rep.json $mode $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("head_lanes=%s #%s  %.3f ms/step  %.1f proposals/s  loss %.8f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["final_loss"]))
except Exception as e:
    print("head_lanes=%s #%s no line (%r)" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/v_ab.txt
tail -3 $O/v_err.txt
echo "V: done at ${SECONDS}s"
