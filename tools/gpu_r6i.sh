#!/bin/bash
# round 6, call I: the parity classes of the stride-2 data gradients on two streams (SSN_S2_CLASS_LANES), same library, alternating;
# and the objective in one launch vs the three criterion modules + Python mix (SSN_BENCH_SEPARATE_LOSSES).
O=gpurun_out/r6; mkdir -p $O
timeout 600 python -m pytest tests/test_planes.py tests/test_inceptionv3.py -x -q -m gpu -k "stride2 or backbone_backward" > $O/i_tests.txt 2>&1; tail -2 $O/i_tests.txt
for rep in 1 2 3; do for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  SSN_S2_CLASS_LANES=$1 SSN_BENCH_SEPARATE_LOSSES=$2 timeout 300 python bench.py --cpu-baseline-videos 0 --no-secondary > $O/i_$1$2_$rep.json 2> $O/i_$1$2_$rep.err
  python - $O/i_$1$2_$rep.json "classlanes=$1 separate_losses=$2" $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    keep = {k: (round(v.get("ms_per_step", 0), 3), round(v.get("tflops", 0), 1)) for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}
    print("%s #%s  %.3f ms/step  %.1f proposals/s  frac %.4f loss %.9g %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], keep))
except Exception as e:
    print("   (no line: %r)" % e)
PY
done; done 2>&1 | tee $O/i_ab.txt
echo "I: done at ${SECONDS}s"
