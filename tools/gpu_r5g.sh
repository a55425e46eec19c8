#!/bin/bash
# Round 5, call G: the stem's weight gradient on planes (wgrad_stem_body in the grouped launch) -- parity, then A/B against the
# fp32-layout kernel (SSN_STEM_PLANES=0), same box, alternating.
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_planes.py -m gpu -q -x -k "wgrad_group or pools" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "fwd_bwd_matches_oracle" 2>&1 | tail -3
for rep in 1 2 3; do
  for sp in 0 1; do
    SSN_STEM_PLANES=$sp timeout 300 python bench.py --cpu-baseline-videos 0 > $O/g_bench_stem${sp}_$rep.json 2> $O/g_bench_stem${sp}_$rep.err
    python - $O/g_bench_stem${sp}_$rep.json $sp $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    print("stem_planes=%s #%s  %.3f ms/step  %.1f proposals/s  wgrad_all: %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], rd.get("conv_wgrad_all")))
except Exception as e:
    print("   (no line: %r)" % e)
PY
    grep -i "error\|Traceback" $O/g_bench_stem${sp}_$rep.err | head -3
  done
done
