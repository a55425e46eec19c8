#!/bin/bash
# round 5, call H: haloed 3x3 kernel with the nine taps unrolled -- parity, A/B against the previous library on this box, re-tune of the
# 3x3 / stride 1 layers, bench with the new table
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_planes.py -m gpu -x -q -k "conv_pl" 2>&1 | tail -4
REPS="1 2" bash tools/gpu_ab_lib.sh
cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_before.json
HALO_ONLY=1 KINDS=fwd,dgrad timeout 600 python tools/autotune_pl.py 288 BNInception 2>&1 | tail -40
cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_after.json
for r in 1 2; do
  timeout 300 python bench.py --cpu-baseline-videos 0 > $O/retuned_$r.json 2> $O/retuned_$r.err
  python - $O/retuned_$r.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rd = d.get("roofline_detail", {})
print("retuned  %.3f ms/step  %.1f proposals/s  %s" % (d["ms_per_step"], d["value"], {k: (round(v.get("ms_per_step", 0), 3), round(v.get("tflops", 0), 1)) for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}))
PY
done
