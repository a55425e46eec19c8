#!/bin/bash
# round 6, call F: the stem on the haloed kernel -- GPU parity, every tile on the two stem shapes (cold operands), and the bench with the
# stem's table entry set to the haloed tile against the plain one.
O=gpurun_out/r6; mkdir -p $O
T=action-detection_amd/tuned_tiles_pl.json
timeout 600 python -m pytest tests/test_planes.py -x -q -m gpu -k "stem_on_the_haloed or conv_pl_forward" > $O/f_tests.txt 2>&1; tail -3 $O/f_tests.txt
cp $T /tmp/old_table.json
COLD=1 KINDS=fwd ONLY="|64|7|7|2|224" timeout 600 python tools/autotune_pl.py 288 BNInception > $O/f_autotune.txt 2> $O/f_autotune.err; grep "7|7" $O/f_autotune.txt
cp $T /tmp/new_table.json; cp $T $O/f_tuned_tiles_pl.json
for rep in 1 2; do for which in old new; do
  cp /tmp/${which}_table.json $T
  timeout 300 python bench.py --cpu-baseline-videos 0 --no-secondary > $O/f_${which}_$rep.json 2> $O/f_${which}_$rep.err
  python - $O/f_${which}_$rep.json $which $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    keep = {k: (round(v.get("ms_per_step", 0), 3), round(v.get("tflops", 0), 1)) for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}
    print("%s #%s  %.3f ms/step  %.1f proposals/s  frac %.4f %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["roofline"]["frac"], keep))
except Exception as e:
    print("   (no line: %r)" % e)
PY
done; done 2>&1 | tee $O/f_ab.txt
cp /tmp/new_table.json $T
echo "F: done at ${SECONDS}s"
