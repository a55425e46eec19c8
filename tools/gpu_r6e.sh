#!/bin/bash
# round 6, call E: tile table re-tuned with COLD operands (rotating buffer sets) and the 16-byte epilogue, forward + dgrad, then the
# default bench line with the old and the new table alternating on this box.
O=gpurun_out/r6; mkdir -p $O
T=action-detection_amd/tuned_tiles_pl.json
cp $T /tmp/old_table.json
COLD=1 KINDS=fwd,dgrad timeout 1500 python tools/autotune_pl.py 288 BNInception > $O/e_autotune.txt 2> $O/e_autotune.err; tail -3 $O/e_autotune.txt
cp $T $O/e_tuned_tiles_pl_cold.json; cp $T /tmp/new_table.json
for rep in 1 2 3; do for which in old new; do
  cp /tmp/${which}_table.json $T
  timeout 300 python bench.py --cpu-baseline-videos 0 --no-secondary > $O/e_${which}_$rep.json 2> $O/e_${which}_$rep.err
  python - $O/e_${which}_$rep.json $which $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    keep = {k: (round(v.get("ms_per_step", 0), 3), round(v.get("tflops", 0), 1)) for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}
    print("%s #%s  %.3f ms/step  %.1f proposals/s  frac %.4f %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["roofline"]["frac"], keep))
except Exception as e:
    print("   (no line: %r)" % e)
PY
done; done 2>&1 | tee $O/e_ab.txt
cp /tmp/new_table.json $T
echo "E: done at ${SECONDS}s"
