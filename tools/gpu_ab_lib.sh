#!/bin/bash
# A/B of two builds of the library on ONE box: the in-tree libssn_hip.so (NEW) against tools/.ab/libssn_prev.so (PREV, built from the
# previous commit on the build host; *.so files travel with the snapshot).  The stamp next to the library stays the NEW sources', so
# build() does not recompile on the box.  Alternating bench runs.
O=${O:-gpurun_out/r6}; mkdir -p $O
L=action-detection_amd/libssn_hip.so
cp $L /tmp/new.so
for rep in ${REPS:-1 2 3}; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/.ab/libssn_prev.so $L; else cp /tmp/new.so $L; fi
    timeout 300 python bench.py --cpu-baseline-videos 0 ${BENCH_ARGS:-} > $O/ab_${which}_$rep.json 2> $O/ab_${which}_$rep.err
    python - $O/ab_${which}_$rep.json $which $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    keep = {k: (round(v.get("ms_per_step", 0), 3), round(v.get("tflops", 0), 1)) for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}
    print("%s #%s  %.3f ms/step  %.1f proposals/s  %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], keep))
except Exception as e:
    print("   (no line: %r)" % e)
PY
    grep -i "error\|Traceback" $O/ab_${which}_$rep.err | head -3
  done
done
cp /tmp/new.so $L
