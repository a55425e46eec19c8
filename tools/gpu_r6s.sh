#!/bin/bash
# call S: one-tap weight-gradient family, problems interleaved by bytes per MAC vs longest-first: family time, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for mode in 0 1; do
  SSN_WGRAD_INTERLEAVE=$mode timeout 600 python tools/pmc_wgrad_alone.py run $O/s_alone_$mode > $O/s_alone_$mode.txt 2> $O/s_alone_$mode.err; echo "interleave=$mode: $(tail -1 $O/s_alone_$mode.txt)"
done; done
for rep in 1 2 3; do for mode in 0 1; do
  SSN_WGRAD_INTERLEAVE=$mode timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary > $O/s_${mode}_$rep.json 2> $O/s_${mode}_$rep.err
  python - $O/s_${mode}_$rep.json $mode $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    print("interleave=%s #%s  %.3f ms/step  %.1f proposals/s  frac %.4f loss %.8f %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], {k: v for k, v in rd.items() if "wgrad" in k}))
except Exception as e:
    print("interleave=%s #%s no line (%r)" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/s_ab.txt
echo "S: done at ${SECONDS}s"
