#!/bin/bash
# call R: the one-tap weight-gradient family as gangs (wgrad_gang1_kernel) vs the in-order grid: tests, per-problem over-fetch, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_planes.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3
for mode in 1 0; do
  SSN_WGRAD_GANG=$mode timeout 600 python tools/pmc_wgrad_alone.py run $O/r_alone_$mode > $O/r_alone_$mode.txt 2> $O/r_alone_$mode.err; echo "events gang=$mode rc=$?"; tail -1 $O/r_alone_$mode.txt
  (cd /tmp; SSN_WGRAD_GANG=$mode timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/$O/r_pmc_$mode" -o p -- python "$R/tools/pmc_wgrad_alone.py" run "$R/$O/r_pmcrun_$mode" > "$R/$O/r_pmc_$mode.log" 2>&1; echo "pmc rc=$?")
  python tools/pmc_wgrad_alone.py parse $O/r_pmcrun_$mode $O/r_pmc_$mode > $O/r_overfetch_$mode.txt 2>&1; tail -1 $O/r_overfetch_$mode.txt
  find $O/r_pmc_$mode -name "*kernel_trace.csv" -delete
done
for rep in 1 2 3; do for mode in 0 1; do
  SSN_WGRAD_GANG=$mode timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary > $O/r_${mode}_$rep.json 2> $O/r_${mode}_$rep.err
  python - $O/r_${mode}_$rep.json $mode $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    rd = d.get("roofline_detail", {})
    print("gang=%s #%s  %.3f ms/step  %.1f proposals/s  frac %.4f loss %.8f %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], {k: (v.get("ms"), v.get("achieved")) for k, v in rd.items() if isinstance(v, dict) and "wgrad" in k}))
except Exception as e:
    print("gang=%s #%s no line (%r)" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/r_ab.txt
echo "R: done at ${SECONDS}s"
