#!/bin/bash
# call U: max-pool forward by bands (LDS) vs per output: parity tests, cold microbench, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_planes.py -q -x -m gpu -k "pools" 2>&1 | tail -2
for mode in 0 1 0 1; do SSN_POOL_BANDS=$mode timeout 600 python tools/bench_pool_fwd.py 2>/dev/null | tee -a $O/u_pool_fwd.txt; done
for rep in 1 2 3; do for mode in 0 1; do
  SSN_POOL_BANDS=$mode timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary > $O/u_${mode}_$rep.json 2> $O/u_err.txt
  python - $O/u_${mode}_$rep.json $mode $rep <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("bands=%s #%s  %.3f ms/step  %.1f proposals/s  loss %.8f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["final_loss"]))
except Exception as e:
    print("bands=%s #%s no line (%r)" % (sys.argv[2], sys.argv[3], e))
PY
done; done 2>&1 | tee $O/u_ab.txt
echo "U: done at ${SECONDS}s"
