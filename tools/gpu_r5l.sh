#!/bin/bash
# round 5, call L: grouped weight gradients on two lanes (SSN_WGRAD_LANES: the bandwidth-bound everything-else family on the side stream
# beside the nine-tap families).  Same library, alternating switch; the final loss after the 13 steps must be bit-identical (same kernels,
# same split plans, same reduction order).  If it wins: the 400-step line and the launch inventory with the switch on.
O=gpurun_out/r5; mkdir -p $O
for rep in ${REPS:-1 2}; do for w in 0 1; do
  SSN_WGRAD_LANES=$w timeout 60 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/l_${w}_$rep.json 2> $O/l_${w}_$rep.err
done; done
python - $O <<'PY' | tee $O/l_decision.txt
import json, sys
O = sys.argv[1]
def load(p):
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    return d["ms_per_step"], d["final_loss"]
import glob, re
reps = sorted({int(re.search(r"l_\d_(\d+)\.json", f).group(1)) for f in glob.glob(O + "/l_0_*.json")})
r = {(w, rep): load("%s/l_%d_%d.json" % (O, w, rep)) for w in (0, 1) for rep in reps}
for k in sorted(r):
    print("lanes=%d #%d  %.3f ms/step  final_loss %.9g" % (k[0], k[1], r[k][0], r[k][1]))
same = len({v[1] for v in r.values()}) == 1
m0, m1 = sum(r[(0, i)][0] for i in reps) / len(reps), sum(r[(1, i)][0] for i in reps) / len(reps)
win = same and m1 < m0 - 0.05 and sum(r[(1, i)][0] < r[(0, i)][0] for i in reps) >= len(reps) - 1
print("bit-identical final loss: %s; mean %.3f -> %.3f ms -> %s" % (same, m0, m1, "ADOPT" if win else "KEEP OFF"))
PY
if grep -q ADOPT $O/l_decision.txt && [ $SECONDS -lt 55 ]; then
  SSN_WGRAD_LANES=1 timeout 40 python bench.py --steps 400 --warmup 5 --cpu-baseline-videos 0 --no-kernel-events > $O/l_bench_400steps.json 2>/dev/null; cut -c1-200 $O/l_bench_400steps.json
fi
echo "L: done at ${SECONDS}s"
