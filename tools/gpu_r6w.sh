#!/bin/bash
# call W: grid cap of the elementwise planes kernels (grid-stride loops; fewer waves = fewer amax atomics on one address): launch sequence + step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r6; mkdir -p $O
for cap in ${CAPS:-65536 8192 4096 2048}; do
  export SSN_PL_GRID_CAP=$cap
  STAGES=seq bash tools/gpu_r6_evidence.sh > /dev/null 2>&1; cp gpurun_out/r6ev/step_launch_sequence.txt $O/w_seq_$cap.txt
  python - $O/w_seq_$cap.txt $cap <<'PY'
import sys, re, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for l in open(sys.argv[1]):
    m = re.match(r"\s*\d+\s+([\d.]+)\s+(\S+)", l)
    if m and not m.group(2).startswith(("conv_pl", "wgrad_group")):
        k = re.sub(r"<.*", "", m.group(2)); tot[k] += float(m.group(1)); n[k] += 1
print("cap %s: non-conv launches %.1f us: " % (sys.argv[2], sum(tot.values())) + ", ".join("%s %.0f" % (k, v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]))
PY
  for rep in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-videos 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   cap $cap #$rep  %.3f ms/step  loss %.8f' % (d['ms_per_step'], d['final_loss']))"; done
done 2>&1 | tee $O/w_ab.txt
echo "W: done at ${SECONDS}s"
