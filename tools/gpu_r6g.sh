#!/bin/bash
# round 6, call G: the driver's command (default bench line incl. the secondary runs), wall-clocked.
O=gpurun_out/r6; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/g_bench.json 2> $O/g_bench.err; tail -1 $O/g_bench.json | cut -c1-300; grep -E "Elapsed|Traceback|Error" $O/g_bench.err | head
python - $O/g_bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"])
print(json.dumps(d.get("secondary"), indent=1)[:3000])
PY
echo "bench wall ${SECONDS}s"; echo "G: done at ${SECONDS}s"
