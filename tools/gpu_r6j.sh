#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
for cfg in "1 --no-graph" "0 --no-graph" "1 " "0 " "1 " "0 "; do set -- $cfg
  SSN_S2_CLASS_LANES=$1 timeout 300 python bench.py --cpu-baseline-videos 0 --no-secondary --no-kernel-events $2 > $O/j.json 2> $O/j.err; echo "classlanes=$1 $2 rc=$? $(python -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/j.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['final_loss'], d['config']['launch'][:30])
except Exception as e: print('no line')")"; tail -2 $O/j.err | grep -v amdgpu.ids
done
