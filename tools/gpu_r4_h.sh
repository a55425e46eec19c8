#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -k "fused_heads" > $O/heads_tests.log 2>&1; tail -2 $O/heads_tests.log
for f in 1 0 1 0; do SSN_FUSED_HEADS=$f timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fused_heads=$f: %.3f ms  %.1f proposals/s' % (d['ms_per_step'], d['value']))" | tee -a $O/fused_heads_ab.txt; done
timeout 600 python bench.py --cpu-baseline-videos 0 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['roofline']['frac'], d['hbm_kernels']['total_ms_per_step'], sorted((k, v['avg_us']) for k, v in d['hbm_kernels']['kernels'].items()))"
bash tools/gpu_r4_g.sh
