#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
timeout 1800 python -m pytest tests/test_dense_test.py tests/test_inceptionv3.py tests/test_scale_guard.py tests/test_kernels.py -m gpu -q -k "dense or test_forward or fused_heads or guard or dark" > $O/tests.log 2>&1; tail -3 $O/tests.log
for f in 1 0 1 0; do SSN_FUSED_HEADS=$f timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fused_heads=$f: %.3f ms  %.1f proposals/s' % (d['ms_per_step'], d['value']))" | tee -a $O/fused_heads_ab.txt; done
timeout 600 python bench.py --cpu-baseline-videos 0 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['roofline']['frac'], d['hbm_kernels']['total_ms_per_step'], sorted((k, v['avg_us']) for k, v in d['hbm_kernels']['kernels'].items()))"
timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-200 $O/bench_dense_inceptionv3.json; tail -1 $O/bench_dense_inceptionv3.err | cut -c1-300
SSN_INFER_CACHE=0 timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 2>/dev/null | cut -c1-200
SSN_SCALE_GUARD=deferred timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 2>/dev/null | cut -c1-200
SSN_LAYOUT=f32 timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 2>/dev/null | cut -c1-200
timeout 900 python bench.py --mode dense-test --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_bninception.json 2>/dev/null; cut -c1-200 $O/bench_dense_bninception.json
du -sh $O
