#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
timeout 1800 python -m pytest tests/test_dense_test.py tests/test_inceptionv3.py tests/test_model_gpu.py -m gpu -q -k "dense or chunked" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-200 $O/bench_dense_inceptionv3.json; tail -1 $O/bench_dense_inceptionv3.err | cut -c1-300
SSN_LAYOUT=f32 timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 2>/dev/null | cut -c1-200
timeout 900 python bench.py --mode dense-test --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_bninception.json 2>/dev/null; cut -c1-200 $O/bench_dense_bninception.json
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | cut -c1-200
du -sh $O
