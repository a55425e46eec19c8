#!/bin/bash
# Round 4, call B: clock control v2, Inception-v3 on the planes path (tile table, tests, bench lines), GPU augmentation test.
O=gpurun_out/r4b; mkdir -p $O
STAGES=${STAGES:-clock,aug,tune3,test3,bench3,bench}
has() { [[ ",$STAGES," == *",$1,"* ]]; }
if has clock; then timeout 300 tools/clock/clock_control 3 > $O/clock_control.txt 2>&1; cat $O/clock_control.txt; fi
if has aug; then timeout 600 python -m pytest tests/test_transforms.py tests/test_scale_guard.py -m gpu -q -k "augmentation or dense_tester" > $O/aug_tests.log 2>&1; tail -3 $O/aug_tests.log; fi
if has tune3; then timeout 900 python tools/autotune_pl.py 144 InceptionV3 > $O/autotune_pl_inceptionv3.txt 2>&1; tail -2 $O/autotune_pl_inceptionv3.txt; cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_pl.json; fi
if has test3; then timeout 1500 python -m pytest tests/test_inceptionv3.py -m gpu -q -s > $O/v3_tests.log 2>&1; grep -v "^$" $O/v3_tests.log | tail -12; fi
if has bench3; then
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-260 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 2 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-260 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err
  SSN_LAYOUT=f32 timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 --no-kernel-events > $O/bench_train_inceptionv3_f32layout.json 2>/dev/null; cut -c1-200 $O/bench_train_inceptionv3_f32layout.json
fi
if has bench; then timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['scale_guard'])[:1500])"; fi
du -sh $O
