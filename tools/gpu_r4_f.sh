#!/bin/bash
# Round 4, call F: fused heads on the GPU (kernel test + the end-to-end model tests that run through them), bench line, launch
# inventory, Inception-v3 tile re-tune (12 tiles) + its lines.
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -k "fused_heads or linear or stpp" > $O/heads_tests.log 2>&1; tail -2 $O/heads_tests.log
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_golden.py tests/test_end_to_end.py -m gpu -q > $O/model_tests.log 2>&1; tail -3 $O/model_tests.log
for f in 1 0 1 0; do SSN_FUSED_HEADS=$f timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fused_heads=$f: %.3f ms  %.1f proposals/s' % (d['ms_per_step'], d['value']))" | tee -a $O/fused_heads_ab.txt; done
timeout 600 python bench.py --cpu-baseline-videos 0 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['roofline']['frac'], d['hbm_kernels']['total_ms_per_step'], sorted((k, v['avg_us']) for k, v in d['hbm_kernels']['kernels'].items()))"
KINDS=fwd,dgrad timeout 1500 python tools/autotune_pl.py 144 InceptionV3 > $O/autotune_pl_inceptionv3.txt 2>&1; tail -1 $O/autotune_pl_inceptionv3.txt | cut -c1-200; cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_pl.json
timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-200 $O/bench_train_inceptionv3.json
timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 2 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-200 $O/bench_dense_inceptionv3.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 1 --warmup 2 > $R/$O/trace.log 2>&1
cd $R; f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")) for r in rows]
sgd = [i for i, n in enumerate(names) if "sgd_multi" in n]
ends = [i for k, i in enumerate(sgd) if k + 1 == len(sgd) or sgd[k + 1] != i + 1]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
with open("gpurun_out/r4f/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
du -sh $O
