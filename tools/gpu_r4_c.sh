#!/bin/bash
# Round 4, call C: Inception-v3 training on the planes path (diagnose the non-settling forward), v3 tile table, BN-Inception bench
# with the gradient head-room, kernel trace of one eager step (launch inventory).
O=gpurun_out/r4c; mkdir -p $O
STAGES=${STAGES:-v3train,tune3,bench3,bench,trace}
has() { [[ ",$STAGES," == *",$1,"* ]]; }
R=$(pwd)
if has v3train; then timeout 600 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 3 --warmup 1 --cpu-baseline-videos 0 --no-graph --no-kernel-events > $O/v3_eager.json 2> $O/v3_eager.err; cut -c1-200 $O/v3_eager.json; tail -3 $O/v3_eager.err | cut -c1-1500; fi
if has tune3; then timeout 1200 python tools/autotune_pl.py 144 InceptionV3 > $O/autotune_pl_inceptionv3.txt 2>&1; tail -2 $O/autotune_pl_inceptionv3.txt | cut -c1-300; cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_pl.json; fi
if has bench3; then
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-260 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err | cut -c1-600
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 2 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-260 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err | cut -c1-600
fi
if has bench; then timeout 600 python bench.py --cpu-baseline-videos 0 --steps 40 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['scale_guard'])[:1500])"
  timeout 600 python bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 40 > $O/bench_eager40.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_eager40.json')); print(d['ms_per_step'], json.dumps(d['scale_guard'])[:1500])"; fi
if has trace; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 1 --warmup 2 > $R/$O/trace.log 2>&1
  cd $R; f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = the rows after the last sgd_multi_kernel group before the final one
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("void ", "") for r in rows]
sgd = [i for i, n in enumerate(names) if "sgd_multi" in n]
# steps end with the last sgd launch of a group: find group ends
ends = [i for k, i in enumerate(sgd) if k + 1 == len(sgd) or sgd[k + 1] != i + 1]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
with open("gpurun_out/r4c/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
  find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
fi
du -sh $O
