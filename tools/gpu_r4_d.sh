#!/bin/bash
# Round 4, call D: tile tables with the high-occupancy conv_pl variants (BN-Inception RGB + Flow, Inception-v3; forward + dgrad), then
# the bench lines, the launch inventory of one eager step.
O=gpurun_out/r4d; mkdir -p $O
STAGES=${STAGES:-tune,tune3,bench,bench3,trace}
has() { [[ ",$STAGES," == *",$1,"* ]]; }
R=$(pwd)
if has tune; then KINDS=fwd,dgrad timeout 1500 python tools/autotune_pl.py 288 BNInception > $O/autotune_pl.txt 2>&1; tail -1 $O/autotune_pl.txt | cut -c1-300; fi
if has tune3; then KINDS=fwd,dgrad timeout 1500 python tools/autotune_pl.py 144 InceptionV3 > $O/autotune_pl_inceptionv3.txt 2>&1; tail -1 $O/autotune_pl_inceptionv3.txt | cut -c1-300; fi
cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_pl.json
if has bench; then timeout 600 python bench.py --cpu-baseline-videos 0 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['roofline']['frac'], json.dumps(d['roofline_detail'])[:900])"; fi
if has bench3; then
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-260 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err | cut -c1-600
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 4 --steps 5 --warmup 2 --cpu-baseline-videos 0 --no-kernel-events > $O/bench_train_inceptionv3_4videos.json 2>/dev/null; cut -c1-200 $O/bench_train_inceptionv3_4videos.json
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 2 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-260 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err | cut -c1-600
fi
if has trace; then
  cd /tmp && export TMPDIR=/tmp
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
with open("gpurun_out/r4d/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
  find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
fi
du -sh $O
