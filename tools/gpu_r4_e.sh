#!/bin/bash
# Round 4, call E: A/B of the tile tables (round-3 table vs the one tuned with the high-occupancy tiles 12-15) on ONE box, alternating;
# deferred weight-gradient reduction on / off; launch inventory; input pipeline from un-cropped frames; new GPU tests.
O=gpurun_out/r4e; mkdir -p $O
T=action-detection_amd/tuned_tiles_pl.json
cp $T /tmp/tiles_saved.json
for rep in 1 2; do for tab in r3 r4d; do
  cp tools/tiles_ab/tiles_$tab.json $T
  timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('tiles $tab rep $rep: %.3f ms  %.1f proposals/s' % (d['ms_per_step'], d['value']))" | tee -a $O/tiles_ab.txt
done; done
cp tools/tiles_ab/tiles_r3.json $T
for d in 1 0 1 0; do SSN_DEFER_WGRAD_REDUCE=$d timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('defer_wgrad_reduce=$d: %.3f ms  %.1f proposals/s' % (d['ms_per_step'], d['value']))" | tee -a $O/defer_ab.txt; done
cp /tmp/tiles_saved.json $T
timeout 900 python -m pytest tests/test_planes.py tests/test_input_pipeline.py tests/test_transforms.py -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python tools/bench_input_pipeline.py > $O/input_pipeline.json 2> $O/input_pipeline.err; cat $O/input_pipeline.json; tail -2 $O/input_pipeline.err
timeout 600 python tools/bench_input_pipeline.py --precropped > $O/input_pipeline_precropped.json 2>/dev/null; cat $O/input_pipeline_precropped.json
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
with open("gpurun_out/r4e/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
du -sh $O
