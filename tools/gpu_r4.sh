#!/bin/bash
# Round 4: the evidence set in ONE gpurun call (run from the repo root on the GPU box; writes gpurun_out/r4/).  STAGES=a,b,... selects
# the stages and their order.
#   tests   the whole GPU tier                      smoke   __graft_entry__.smoke()
#   bench   the default line (BASELINE config 2)    flow    config 3            dist1  both --collectives modes on a 1-rank RCCL group
#   v3      Inception-v3 training + dense-test lines, BN-Inception dense test   input  prefetcher from un-cropped frames (+ --precropped)
#   prof    rocprofv3 kernel stats, eager single stream, 60 steps               pmc    tools/gpu_pmc.sh -> summary JSON
#   seq     launch inventory of one eager step      clock   tools/clock/clock_control (shader clock under known loops)
#   calib   FETCH_SIZE / WRITE_SIZE against known byte counts           bnmode  bench --bn-mode partial, planes vs fp32-layout executor
O=gpurun_out/r4; mkdir -p $O
STAGES=${STAGES:-tests,smoke,bench,flow,dist1,v3,input,prof,pmc,seq,clock,calib}
R=$(pwd)
stage_tests() { timeout 3000 python -m pytest tests/ -m gpu -q --durations=15 > $O/gpu_tests.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -12; }
stage_smoke() { timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log; }
stage_bench() { timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; }
stage_flow() { timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 > $O/bench_flow.json 2>/dev/null; cut -c1-200 $O/bench_flow.json; }
stage_dist1() {
  for m in separate overlapped; do
    SSN_FORCE_ALLREDUCE=1 timeout 600 python bench.py --collectives $m --cpu-baseline-videos 0 --no-kernel-events 2> $O/bench_dist1_$m.err | grep '^{' > $O/bench_dist1_$m.json   # (RCCL prints its banner on stdout)
    cut -c1-200 $O/bench_dist1_$m.json; tail -2 $O/bench_dist1_$m.err
  done
}
stage_v3() {
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-300 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-300 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err
  SSN_LAYOUT=f32 timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3_f32layout.json 2>/dev/null; cut -c1-200 $O/bench_dense_inceptionv3_f32layout.json
  timeout 900 python bench.py --mode dense-test --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_bninception.json 2>/dev/null; cut -c1-200 $O/bench_dense_bninception.json
}
stage_bnmode() {      # --bn_mode partial: the stem's BatchNorm in training mode, on the planes kernels (1) and on the fp32-layout executor (0)
  for e in 1 0; do
    SSN_PLANES_TRAIN_BN=$e timeout 600 python bench.py --bn-mode partial --cpu-baseline-videos 0 --no-kernel-events > $O/bench_bn_partial_planes$e.json 2> $O/bench_bn_partial_planes$e.err
    cut -c1-200 $O/bench_bn_partial_planes$e.json; tail -2 $O/bench_bn_partial_planes$e.err
  done
}
stage_input() {
  timeout 600 python tools/bench_input_pipeline.py > $O/input_pipeline.json 2> $O/input_pipeline.err; cat $O/input_pipeline.json; tail -2 $O/input_pipeline.err
  timeout 600 python tools/bench_input_pipeline.py --precropped > $O/input_pipeline_precropped.json 2>/dev/null; cat $O/input_pipeline_precropped.json
}
stage_prof() {
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events --steps 60 --warmup 3 > $R/$O/prof.log 2>&1
  cd $R; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_eager.csv; head -12 "$f" | cut -c1-160
}
stage_pmc() { bash tools/gpu_pmc.sh; cp gpurun_out/pmc/summary.json $O/pmc_summary_planes.json; }
stage_seq() {
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
with open("gpurun_out/r4/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
  find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
}
stage_clock() { timeout 300 tools/clock/clock_control 3 > $O/clock_control.txt 2>&1; cat $O/clock_control.txt; }
stage_calib() {
  cd /tmp && export TMPDIR=/tmp
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/calib$i -o c -- $R/tools/clock/fetch_calib > $R/$O/calib$i.log 2>&1; echo "calib $i rc=$?"
  done
  cd $R
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r4/calib*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = open("gpurun_out/r4/fetch_calib_summary.txt", "w")
for k in sorted(acc):
    line = "%-28s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
    print(line); out.write(line + "\n")
out.write("known: 1 GiB = 1048576 KiB touched once per kernel (FETCH_SIZE / WRITE_SIZE are KiB)\n")
PY
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
}
for st in ${STAGES//,/ }; do echo "== $st"; cd $R; stage_$st; done
cd $R; du -sh $O
