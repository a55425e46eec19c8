#!/bin/bash
# Round 6: the evidence set (tools/gpu_r5.sh with the output directory of this round) (run from the repo root on the GPU box; writes gpurun_out/r6ev/).  STAGES=a,b,... selects stages / order.
#   tests   the whole GPU tier                      smoke   __graft_entry__.smoke()
#   bench   the default line (BASELINE config 2)    bench400  the same with 400 timed steps (sustained)
#   flow    config 3            dist1  both --collectives modes on a 1-rank RCCL group
#   v3      Inception-v3 training + dense-test lines, BN-Inception dense test
#   prof    rocprofv3 kernel stats, eager single stream, 60 steps               pmc    tools/gpu_pmc.sh -> summary JSON
#   seq     launch inventory of one eager step      clock   tools/clock/clock_control (burst / zero-vs-random operands)
O=gpurun_out/r6ev; mkdir -p $O
STAGES=${STAGES:-pmc,prof,seq,bench,bench400,dist1,v3,smoke,tests}
R=$(pwd)
stage_tests() { timeout 3000 python -m pytest tests/ -m gpu -q --durations=15 > $O/gpu_tests.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -12; }
stage_smoke() { timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log; }
stage_bench() { timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; }
stage_bench400() { timeout 600 python bench.py --steps 400 --warmup 5 --cpu-baseline-videos 0 --no-kernel-events > $O/bench_400steps.json 2> $O/bench_400steps.err; cut -c1-200 $O/bench_400steps.json; }
stage_flow() { timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 > $O/bench_flow.json 2>/dev/null; cut -c1-200 $O/bench_flow.json; }
stage_dist1() {
  for m in separate overlapped; do
    SSN_FORCE_ALLREDUCE=1 timeout 600 python bench.py --collectives $m --cpu-baseline-videos 0 --no-kernel-events 2> $O/bench_dist1_$m.err | grep '^{' > $O/bench_dist1_$m.json
    cut -c1-200 $O/bench_dist1_$m.json; tail -2 $O/bench_dist1_$m.err
  done
}
stage_v3() {
  timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-300 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-300 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err
  timeout 900 python bench.py --mode dense-test --steps 3 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_bninception.json 2>/dev/null; cut -c1-200 $O/bench_dense_bninception.json
}
stage_prof() {
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --single-stream --no-kernel-events --steps 60 --warmup 3 > $R/$O/prof.log 2>&1
  cd $R; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_eager.csv; head -14 "$f" | cut -c1-160
}
stage_pmc() { bash tools/gpu_pmc.sh; cp gpurun_out/pmc/summary.json $O/pmc_summary.json; cp gpurun_out/pmc/summary.json profiles/r6_pmc_summary.json; }
stage_seq() {
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --single-stream --no-kernel-events --steps 1 --warmup 2 > $R/$O/trace.log 2>&1
  cd $R; f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")) for r in rows]
sgd = [i for i, n in enumerate(names) if "sgd_multi" in n]
ends = [i for k, i in enumerate(sgd) if k + 1 == len(sgd) or sgd[k + 1] != i + 1]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
with open("gpurun_out/r6ev/step_launch_sequence.txt", "w") as f:
    f.write("# launches of ONE eager training step in issue order (rocprofv3 --kernel-trace): index, duration us, kernel\n")
    for k in range(lo, hi):
        f.write("%4d %8.1f %s\n" % (k - lo, (int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3, names[k][:110]))
print("launches in the last step:", hi - lo)
PY
  find $O/trace -name "*.csv" -delete; find $O/trace -name "*.db" -delete
}
stage_clock() { timeout 300 tools/clock/clock_control 3 > $O/clock_control.txt 2>&1; cat $O/clock_control.txt; }
for st in ${STAGES//,/ }; do echo "== $st $(date +%T)"; cd $R; stage_$st; done
cd $R; du -sh $O
