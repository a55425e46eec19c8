#!/bin/bash
# Round 4, call A: the range-guard tests on the real model, smoke, a bench line, and the two control experiments of VERDICT r3 #7:
# (a) shader clock through the kernels' own stamps on loops of known content + rocm-smi sclk while the training step runs,
# (b) FETCH_SIZE / WRITE_SIZE against known byte counts per access width.
O=gpurun_out/r4a; mkdir -p $O
STAGES=${STAGES:-guard,smoke,bench,clock,calib}
has() { [[ ",$STAGES," == *",$1,"* ]]; }
R=$(pwd)
if has guard; then timeout 1500 python -m pytest tests/test_scale_guard.py tests/test_planes.py -m gpu -q -s > $O/guard_tests.log 2>&1; tail -12 $O/guard_tests.log; fi
if has smoke; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log; fi
if has bench; then timeout 600 python bench.py --cpu-baseline-videos 0 > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json; tail -2 $O/bench.err
  SSN_PL_OVERLAP_WGRAD=0 timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench_no_overlap.json 2>/dev/null; cut -c1-200 $O/bench_no_overlap.json; fi
if has clock; then
  ( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | tr '\n' ' '; echo; sleep 0.1; done ) > $O/smi_during_control.txt &
  timeout 300 tools/clock/clock_control 30 > $O/clock_control.txt 2>&1; wait; cat $O/clock_control.txt
  ( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | tr '\n' ' '; echo; sleep 0.1; done ) > $O/smi_during_step.txt &
  timeout 300 python bench.py --cpu-baseline-videos 0 --no-kernel-events --steps 400 --warmup 5 > $O/bench_400.json 2>/dev/null; wait
  sort $O/smi_during_step.txt | uniq -c | sort -rn | head -5; cut -c1-160 $O/bench_400.json
fi
if has calib; then
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
for path in glob.glob("gpurun_out/r4a/calib*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = open("gpurun_out/r4a/fetch_calib_summary.txt", "w")
for k in sorted(acc):
    line = "%-28s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
    print(line); out.write(line + "\n")
out.write("known: 1 GiB = 1048576 KiB touched once per kernel (FETCH_SIZE / WRITE_SIZE are KiB)\n")
PY
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
fi
du -sh $O
