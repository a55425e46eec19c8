#!/bin/bash
# Round 3: the evidence set in ONE gpurun call (run from the repo root on the GPU box; writes gpurun_out/r3/):
#   GPU test tier, smoke(), default bench line, Flow line, forced-collective 1-rank RCCL lines (both --collectives modes),
#   Inception-v3 training + dense-test lines (bench.py --arch / --mode), input-pipeline wait, rocprofv3 kernel stats (eager,
#   single stream) of the default step.   STAGES=tests,bench,... selects a subset.
O=gpurun_out/r3; mkdir -p $O
STAGES=${STAGES:-tests,smoke,bench,flow,dist1,v3,dense,input,prof}
has() { [[ ",$STAGES," == *",$1,"* ]]; }
if has tests; then timeout 2400 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; fi
if has smoke; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log; fi
if has bench; then timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; fi
if has flow; then timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 > $O/bench_flow.json 2>/dev/null; cut -c1-200 $O/bench_flow.json; fi
if has dist1; then
  for m in separate overlapped; do
    SSN_FORCE_ALLREDUCE=1 timeout 600 python bench.py --collectives $m --cpu-baseline-videos 0 --no-kernel-events 2> $O/bench_dist1_$m.err | grep '^{' > $O/bench_dist1_$m.json   # (RCCL prints its banner on stdout)
    cut -c1-200 $O/bench_dist1_$m.json; tail -2 $O/bench_dist1_$m.err
  done
fi
if has v3; then timeout 900 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/bench_train_inceptionv3.json 2> $O/bench_train_inceptionv3.err; cut -c1-300 $O/bench_train_inceptionv3.json; tail -2 $O/bench_train_inceptionv3.err; fi
if has dense; then
  timeout 900 python bench.py --mode dense-test --arch InceptionV3 --steps 2 --warmup 1 > $O/bench_dense_inceptionv3.json 2> $O/bench_dense_inceptionv3.err; cut -c1-300 $O/bench_dense_inceptionv3.json; tail -2 $O/bench_dense_inceptionv3.err
  timeout 900 python bench.py --mode dense-test --steps 2 --warmup 1 --cpu-baseline-videos 0 > $O/bench_dense_bninception.json 2>/dev/null; cut -c1-200 $O/bench_dense_bninception.json
fi
if has input; then timeout 600 python tools/bench_input_pipeline.py > $O/input_pipeline.json 2> $O/input_pipeline.err; cat $O/input_pipeline.json; tail -2 $O/input_pipeline.err; fi
if has prof; then
  R=$(pwd); cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o eager -- python $R/bench.py --cpu-baseline-videos 0 --no-graph --no-kernel-events > $R/$O/prof.log 2>&1
  cd $R; find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
fi
