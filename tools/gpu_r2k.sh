#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2k; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_model_gpu.py -q -m gpu -k "not fwd_bwd_matches_oracle" > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -o p -- python $R/bench.py --cpu-baseline-videos 0 --no-kernel-events --no-graph --single-stream > $R/$O/prof_eager.log 2>&1; echo "prof eager rc=$?"; cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
