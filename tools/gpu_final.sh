#!/bin/bash
# last checks of the round: input-pipeline + Inception-v3 GPU tests, Inception-v3 dense-test line, PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/fin
timeout 900 python -m pytest tests/test_input_pipeline.py tests/test_inceptionv3.py tests/test_dense_test.py -x -q -m gpu > gpurun_out/fin/tests.log 2>&1; echo "rc=$?" >> gpurun_out/fin/tests.log; tail -3 gpurun_out/fin/tests.log
timeout 600 python tools/bench_dense_test.py --arch InceptionV3 --tick-batch 30 --cpu-ticks 4 > gpurun_out/fin/dense_v3.log 2>&1; echo "rc=$?" >> gpurun_out/fin/dense_v3.log; tail -2 gpurun_out/fin/dense_v3.log | cut -c1-400
bash tools/gpu_pmc.sh > gpurun_out/fin/pmc.log 2>&1; tail -2 gpurun_out/fin/pmc.log
