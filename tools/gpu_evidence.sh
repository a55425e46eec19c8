#!/bin/bash
# round evidence in one call: GPU test tier, default bench line (+ CPU baseline), Flow bench line, forced-collective
# bench (RCCL + hipGraph path on one GPU), per-layer table, rocprofv3 kernel stats, PMC passes + summary
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/ev; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/ev/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/ev/gpu_tests.log; tail -3 gpurun_out/ev/gpu_tests.log
timeout 900 python bench.py > gpurun_out/ev/bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_default.log; tail -2 gpurun_out/ev/bench_default.log | cut -c1-300
timeout 600 python bench.py --modality Flow --cpu-baseline-videos 0 > gpurun_out/ev/bench_flow.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_flow.log
timeout 600 python bench.py --precision f32 --cpu-baseline-videos 0 > gpurun_out/ev/bench_f32.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_f32.log
SSN_FORCE_ALLREDUCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-videos 0 --no-kernel-events > gpurun_out/ev/bench_dist1.log 2>&1; echo "rc=$?" >> gpurun_out/ev/bench_dist1.log; tail -2 gpurun_out/ev/bench_dist1.log | cut -c1-200
timeout 300 python tools/layer_table.py > gpurun_out/ev/layers.txt 2>&1
timeout 300 python tools/bench_dense_test.py > gpurun_out/ev/dense_test.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ev/prof" -o run -- python "$R/bench.py" --cpu-baseline-videos 0 > "$R/gpurun_out/ev/prof.log" 2>&1
cd "$R"; find gpurun_out/ev -name "*kernel_trace.csv" -delete
bash tools/gpu_pmc.sh > gpurun_out/ev/pmc.log 2>&1; tail -2 gpurun_out/ev/pmc.log
du -sh gpurun_out/ev gpurun_out/pmc
