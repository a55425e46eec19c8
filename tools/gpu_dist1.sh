#!/bin/bash
# exercise the multi-GPU code path (RCCL collectives + hipGraph capture + two-stream backward) on a 1-GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export SSN_FORCE_ALLREDUCE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-videos 0 > gpurun_out/bench_dist1.log 2>&1
echo "rc=$?" >> gpurun_out/bench_dist1.log; tail -3 gpurun_out/bench_dist1.log | cut -c1-900
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-videos 0 --no-graph > gpurun_out/bench_dist1_eager.log 2>&1
echo "rc=$?" >> gpurun_out/bench_dist1_eager.log; tail -2 gpurun_out/bench_dist1_eager.log | cut -c1-600
