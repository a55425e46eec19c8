#!/bin/bash
# control flow of the N > 1 bench path on a 1-GPU box: two ranks share the GPU, gloo collectives, eager launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
export SSN_BENCH_ONE_DEVICE=1 SSN_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --no-graph --cpu-baseline-videos 1 > gpurun_out/bench_dist2.log 2>&1
echo "rc=$?" >> gpurun_out/bench_dist2.log; tail -4 gpurun_out/bench_dist2.log | cut -c1-600
