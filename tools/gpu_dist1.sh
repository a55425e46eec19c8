#!/bin/bash
# the N > 1 code paths on a 1-GPU box (RCCL collectives on a 1-rank group): separate (two graphs) and overlapped (one graph)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/d1
export SSN_FORCE_ALLREDUCE=1
for mode in separate overlapped; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-videos 0 --no-kernel-events --collectives $mode > gpurun_out/d1/bench_$mode.log 2>&1
  echo "rc=$?" >> gpurun_out/d1/bench_$mode.log; grep "^{" gpurun_out/d1/bench_$mode.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['collectives'], d['value'], d['ms_per_step'], d['config']['launch'], d['final_loss'])"; tail -1 gpurun_out/d1/bench_$mode.log
done
unset SSN_FORCE_ALLREDUCE
timeout 600 python bench.py --cpu-baseline-videos 0 --no-kernel-events > gpurun_out/d1/bench_plain.log 2>&1; grep "^{" gpurun_out/d1/bench_plain.log | tail -1 | cut -c1-200
