#!/bin/bash
# round 6, call L: the N > 1 control flow on a 1-GPU box with the closing tree -- (a) two ranks sharing the GPU over gloo, launched by
# the driver's own command line (torch.distributed.run), eager; (b) the same with hipGraph capture; (c) 1-rank RCCL both collective modes.
O=gpurun_out/r6l; mkdir -p $O
export SSN_BENCH_ONE_DEVICE=1 SSN_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --no-graph --cpu-baseline-videos 0 > $O/dist2_eager.log 2>&1; echo "eager rc=$?"; grep '^{' $O/dist2_eager.log | cut -c1-220
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 > $O/dist2_graph.log 2>&1; echo "graph rc=$?"; grep '^{' $O/dist2_graph.log | cut -c1-220; grep -i "error\|Traceback" $O/dist2_graph.log | head -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 --collectives overlapped > $O/dist2_overlapped.log 2>&1; echo "overlapped rc=$?"; grep '^{' $O/dist2_overlapped.log | cut -c1-220; grep -i "error\|Traceback" $O/dist2_overlapped.log | head -3
echo "L: done at ${SECONDS}s"
