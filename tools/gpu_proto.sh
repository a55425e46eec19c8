#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python tools/proto/run_bf16x6.py > gpurun_out/proto.txt 2>&1; echo "rc=$?" >> gpurun_out/proto.txt
