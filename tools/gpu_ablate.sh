#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python tools/ablate_conv.py > gpurun_out/ablate.txt 2>&1; echo "rc=$?" >> gpurun_out/ablate.txt
