#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python tools/layer_table.py > gpurun_out/layers.txt 2>&1; echo "rc=$?" >> gpurun_out/layers.txt
