#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
timeout 600 python tools/trace_x6.py > gpurun_out/trace_x6.log 2>&1; echo "rc=$?" >> gpurun_out/trace_x6.log; tail -20 gpurun_out/trace_x6.log
