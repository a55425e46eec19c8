#!/bin/bash
# Inception-v3 backward on the GPU: new kernel tests, backbone / SSN training parity, training-step bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out/r2n; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -k "rect_backward or pad0 or dgrad_s2 or relu_bn or conv_wgrad_x6" > $O/gpu_tests_k.log 2>&1; echo "rc=$?" >> $O/gpu_tests_k.log; tail -5 $O/gpu_tests_k.log
timeout 900 python -m pytest tests/test_inceptionv3.py -q -m gpu -s > $O/gpu_tests_v3.log 2>&1; echo "rc=$?" >> $O/gpu_tests_v3.log; grep -n "median\|passed\|failed\|Error\|rc=" $O/gpu_tests_v3.log | head -20
