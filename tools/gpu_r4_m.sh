#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout 1800 python -m pytest tests/test_scale_guard.py -m gpu -q -s -k "graph_replay" > $O/tests.log 2>&1; tail -15 $O/tests.log | cut -c1-300
