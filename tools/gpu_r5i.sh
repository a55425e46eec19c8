#!/bin/bash
# round 5, call I: Inception-v3 3x3 layers re-tuned for the unrolled haloed kernel; full GPU tier + smoke at the new library
O=gpurun_out/r5; mkdir -p $O
HALO_ONLY=1 KINDS=fwd,dgrad timeout 600 python tools/autotune_pl.py 144 InceptionV3 2>&1 | tail -12
cp action-detection_amd/tuned_tiles_pl.json $O/tuned_tiles_after_v3.json
timeout 300 python tools/bench_train_v3.py > $O/v3_retuned.json 2> $O/v3_retuned.err; tail -c 400 $O/v3_retuned.json; echo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
