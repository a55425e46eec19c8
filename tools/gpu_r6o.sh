#!/bin/bash
# round 6, call O: Inception-v3 shapes of the tile table re-tuned on cold operands (forward + dgrad), then the dense-test and training
# lines with the old and the new table alternating.
O=gpurun_out/r6; mkdir -p $O
T=action-detection_amd/tuned_tiles_pl.json
cp $T /tmp/old_table.json
COLD=1 KINDS=fwd,dgrad timeout 1800 python tools/autotune_pl.py 288 InceptionV3 > $O/o_autotune.txt 2> $O/o_autotune.err; tail -2 $O/o_autotune.txt
cp $T /tmp/new_table.json; cp $T $O/o_tuned_tiles_pl.json
for rep in 1 2; do for which in old new; do
  cp /tmp/${which}_table.json $T
  timeout 600 python bench.py --mode dense-test --arch InceptionV3 --steps 7 --warmup 1 --proposal-list tests/golden/proposal_list_processed.txt --cpu-baseline-videos 0 > $O/o_dense_${which}_$rep.json 2>/dev/null
  timeout 600 python bench.py --arch InceptionV3 --videos-per-gpu 2 --steps 5 --warmup 2 --cpu-baseline-videos 0 --no-kernel-events > $O/o_train_${which}_$rep.json 2>/dev/null
  python - $O/o_dense_${which}_$rep.json $O/o_train_${which}_$rep.json $which $rep <<'PY'
import json, sys
def line(p):
    try:
        d = json.loads([l for l in open(p) if l.startswith("{")][-1]); return "%.1f %s (%.3f ms)" % (d["value"], d["unit"], d["ms_per_step"])
    except Exception as e:
        return "no line"
print("%s #%s  dense %s | train %s" % (sys.argv[3], sys.argv[4], line(sys.argv[1]), line(sys.argv[2])))
PY
done; done 2>&1 | tee $O/o_ab.txt
cp /tmp/new_table.json $T
echo "O: done at ${SECONDS}s"
