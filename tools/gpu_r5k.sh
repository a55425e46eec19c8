#!/bin/bash
# round 5, call K (the last GPU seconds of the round): one-tap weight-gradient body with incremental fetch addresses.  Parity on the GPU,
# ONE alternating pair against the previous library, and -- only if the weight-gradient family got faster by more than run-to-run noise
# and enough of the call's time is left -- the evidence stages that depend on the library (kernel stats, PMC, 400 steps, launch list).
O=gpurun_out/r5; mkdir -p $O
L=action-detection_amd/libssn_hip.so
timeout 60 python -m pytest tests/test_planes.py -m gpu -x -q -k "wgrad_group" > $O/k_parity.txt 2>&1; tail -2 $O/k_parity.txt
if ! grep -q " passed" $O/k_parity.txt || grep -q "failed\|error" $O/k_parity.txt; then echo "K: PARITY NOT GREEN -> nothing adopted"; exit 0; fi
echo "K: parity green at ${SECONDS}s"
cp $L /tmp/new.so
run() { timeout 60 python bench.py --cpu-baseline-videos 0 > $O/k_$1.json 2> $O/k_$1.err; }
cp tools/.ab/libssn_prev.so $L; run prev
cp /tmp/new.so $L; run new
python - $O/k_prev.json $O/k_new.json $SECONDS <<'PY' | tee $O/k_decision.txt
import json, sys
def load(p):
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    rd = d["roofline_detail"]
    return d["ms_per_step"], {k: v["ms_per_step"] for k, v in rd.items() if isinstance(v, dict) and k.endswith("_all")}
(ms0, f0), (ms1, f1) = load(sys.argv[1]), load(sys.argv[2])
print("prev %.3f ms/step %s" % (ms0, f0))
print("new  %.3f ms/step %s" % (ms1, f1))
gain = f0["conv_wgrad_all"] - f1["conv_wgrad_all"]
ok = gain > 0.08 and ms1 < ms0 + 0.05 and int(sys.argv[3]) < 62
print("weight-gradient family: %+.3f ms; elapsed %ss -> %s" % (-gain, sys.argv[3], "ADOPT" if ok else "KEEP THE PREVIOUS LIBRARY"))
PY
if grep -q ADOPT $O/k_decision.txt; then
  cp $O/k_new.json $O/bench_closing.json
  STAGES=prof,pmc,bench400,seq bash tools/gpu_r5.sh
else
  cp tools/.ab/libssn_prev.so $L
fi
echo "K: done at ${SECONDS}s"
