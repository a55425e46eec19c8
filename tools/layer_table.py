#!/usr/bin/env python
"""Per-layer conv timing table on the GPU (HIP events around every launch): shape, tile, ms, TFLOP/s."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import action_detection_amd as pkg
from action_detection_amd.ssn_models import SSN
from action_detection_amd.synthetic import init_backbone_synthetic, make_batch
from action_detection_amd.bninception_spec import build_manifest
from action_detection_amd import _lib
pkg.build()
dev = torch.device("cuda:0")
v = int(os.environ.get("VIDEOS", "4"))
m = SSN(20, 2, 5, 2, "RGB", dropout=0.8, stpp_cfg=(1, 1, 1))
init_backbone_synthetic(m.base_model)
m.to(dev).train()
m.base_model.overlap_wgrad = False   # one kernel at a time: clean per-launch timings
m.base_model.branch_streams = False
x = make_batch(v, "RGB", 20, seed=0)[0].to(dev).reshape(-1, 3, 224, 224)
plan, shapes = m.base_model._plan(x[:1])
info = {}
for op in plan:
    if op["kind"] == "conv":
        info[op["lids"][0]] = (op["cin"], op["cout"], op["k"], op["s"], shapes[op["src"]][1], shapes[op["dst"]][1])
def run():
    f = m.base_model.features(x)
    f.sum().backward()
for _ in range(2):
    run()
prof = []
m.base_model.profiler = prof
reps = 3
for _ in range(reps):
    run()
torch.cuda.synchronize()
agg = {}
for fam, lid, flops, s, e in prof:
    base, path = fam.rsplit("_", 1)          # conv_fwd_x6 -> (conv_fwd, x6)
    a = agg.setdefault((lid, base), [0.0, flops, path])
    a[0] += s.elapsed_time(e) / reps
lib = _lib.get_lib()
n = x.shape[0]
print("%-34s %5s %5s %2s %2s %4s | %-4s %8s %6s | %8s %6s | %8s %6s" % ("layer", "cin", "cout", "k", "s", "Ho", "tile", "fwd_ms", "TF", "dgrad_ms", "TF", "wgrad_ms", "TF"))
tot = {"conv_fwd": 0, "conv_dgrad": 0, "conv_wgrad": 0}
for lid, (cin, cout, k, s, hi, ho) in info.items():
    tile = lib.cdll.ssn_conv_pick_tile(cout, n * ho * ho)
    row = []
    for fam in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        a = agg.get((lid, fam))
        if a:
            tot[fam] += a[0]
            row += ["%8.3f" % a[0], "%5.0f%s" % (a[1] / a[0] / 1e9, "*" if a[2] == "x6" else " ")]
        else:
            row += ["       -", "     -"]
    print("%-34s %5d %5d %2d %2d %4d | %-4d %s %s | %s %s | %s %s" % ((lid, cin, cout, k, s, ho, tile) + tuple(row)))
print("totals ms:", tot, "(* = split-operand (2 x f16, 3 products) x6 kernel, otherwise exact-f32 MFMA kernel)")
