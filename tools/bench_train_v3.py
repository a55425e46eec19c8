"""Training step of SSN on the Inception-v3 backbone on one MI355X (the reference trains on it with
`ssn_train.py <dataset> <modality> --arch InceptionV3`; /root/reference/ssn_models.py:133-139, ssn_train.py:205-253).

Not the headline metric (bench.py: BN-Inception, BASELINE.json configs[1]); the same step -- forward, three losses, backward,
SGD -- at 299x299 on synthetic frames, eager launches, timed with HIP events.  Prints one JSON line.

  python tools/bench_train_v3.py [--videos 2] [--steps 5] [--warmup 2]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa: E402,F401
from action_detection_amd.inceptionv3_spec import build_manifest, conv_macs  # noqa: E402
from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss  # noqa: E402
from action_detection_amd.optim import SSNSGD  # noqa: E402
from action_detection_amd.ssn_models import SSN  # noqa: E402
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--num-class", type=int, default=100)
ap.add_argument("--families", action="store_true", help="per-launch HIP events: time per conv kernel family")
ap.add_argument("--layers", type=int, default=0, help="with --families: also list the N slowest launches on stderr")
args = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = SSN(args.num_class, 2, 5, 2, "RGB", base_model="InceptionV3", dropout=0.8, stpp_cfg=(1, 1, 1))
init_backbone_synthetic(model.base_model)
init_heads_synthetic(model)
model.to(dev).train()
opt = SSNSGD(model.get_optim_policies(), lr=1e-5, momentum=0.9, weight_decay=5e-4)     # (synthetic weights: keep the steps small)
batch = [t.to(dev) for t in make_batch(args.videos, "RGB", args.num_class, seed=0, input_size=299)]
crit = (ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss())


def step():
    out = model(*batch)
    loss = crit[0](out[0], out[1]) + 0.1 * crit[1](out[2], out[3], 1, 7) + 0.1 * crit[2](out[4], out[5], out[6])
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


first_loss = None
for _ in range(args.warmup):
    l_ = step()
    first_loss = l_.item() if first_loss is None else first_loss
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(args.steps):
    loss = step()
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / args.steps
frames = args.videos * 8 * 9
ops, shapes = build_manifest(3, 299)
macs = conv_macs(ops, shapes)
first = next(op for op in ops if op[0] == "conv")
first_macs = shapes[first[3]][1] * shapes[first[3]][2] * first[5] * first[6] * first[7] * first[8]
gflop = 2.0 * (3 * macs - first_macs) * frames / 1e9          # fwd + dgrad + wgrad, no data gradient for the first layer
line = {"metric": "ssn_inceptionv3_train_proposals_per_s", "value": round(args.videos * 8 / (ms * 1e-3), 1), "unit": "proposals/s",
        "ms_per_step": round(ms, 3), "frames_per_step": frames, "conv_tflops": round(gflop / ms, 2), "first_step_loss": None if first_loss is None else round(first_loss, 5), "loss": round(loss.item(), 5),
        "dtype": "f32 (f16 x 3 split MFMA)", "data": "synthetic", "steps": args.steps, "warmup": args.warmup,
        "config": {"workload": "InceptionV3 RGB SSN, %d videos x 8 proposals x 9 segments (299x299), fwd + losses + bwd + SGD, eager"
                   % args.videos}}
if args.families:
    prof = []
    model.base_model.profiler = prof
    step()
    torch.cuda.synchronize()
    model.base_model.profiler = None
    fam = {}
    for family, lid, flops, s, e in prof:
        f = fam.setdefault(family, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += s.elapsed_time(e)
        f[2] += 1
    if args.layers:
        rows = sorted(((s.elapsed_time(e), family, lid, flops) for family, lid, flops, s, e in prof), reverse=True)
        for ms_, family, lid, flops in rows[:args.layers]:
            print("%8.3f ms %7.1f TF  %-16s %s" % (ms_, flops / (ms_ * 1e-3) / 1e12, family, lid), file=sys.stderr)
    line["families"] = {k: {"launches": v[2], "ms": round(v[1], 3), "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2)} for k, v in fam.items()}
print(json.dumps(line))
