#!/usr/bin/env python
"""Dense-testing throughput (the per-video loop of /root/reference/ssn_test.py:66-92) on one MI355X.

Synthetic ActivityNet-1.2-shape video (BASELINE.json configs[4]; `--arch InceptionV3` for its 299x299 backbone,
default BNInception): `--ticks` sampled frames x 10 crops, RGB, C = 100 classes (test_fc out = 1001), `--proposals`
proposals.  Frames are resident in HBM (the JPEG decode / crop side is outside the path).  Prints one JSON line with
frames/s (crops counted) and videos/s, next to the CPU oracle on a bounded sample of the same video.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import action_detection_amd as pkg  # noqa: E402
from action_detection_amd.dense_test import DenseTester  # noqa: E402
from action_detection_amd.ssn_models import SSN  # noqa: E402
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="BNInception", choices=["BNInception", "InceptionV3"])
ap.add_argument("--ticks", type=int, default=600)
ap.add_argument("--proposals", type=int, default=50)
ap.add_argument("--crops", type=int, default=10)
ap.add_argument("--tick-batch", type=int, default=60, help="ticks per backbone call (the reference uses 4)")
ap.add_argument("--videos", type=int, default=3)
ap.add_argument("--cpu-ticks", type=int, default=8, help="ticks in the CPU-oracle sample (0 disables)")
args = ap.parse_args()
pkg.build()
dev = torch.device("cuda:0")
num_class = 100
torch.manual_seed(0)
net = SSN(num_class, 2, 5, 2, "RGB", base_model=args.arch, test_mode=True, stpp_cfg=(1, 1, 1))
size = net.input_size
gflop_per_frame = {"BNInception": 4.063152128, "InceptionV3": 2 * 5.711168096}[args.arch]   # 2 * conv MACs
init_backbone_synthetic(net.base_model)
init_heads_synthetic(net, std=0.01)
net.prepare_test_fc()
net.to(dev).eval()
tester = DenseTester(net, num_class, stats=np.array([[0.0, 0.0], [1.0, 1.0]]), tick_batch=args.tick_batch)
g = torch.Generator().manual_seed(1)
# one crop-major batch of `tick_batch` ticks, reused for every call (the content does not change the work)
batch = (torch.randint(0, 256, (args.crops * args.tick_batch, 3, size, size), generator=g).float() - 110.0).to(dev)
n_calls = (args.ticks + args.tick_batch - 1) // args.tick_batch
ticks_total = n_calls * args.tick_batch
rs = np.random.RandomState(0)
starts = rs.randint(0, ticks_total - 8, size=args.proposals)
lens = rs.randint(2, ticks_total // 3, size=args.proposals)
pt = np.stack([np.maximum(starts - lens // 2, 0), starts, np.minimum(starts + lens, ticks_total),
               np.minimum(starts + lens + lens // 2, ticks_total)], axis=1).astype(np.int64)
sc = rs.rand(args.proposals, 2)


def one_video():
    return tester.score_video((batch for _ in range(n_calls)), ticks_total, torch.from_numpy(pt), torch.from_numpy(sc),
                              num_crop=args.crops)


one_video()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.videos):
    out = one_video()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.videos
res = {"metric": "dense-test frames/s (ssn_test.py per-video loop, %s RGB %d^2, C=100)" % (args.arch, size),
       "value": round(ticks_total * args.crops / dt, 1), "unit": "frames/s", "videos_per_s": round(1.0 / dt, 3),
       "s_per_video": round(dt, 4), "n_gpus": 1, "data": "synthetic",
       "config": {"ticks": ticks_total, "crops": args.crops, "proposals": args.proposals, "tick_batch": args.tick_batch,
                  "test_fc_out": net.test_fc.out_features},
       "fwd_tflops": round(ticks_total * args.crops * gflop_per_frame * 1e9 / dt / 1e12, 2),
       "fwd_frac_of_f32_mfma_peak": round(ticks_total * args.crops * gflop_per_frame * 1e9 / dt / 1e12 / 157.3, 4)}
if args.cpu_ticks > 0:
    import ssn_oracle as O
    oracle = O.OracleSSN(num_class, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1), base_model=args.arch)
    oracle.load_state_dict({k: v.cpu() for k, v in net.state_dict().items() if not k.startswith("test_fc")})
    oracle.prepare_test_fc()
    oracle.eval()
    nb = max(1, args.cpu_ticks // 4)
    cb = batch[:args.crops * args.tick_batch].view(args.crops, args.tick_batch, 3, size, size)[:, :4].reshape(-1, 3, size, size).cpu()
    cpt = np.clip(pt, 0, 4 * nb)
    c0 = time.perf_counter()
    r = O.dense_test_video(oracle, (cb for _ in range(nb)), 4 * nb, cpt, sc, num_class, num_crop=args.crops,
                           stats=np.array([[0.0, 0.0], [1.0, 1.0]]))
    ct = time.perf_counter() - c0
    res["cpu_baseline"] = {"value": round(4 * nb * args.crops / ct, 2), "unit": "frames/s", "cores": torch.get_num_threads(),
                           "kind": "port", "sample": "%d ticks x %d crops, reference batching (4 ticks per call), %.1f s"
                                                      % (4 * nb, args.crops, ct)}
print(json.dumps(res))
