"""Does the input side keep up with the training step?  (the reference's `Data` meter,
/root/reference/ssn_train.py:194,262 -- the time the loop waits for its next batch.)

Drives `input_pipeline.TrainingBatchPrefetcher` from host-resident uint8 frames into the real training step -- by default
UN-CROPPED decoded frames (256 x 340 x 3, what a loader worker holds right after JPEG decoding): the scale-jittered crop, PIL's
bilinear resize, the flip and the normalisation all run on the GPU (`GpuTrainAugment`, round 4); `--precropped` = the round-3
measurement (frames already cropped / resized to 224 x 224 by PIL on the host) -- (SSN forward, losses,
backward, SGD on the MI355X, replayed as one hipGraph on a static batch the prefetched one is copied into) and reports, per step:

  * host_wait_ms   -- time `next(prefetcher)` blocks the training loop's thread (staging / upload not ready yet),
  * stream_wait_ms -- time the compute stream stalls on the batch's ready event (HIP events around the wait),
  * step_ms with the prefetcher vs. with the same batch resident in HBM, and the frame rate the pipeline sustained.

    python tools/bench_input_pipeline.py [--steps 20] [--videos 4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd.input_pipeline import GpuFrameTransform, GpuTrainAugment, TrainingBatchPrefetcher  # noqa: E402
from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss  # noqa: E402
from action_detection_amd.optim import SSNSGD  # noqa: E402
from action_detection_amd.ssn_models import SSN  # noqa: E402
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--videos", type=int, default=4)
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--precropped", action="store_true", help="host frames already at 224 x 224 (crop / resize done by PIL on the host)")
ap.add_argument("--decoded-size", type=int, nargs=2, default=[256, 340], help="H W of the decoded frames")
args = ap.parse_args()
pkg.build()
dev = torch.device("cuda:0")
v, num_class = args.videos, 20
torch.manual_seed(0)
model = SSN(num_class, 2, 5, 2, "RGB", dropout=0.8, stpp_cfg=(1, 1, 1))
init_backbone_synthetic(model.base_model)
init_heads_synthetic(model, std=0.001)
model.to(dev).train()
opt = SSNSGD(model.get_optim_policies(), lr=0.001, momentum=0.9, weight_decay=5e-4)
crit = (ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss())
_, scaling, target, reg_target, prop_type = make_batch(v, "RGB", num_class, seed=0)
rs = np.random.RandomState(0)
# a few distinct host batches (uint8, as decoded), cycled: the content does not change the work, the copies are real
fh, fw = (224, 224) if args.precropped else args.decoded_size
host = [rs.randint(0, 256, size=(v, 72, fh, fw, 3), dtype=np.uint8) for _ in range(3)]


def source(n):
    for i in range(n):
        yield host[i % len(host)], scaling, target, reg_target, prop_type


def train_step(batch):
    out = model(*batch)
    loss = crit[0](out[0], out[1]) + 0.1 * crit[1](out[2], out[3], 1, 7) + 0.1 * crit[2](out[4], out[5], out[6])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


if args.precropped:
    tf, group = GpuFrameTransform(224, model.input_mean, model.input_std, roll=True, device=dev), None
else:       # SSN.get_augmentation() for RGB: GroupMultiScaleCrop(224, [1, .875, .75, .66]) + flip; one box per proposal = 9 frames
    tf, group = GpuTrainAugment(224, model.input_mean, model.input_std, [1, .875, .75, .66], roll=True, device=dev), 9
# ---- the step as one hipGraph on a STATIC batch (how bench.py runs it): a prefetched batch is copied into the static input
# (173 MB device-to-device, ~0.06 ms) and the graph replayed -- the training thread then issues two calls per step, so the
# staging thread's Python work does not compete with ~400 eager launches for the interpreter lock
static = [t.to(dev) for t in make_batch(v, "RGB", num_class, seed=0)]
launch = "hipGraph replay"
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(args.warmup):
        train_step(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    train_step(static)


def run(batch):
    if batch is not static:
        for dst, src in zip(static, batch):
            dst.copy_(src, non_blocking=True)
    graph.replay()


for _ in range(args.warmup):
    run(static)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    run(static)
torch.cuda.synchronize()
resident_ms = 1e3 * (time.perf_counter() - t0) / args.steps

# ---- through the prefetcher
pf = TrainingBatchPrefetcher(source(args.warmup + args.steps), tf, depth=args.depth, group_size=group)
host_wait, ev = [], []
stream = torch.cuda.current_stream(dev)
t_start = None
for i in range(args.warmup + args.steps):
    if i == args.warmup:
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    e0.record(stream)
    batch = next(pf)              # queue wait on the host, then stream.wait_event(batch ready)
    e1.record(stream)
    h1 = time.perf_counter()
    if i >= args.warmup:
        host_wait.append(1e3 * (h1 - h0))
        ev.append((e0, e1))
    run(batch)
torch.cuda.synchronize()
pipe_ms = 1e3 * (time.perf_counter() - t_start) / args.steps
pf.close()
stream_wait = [a.elapsed_time(b) for a, b in ev]
frames_per_step = v * 72
res = {
    "metric": "input pipeline: wait per training step (the reference's Data meter, ssn_train.py:194,262)",
    "step_ms_resident": round(resident_ms, 3), "step_ms_with_prefetcher": round(pipe_ms, 3),
    "host_wait_ms_per_step": round(float(np.mean(host_wait)), 4), "host_wait_ms_max": round(float(np.max(host_wait)), 4),
    "stream_wait_ms_per_step": round(float(np.mean(stream_wait)), 4), "stream_wait_ms_max": round(float(np.max(stream_wait)), 4),
    "wait_frac_of_step": round(float(np.mean(stream_wait)) / pipe_ms, 5),
    "frames_per_s_sustained": round(frames_per_step / (pipe_ms * 1e-3), 1),
    "frames_per_s_needed_by_resident_step": round(frames_per_step / (resident_ms * 1e-3), 1),
    "bytes_per_step_over_pcie": int(frames_per_step * fh * fw * 3),
    "host_frames": "%d x %d x 3 uint8 (%s)" % (fh, fw, "cropped + resized by PIL on the host" if args.precropped else
                                               "as decoded: crop + PIL-exact bilinear resize + flip + normalise on the GPU"),
    "config": {"videos": v, "frames_per_step": frames_per_step, "depth": args.depth, "steps": args.steps, "launch": launch,
               "layout": model.base_model.layout,
               "note": "uint8 frames staged in pinned memory by a background thread, uploaded and normalised on a side stream "
                       "while the compute stream trains on the previous batch"},
}
print(json.dumps(res))
