#!/usr/bin/env python
"""Per-video detection post-processing (eval_detection_results.py + temporal_nms) on the GPU vs the CPU oracle."""
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
import ssn_oracle as O  # noqa: E402
from action_detection_amd.detection_post import DetectionPostProcessor  # noqa: E402

pkg.build()
dev = torch.device("cuda:0")
res = []
for name, p, c, top_k, thr in (("thumos14", 700, 20, 2000, 0.2), ("activitynet1.2", 187, 100, 60, 0.6)):
    rs = np.random.RandomState(0)
    start = rs.uniform(0, 0.8, p)
    rel = np.stack([start, np.minimum(start + rs.uniform(0.02, 0.5, p), 1.0)], axis=1)
    act = rs.standard_normal((p, c + 1)).astype(np.float32) * 2
    comp = rs.standard_normal((p, c)).astype(np.float32)
    reg = (rs.standard_normal((p, c, 2)) * 0.3).astype(np.float32)
    post = DetectionPostProcessor(c, thr, top_k)
    args = (torch.from_numpy(rel).to(dev), torch.from_numpy(act).to(dev), torch.from_numpy(comp).to(dev),
            torch.from_numpy(reg).to(dev))
    post.process_video(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        out, _ = post.process_video(*args)          # includes the D2H copy of the detections
    gpu = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(5):
        ref, _ = O.detections_for_video(rel, act, comp, reg, c, thr, top_k)
    cpu = (time.perf_counter() - t0) / 5
    res.append({"shape": name, "proposals": p, "classes": c, "top_k": top_k, "nms": thr,
                "gpu_ms_per_video": round(gpu * 1e3, 3), "cpu_oracle_ms_per_video": round(cpu * 1e3, 3),
                "detections": int(sum(len(v) for v in out.values()))})
print(json.dumps({"metric": "detection post-processing, ms per video (host round trip included)", "cases": res}))
