"""3x3 / stride-2 max-pool forward on the bench batch's four pool shapes, COLD operands (rotating over buffer sets > 1.6 GB): ms per launch
and TB/s of algorithmic bytes (input once + output + argmax).  SSN_POOL_BANDS=0 in a second process times the per-output kernel.

    python tools/bench_pool_fwd.py [n_images]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa: E402,F401
from action_detection_amd import planes as P  # noqa: E402

SHAPES = [("pool1", 64, 112), ("pool2", 192, 56), ("3c pool", 320, 28), ("4e pool", 608, 14)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    dev = torch.device("cuda:0")
    action_detection_amd.build()
    g = torch.Generator().manual_seed(0)
    print("SSN_POOL_BANDS=%s" % os.environ.get("SSN_POOL_BANDS", "1"))
    for name, c, h in SHAPES:
        ho = -(-(h - 3) // 2) + 1
        by = n * c * (h * h * 4 + ho * ho * 5)
        R = max(2, int(1.6e9 / by) + 1)
        x = torch.randn(n, c, h, h, generator=g).clamp(min=0).to(dev)
        sets = []
        for _ in range(R):
            xp = P.from_f32(x)
            y = P.PlaneTensor(n, c, ho, ho, dev)
            am = torch.zeros((n, c // 8, ho * ho, 8), dtype=torch.uint8, device=dev)
            sets.append((xp, y, am))
        fns = [lambda s=s: P.maxpool_fwd(P.pfull(s[0]), P.pfull(s[1]), s[2], 3, 2, 0) for s in sets]
        for f in fns:
            f()
        for s in sets:
            s[1].pool.update()
        for f in fns:
            f()
        torch.cuda.synchronize()
        reps = 4 * R
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(reps):
            fns[i % R]()
        en.record()
        torch.cuda.synchronize()
        ms = st.elapsed_time(en) / reps
        chk = float(P.to_f32(sets[0][1]).double().sum()), int(sets[0][2].long().sum())
        print("%-8s %4d ch %3d -> %3d  %7.1f MB  %d sets  %.4f ms  %.2f TB/s   checksum %.6f / %d" % (name, c, h, ho, by / 1e6, R, ms, by / ms / 1e9, chk[0], chk[1]), flush=True)
        del sets, fns


if __name__ == "__main__":
    main()
