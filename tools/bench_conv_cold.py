"""Is a forward launch slower inside the step than in the autotuner's loop because its operands are COLD?

The tile table's times come from 12 back-to-back repeats of one launch on the same buffers (operands in L2 / Infinity Cache, TLB warm);
inside the step the same forward launches of the 14 x 14 block inputs run 30-35 % longer.  This tool times a launch shape (a) on the same
buffers, (b) rotating over R independent buffer sets (R x (input + output) >> the 256 MiB Infinity Cache), every tile config.

    python tools/bench_conv_cold.py [n_images] [layer-substring]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa: E402,F401
from action_detection_amd import _lib, kernels as K, planes as P  # noqa: E402

LAYERS = [
    # name, cin, cout, k, s, p, hin
    ("4a_block_in", 576, 512, 1, 1, 0, 14), ("4d_block_in", 608, 512, 1, 1, 0, 14), ("3b_block_in", 256, 256, 1, 1, 0, 28),
    ("5a_block_in", 1056, 832, 1, 1, 0, 7), ("4d_double_3x3_2", 192, 192, 3, 1, 1, 14), ("3c_3x3_s2", 128, 160, 3, 2, 1, 28),
]


def time_rot(fns, reps, warm=2):
    """mean ms per call of fns[0], fns[1], ... called round robin"""
    for i in range(warm * len(fns)):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda:0")
    action_detection_amd.build()
    ntiles = int(_lib.get_lib().cdll.ssn_conv_pl_tiles())
    g = torch.Generator().manual_seed(0)
    for (name, cin, cout, k, s, p, hin) in LAYERS:
        if filt and filt not in name:
            continue
        ho = (hin + 2 * p - k) // s + 1
        flops = 2.0 * n * ho * ho * cout * cin * k * k
        mb = n * (cin * hin * hin + cout * ho * ho) * 4 / 1e6
        R = max(2, int(1600 / mb) + 1)
        x = torch.randn(n, cin, hin, hin, generator=g).clamp(min=0).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
        scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
        shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
        wp = K.pack_weights_multi([([w], 0)], x6=True)[0]
        xs = [P.from_f32(x) for _ in range(R)]
        ys = [P.PlaneTensor(n, cout, ho, ho, dev) for _ in range(R)]
        hot, cold = {}, {}
        halo = k == 3 and s == 1
        cfgs = list(range(ntiles)) + ([32 + c for c in (0, 1, 2, 3, 4, 7, 8, 9, 11)] if halo else [])
        for tile in cfgs:
            def mk(i, tile=tile):
                return lambda: P.conv_fwd(P.pfull(xs[i]), wp, scale, shift, P.pfull(ys[i]), k, k, s, p, p, True, tile)
            fns = [mk(i) for i in range(R)]
            for f in fns:
                f()
            for y in ys:
                y.pool.update()
            hot[tile] = time_rot(fns[:1], 24)
            cold[tile] = time_rot(fns, 4 * R)
        bh, bc = min(hot, key=hot.get), min(cold, key=cold.get)
        print(json.dumps(dict(layer=name, MB=round(mb, 1), sets=R, hot_best=bh, hot_ms=round(hot[bh], 4), hot_tf=round(flops / hot[bh] / 1e9, 1),
                              cold_best=bc, cold_ms=round(cold[bc], 4), cold_tf=round(flops / cold[bc] / 1e9, 1),
                              hot_ms_all={t: round(v, 4) for t, v in hot.items()}, cold_ms_all={t: round(v, 4) for t, v in cold.items()})),
              flush=True)
        del xs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
