"""Planes-layout convolution kernel (csrc/conv_pl.hip) against the fp32-layout split kernel (csrc/conv_x6.hip) on the
BN-Inception layer shapes at the bench batch (288 frames): per layer, forward and dgrad, every tile config of the new
kernel next to the tuned tile of the old one.  Output: one line per (layer, direction), TFLOP/s of algorithmic fp32 work.

    python tools/bench_conv_pl.py [n_images] [fwd|dgrad|both] [layer-substring]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd  # noqa: E402,F401
from action_detection_amd import _lib, kernels as K, planes as P  # noqa: E402
from action_detection_amd.bninception import tuned_tile  # noqa: E402

LAYERS = [
    # name, cin, cout, k, s, p, hin
    ("conv2_3x3", 64, 192, 3, 1, 1, 56), ("3a_block_in", 192, 224, 1, 1, 0, 28), ("3a_double_3x3_2", 96, 96, 3, 1, 1, 28),
    ("3b_3x3", 64, 96, 3, 1, 1, 28), ("3c_3x3_s2", 128, 160, 3, 2, 1, 28), ("4a_block_in", 576, 512, 1, 1, 0, 14),
    ("4a_double_3x3_2", 128, 128, 3, 1, 1, 14), ("4c_3x3", 128, 160, 3, 1, 1, 14), ("4d_double_3x3_2", 192, 192, 3, 1, 1, 14),
    ("4e_double_3x3_1", 192, 256, 3, 1, 1, 14), ("5a_block_in", 1056, 832, 1, 1, 0, 7), ("5a_3x3", 192, 320, 3, 1, 1, 7),
    ("5b_double_3x3_2", 224, 224, 3, 1, 1, 7), ("5b_pool_proj", 1024, 128, 1, 1, 0, 7), ("conv2_reduce", 64, 64, 1, 1, 0, 56),
]


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    dev = torch.device("cuda:0")
    action_detection_amd.build()
    ntiles = int(_lib.get_lib().cdll.ssn_conv_pl_tiles())
    out = {}
    g = torch.Generator().manual_seed(0)
    for (name, cin, cout, k, s, p, hin) in LAYERS:
        if filt and filt not in name:
            continue
        ho = (hin + 2 * p - k) // s + 1
        flops = 2.0 * n * ho * ho * cout * cin * k * k
        x = torch.randn(n, cin, hin, hin, generator=g).clamp(min=0).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
        scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
        shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
        wp = K.pack_weights_multi([([w], 0)], x6=True)[0]
        if which in ("fwd", "both"):
            xg = K.guarded_empty((n, cin, hin, hin), dev)
            xg.copy_(x)
            K.attach_amax(xg, K.tensor_amax(xg))
            y = K.attach_amax(torch.empty((n, cout, ho, ho), device=dev))
            t_old = timeit(lambda: K.conv_x6_fwd(K.full(xg), wp, scale, shift, K.full(y), k, s, p, True,
                                                 tuned_tile("fwd6", n, cin, cout, k, s, hin)))
            xp = P.from_f32(x)
            yp = P.PlaneTensor(n, cout, ho, ho, dev)
            res = {}
            for tile in range(ntiles):
                P.conv_fwd(P.pfull(xp), wp, scale, shift, P.pfull(yp), k, k, s, p, p, True, tile)
                yp.pool.update()
                res[tile] = timeit(lambda: P.conv_fwd(P.pfull(xp), wp, scale, shift, P.pfull(yp), k, k, s, p, p, True, tile))
            best = min(res, key=res.get)
            err = ((P.to_f32(yp) - y).abs().max() / y.abs().max()).item()
            line = dict(layer=name, dir="fwd", old_ms=round(t_old, 4), old_tf=round(flops / t_old / 1e9, 1), best_tile=best,
                        new_ms=round(res[best], 4), new_tf=round(flops / res[best] / 1e9, 1),
                        all_tf={t: round(flops / v / 1e9, 1) for t, v in res.items()}, new_vs_old_err=err)
            print(json.dumps(line), flush=True)
            out[name + "|fwd"] = line
        if which in ("dgrad", "both") and s == 1:
            gy = (torch.randn(n, cout, ho, ho, generator=g) * 1e-3).to(dev)
            wt = K.pack_weights_multi([([w], 1)], x6=True)[0]
            gg = K.guarded_empty((n, cout, ho, ho), dev)
            gg.copy_(gy)
            K.attach_amax(gg, K.tensor_amax(gg))
            dx = K.attach_amax(torch.empty((n, cin, hin, hin), device=dev))
            msc = (torch.rand(cin, generator=g) + 0.5).to(dev)
            t_old = timeit(lambda: K.conv_x6_dgrad(K.full(gg), wt, K.full(dx), k, p, False,
                                                   tuned_tile("dgrad6", n, cin, cout, k, s, hin), mask_y=K.full(x), mask_scale=msc))
            gp = P.from_f32(gy)
            xp = P.from_f32(x)
            dxp = P.PlaneTensor(n, cin, hin, hin, dev)
            res = {}
            for tile in range(ntiles):
                P.conv_dgrad(P.pfull(gp), wt, P.pfull(dxp), k, k, p, p, False, tile, mask=P.pfull(xp), mask_scale=msc)
                dxp.pool.update()
                res[tile] = timeit(lambda: P.conv_dgrad(P.pfull(gp), wt, P.pfull(dxp), k, k, p, p, False, tile, mask=P.pfull(xp),
                                                        mask_scale=msc))
            best = min(res, key=res.get)
            err = ((P.to_f32(dxp) - dx).abs().max() / dx.abs().max()).item()
            line = dict(layer=name, dir="dgrad", old_ms=round(t_old, 4), old_tf=round(flops / t_old / 1e9, 1), best_tile=best,
                        new_ms=round(res[best], 4), new_tf=round(flops / res[best] / 1e9, 1),
                        all_tf={t: round(flops / v / 1e9, 1) for t, v in res.items()}, new_vs_old_err=err)
            print(json.dumps(line), flush=True)
            out[name + "|dgrad"] = line
        if which in ("wgrad", "both"):
            gy = (torch.randn(n, cout, ho, ho, generator=g) * 1e-3).to(dev)
            dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
            t_old = None
            if s == 1:
                gg = K.guarded_empty((n, cout, ho, ho), dev)
                gg.copy_(gy)
                K.attach_amax(gg, K.tensor_amax(gg))
                xg = K.guarded_empty((n, cin, hin, hin), dev, 256)
                xg.copy_(x)
                K.attach_amax(xg, K.tensor_amax(xg))
                wcfg = tuned_tile("wgrad6", n, cin, cout, k, s, hin)
                ws = torch.empty(K.wgrad_x6_workspace_bytes(n, cin, cout, hin, hin, k, wcfg) // 4, device=dev)
                t_old = timeit(lambda: K.conv_wgrad_x6(K.full(gg), K.full(xg), dw, db, k, p, ws, wcfg))
                dw_old = dw.clone()
            gp, xp = P.from_f32(gy), P.from_f32(x)
            nt = int(_lib.get_lib().cdll.ssn_conv_wgrad_pl_tiles())
            res = {}
            for tile in list(range(nt)) + ([100, 101, 102, 103] if (k, s, p) == (3, 1, 1) else []):
                ws2 = torch.empty(P.wgrad_workspace_bytes(n, cin, cout, ho, ho, k, k, tile) // 4, device=dev)
                res[tile] = timeit(lambda: P.conv_wgrad(P.pfull(gp), P.pfull(xp), dw, db, k, k, s, p, p, ws2, tile))
            best = min(res, key=res.get)
            ws2 = torch.empty(P.wgrad_workspace_bytes(n, cin, cout, ho, ho, k, k, best) // 4, device=dev)
            P.conv_wgrad(P.pfull(gp), P.pfull(xp), dw, db, k, k, s, p, p, ws2, best)
            err = ((dw - dw_old).abs().max() / dw_old.abs().max()).item() if t_old else None
            line = dict(layer=name, dir="wgrad", old_ms=round(t_old, 4) if t_old else 0.0,
                        old_tf=round(flops / t_old / 1e9, 1) if t_old else None, best_tile=best, new_ms=round(res[best], 4),
                        new_tf=round(flops / res[best] / 1e9, 1), all_tf={t: round(flops / v / 1e9, 1) for t, v in res.items()},
                        new_vs_old_err=err)
            print(json.dumps(line), flush=True)
            out[name + "|wgrad"] = line
    tot_old = sum(v["old_ms"] for v in out.values())
    tot_new = sum(v["new_ms"] for v in out.values())
    print(json.dumps(dict(total_old_ms=round(tot_old, 3), total_new_ms=round(tot_new, 3))))


if __name__ == "__main__":
    main()
