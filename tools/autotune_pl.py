"""Tile tables of the planes kernels (conv_pl forward / dgrad, wgrad_pl one-tap and nine-tap) for the launch plans of the
backbones at the bench batch: every launch shape of the plan is timed with every tile config on the MI355X and the fastest
is written to action-detection_amd/tuned_tiles_pl.json (read by bninception.BNInception._pl_tile).

    python tools/autotune_pl.py [n_images] [BNInception|InceptionV3]

COLD=1 (round 6): forward / dgrad launches are timed rotating over R independent operand sets (R x (input + output) > 1.2 GB, beyond the
256 MiB Infinity Cache), as they run inside the step -- the same buffers 12 times in a row keep the operands on die and pick tiles that
lose 6 % inside the step (profiles/r6_cold_operands.txt).
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import _lib, kernels as K, planes as P  # noqa: E402

OUT = os.path.join(ROOT, "action-detection_amd", "tuned_tiles_pl.json")


def timeit(fn, reps=12, warm=2):
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


def timeit_rot(fns, reps=12, warm=1):
    """mean ms per call of fns[0], fns[1], ... called round robin (cold operands)"""
    if len(fns) == 1:
        return timeit(fns[0], reps)
    reps = max(reps, 2 * len(fns))
    for i in range(warm * len(fns)):
        fns[i]()
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
    cold = os.environ.get("COLD") == "1"
    arch = sys.argv[2] if len(sys.argv) > 2 else "BNInception"
    pkg.build()
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    nfwd = int(lib.cdll.ssn_conv_pl_tiles())
    nwg = int(lib.cdll.ssn_conv_wgrad_pl_tiles())
    shapes_done = {}
    plans = []
    if arch == "BNInception":
        from action_detection_amd.bninception import BNInception
        for cin in (3, 10):
            net = BNInception(in_channels=cin)
            net.eval()      # frozen BatchNorm (what SSN.train() leaves): the plan with the fused block-input launches
            plans.append(net._plan(torch.zeros(1, cin, 224, 224)))
    else:
        from action_detection_amd.inceptionv3 import InceptionV3
        net = InceptionV3()
        net.eval()
        plans.append(net._plan(torch.zeros(1, 3, 299, 299)))
    try:
        with open(OUT) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {}
    tiles, ms = table.get("tiles", {}), table.get("ms", {})
    kinds = os.environ.get("KINDS", "fwd,dgrad,wgrad").split(",")      # KINDS=wgrad: re-tune the weight gradients only
    g = torch.Generator().manual_seed(0)
    for plan, shapes in plans:
        for op in plan:
            if op["kind"] != "conv":
                continue
            kh, kw = op.get("kh", op["k"]), op.get("kw", op["k"])
            ph, pw = op.get("ph", op["p"]), op.get("pw", op["p"])
            cin, cout, s = op["cin"], op["cout"], op["s"]
            hin = shapes[op["src"]][1]
            _, ho, wo = shapes[op["dst"]]
            key = "%d|%d|%d|%d|%d|%d" % (cin, cout, kh, kw, s, hin)
            if key in shapes_done:
                continue
            shapes_done[key] = True
            stem = op["src"] == "data" and (kh, s) == (7, 2)
            if stem:       # the space-to-depth form: 4x4 taps on 4C (padded to 16) channels at half the size
                xc, xh, k_, s_, p_ = 4 * cin, hin // 2, 4, 1, 2
            else:
                xc, xh, k_, s_, p_ = cin, hin, kh, s, ph
            rect = (kh != kw) or kh not in (1, 3, 7)
            halo = [b + c for b in (32, 48) for c in range(nfwd) if (kh, kw, s, ph, pw) == (3, 3, 1, 1, 1)
                    and lib.cdll.ssn_conv_pl_halo_taken(n, hin, hin, b + c) == 1]      # the haloed 3x3 kernel (48 + c: per-image tiles)
            if os.environ.get("HALO_ONLY") and not halo:
                continue
            if os.environ.get("ONLY") and os.environ["ONLY"] not in key:
                continue
            x = torch.randn(n, xc, xh, xh, generator=g).clamp(min=0).to(dev)
            w = (torch.randn(cout, xc, k_, k_ if not rect else kw, generator=g) * 0.05).to(dev)
            sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            R = 1
            if cold:
                R = max(2, int(1.2e9 / (4.0 * n * (xc * xh * xh + cout * ho * wo))) + 1)
            xps = [P.from_f32(x) for _ in range(R)]
            yps = [P.PlaneTensor(n, cout, ho, wo, dev) for _ in range(R)]
            xp, yp = xps[0], yps[0]
            xs = P.PSlice(xp, 0, xp.g * 8)
            if rect or stem:
                wp = K.pack_weights_rect(w)
            else:
                wp = K.pack_weights_multi([([w], 0)], x6=True)[0]
            kw_ = k_ if not rect else kw
            pw_ = p_ if not rect else pw
            # ---- forward
            res = {}
            for t in ((list(range(nfwd)) + halo) if "fwd" in kinds else []):
                fns = [(lambda i=i: P.conv_fwd(P.PSlice(xps[i], 0, xps[i].g * 8), wp, sc, sh, P.pfull(yps[i]), k_, kw_, s_, p_, pw_, True, t))
                       for i in range(R)]
                for i in range(R):
                    fns[i]()
                    yps[i].pool.update()
                res[t] = timeit_rot(fns)
            line = "%-28s" % key
            if res:
                best = min(res, key=res.get)
                tiles["fwd|" + key], ms["fwd|" + key] = best, round(res[best], 4)
                line += " fwd tile %2d %.4f ms" % (best, res[best])
                if halo:
                    line += " (plain %.4f)" % min(v for t, v in res.items() if t < 32)
                if stem:
                    line += " all " + " ".join("%d:%.3f" % (t, v) for t, v in sorted(res.items()))
            # ---- dgrad (not for the first layer)
            gy = (torch.randn(n, cout, ho, wo, generator=g) * 1e-3).to(dev)
            gps = [P.from_f32(gy) for _ in range(R if (op["src"] != "data" and "dgrad" in kinds) else 1)]
            gp = gps[0]
            if op["src"] != "data" and "dgrad" in kinds:
                dxps = [P.PlaneTensor(n, cin, hin, hin, dev) for _ in range(R)]
                msc = torch.ones(cin, device=dev)
                res = {}
                if s == 2:
                    wt = K.pack_dgrad_s2(w)
                    mk = lambda t, i: (lambda: P.conv_dgrad_s2(P.pfull(gps[i]), wt, P.pfull(dxps[i]), ph, False, t, mask=P.pfull(xps[i]), mask_scale=msc))  # noqa: E731
                elif rect:
                    wt = K.pack_dgrad_rect(w)
                    mk = lambda t, i: (lambda: P.conv_dgrad(P.pfull(gps[i]), wt, P.pfull(dxps[i]), kh, kw, ph, pw, False, t, mask=P.pfull(xps[i]),  # noqa: E731
                                                            mask_scale=msc, taps_reversed=True))
                else:
                    wt = K.pack_weights_multi([([w], 1)], x6=True)[0]
                    mk = lambda t, i: (lambda: P.conv_dgrad(P.pfull(gps[i]), wt, P.pfull(dxps[i]), kh, kw, ph, pw, False, t, mask=P.pfull(xps[i]),  # noqa: E731
                                                            mask_scale=msc))
                for t in list(range(nfwd)) + halo:
                    fns = [mk(t, i) for i in range(R)]
                    for i in range(R):
                        fns[i]()
                        dxps[i].pool.update()
                    res[t] = timeit_rot(fns)
                del dxps
                best = min(res, key=res.get)
                tiles["dgrad|" + key], ms["dgrad|" + key] = best, round(res[best], 4)
                line += " | dgrad tile %2d %.4f ms" % (best, res[best])
                if halo:
                    line += " (plain %.4f)" % min(v for t, v in res.items() if t < 32)
            # ---- wgrad
            if "wgrad" not in kinds:
                print(line, flush=True)
                del xps, yps, gps
                continue
            dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
            res = {}
            cand = list(range(nwg)) + ([100, 101, 102, 103] if (k_, kw_, s_, p_, pw_) == (3, 3, 1, 1, 1) and ho == xh and xh <= 56 else [])
            cand += [200, 201, 202, 203] if (k_, kw_, s_, p_, pw_) == (1, 1, 1, 0, 0) else []
            for t in cand:
                ws = torch.empty(P.wgrad_workspace_bytes(n, xc, cout, ho, wo, k_, kw_, t) // 4 + 4, device=dev)
                res[t] = timeit(lambda: P.conv_wgrad(P.pfull(gp), xs, dw, db, k_, kw_, s_, p_, pw_, ws, t, cin=xc))
            best = min(res, key=res.get)
            tiles["wgrad|" + key], ms["wgrad|" + key] = best, round(res[best], 4)
            line += " | wgrad tile %3d %.4f ms" % (best, res[best])
            print(line, flush=True)
    for path in (OUT, os.path.join(ROOT, "gpurun_out", "tuned_tiles_pl.json")):      # (gpurun_out/ is what travels back from the GPU box)
        if os.path.isdir(os.path.dirname(path)):
            with open(path, "w") as f:
                # (the table's batch-size gate stays BN-Inception's)
                json.dump({"n_images": n if arch == "BNInception" else table.get("n_images", n), "tiles": tiles, "ms": ms}, f, indent=0, sort_keys=True)
    print("wrote", OUT, "fwd %.3f dgrad %.3f wgrad %.3f ms (one launch per distinct shape)" % tuple(
        sum(v for k, v in ms.items() if k.startswith(p + "|")) for p in ("fwd", "dgrad", "wgrad")))


if __name__ == "__main__":
    main()
