#!/usr/bin/env python
"""Pick the fastest tile configuration per distinct conv shape on the MI355X (fwd / dgrad / wgrad).

Writes action-detection_amd/tuned_tiles.json; the executor (bninception.py) looks shapes up there
and falls back to the C-side heuristic for unknown shapes.  Run on the GPU box:
    python tools/autotune.py [N_IMAGES] [KINDS] [ARCH]
ARCH = InceptionV3: add the square-tap (1x1 / 3x3) launches of the Inception-v3 plan at 299x299 to the existing table (split
kernels only; its rectangular-tap layers use the C-side heuristic).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import kernels as K  # noqa: E402
from action_detection_amd.bninception import BNInception  # noqa: E402

pkg.build()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
# optional second argument: comma list of kinds to (re)tune; the other kinds keep their entries
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None     # e.g. fwd6,dgrad6,wgrad6,fwd6s2d,wgrad6s2d
arch = sys.argv[3] if len(sys.argv) > 3 else "BNInception"
out_path = os.path.join(ROOT, "action-detection_amd", "tuned_tiles.json")
shapes = {}
if arch == "InceptionV3":
    from action_detection_amd.bninception import is_rect  # noqa: E402
    from action_detection_amd.inceptionv3 import InceptionV3  # noqa: E402
    only = (only or {"fwd6", "dgrad6", "wgrad6"}) & {"fwd6", "dgrad6", "wgrad6"}
    plan, t = InceptionV3().eval()._plan(torch.zeros(1, 3, 299, 299))
    rect_shapes = {}
    for op in plan:
        if op["kind"] == "conv" and not is_rect(op) and op["k"] in (1, 3) and op["cin"] >= 16:
            shapes[(op["cin"], op["cout"], op["k"], op["s"], op["p"], t[op["src"]][1], t[op["dst"]][1])] = "+".join(op["lids"])
        elif op["kind"] == "conv" and is_rect(op):
            rect_shapes[(op["cin"], op["cout"], op["kh"], op["kw"], op["ph"], op["pw"], t[op["src"]][1])] = op["lids"][0]
for cin0 in ((3, 10) if arch == "BNInception" else ()):
    # the executor's launch plan (fused reduce convolutions included), not the raw manifest
    # (.eval(): the frozen-BatchNorm plan of SSN's default bn_mode -- with the fused block-input launches)
    plan, t = BNInception(in_channels=cin0).eval()._plan(torch.zeros(1, cin0, 224, 224))
    for op in plan:
        if op["kind"] == "conv":
            shapes[(op["cin"], op["cout"], op["k"], op["s"], op["p"], t[op["src"]][1], t[op["dst"]][1])] = \
                "+".join(op["lids"])


if os.environ.get("AUTOTUNE_RECT_ONLY"):      # Inception-v3: only the rectangular-tap launches
    shapes = {}
if os.environ.get("AUTOTUNE_HIN"):       # only the layers at these input sizes, e.g. AUTOTUNE_HIN=7
    keep = {int(v) for v in os.environ["AUTOTUNE_HIN"].split(",")}
    shapes = {k: v for k, v in shapes.items() if k[5] in keep}


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


prev = json.load(open(out_path)) if (only and os.path.exists(out_path)) else {}
table, times = prev.get("tiles", {}), prev.get("ms", {})
report = []
for (cin, cout, k, s, p, hi, ho), lid in sorted(shapes.items()):
    x = K.guarded_empty((n, cin, hi, hi), dev).normal_()
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    y = torch.empty(n, cout, ho, ho, device=dev)
    g = K.guarded_empty((n, cout, ho, ho), dev).normal_()
    K.attach_amax(x, K.tensor_amax(x))     # operand scales of the split kernels: measured once, outside the timed calls
    K.attach_amax(g, K.tensor_amax(g))
    scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    lay = K.dgrad_layout(k, s, p, hi, hi)
    wt = K.pack_weights(w, lay)
    wp = K.pack_weights(w, False)
    wp6, wt6 = K.pack_weights_multi([([w], 0), ([w], 1)], x6=True) if k != 7 else (None, None)
    if (k, s, p) == (7, 2, 3):
        xs2d = K.space_to_depth2(x)
        wps2d = K.pack_weights_rect(K.s2d_weights(w))
        dw2d = torch.empty(cout, 4 * cin, 4, 4, device=dev)
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
    flops = 2.0 * n * ho * ho * cout * cin * k * k
    res = {}
    for kind, cfgs in (("fwd", [0, 1, 2, 3, 4, 5, 6, 7]), ("dgrad", [0, 1, 2, 3, 4, 5, 6, 7]),
                       ("wgrad", [0, 1, 2, 3, 4, 5, 6]), ("fwd6", list(range(22))),
                       ("dgrad6", list(range(22))), ("wgrad6", list(range(12))),
                       ("fwd6s2d", [1, 2, 3, 5, 6]), ("wgrad6s2d", [0, 3, 5, 6])):
        if only and kind not in only:
            continue
        if (kind in ("dgrad", "fwd6") and k == 7) or (kind == "dgrad6" and (k == 7 or s != 1)):
            continue
        if kind in ("fwd6s2d", "wgrad6s2d") and (k, s, p) != (7, 2, 3):
            continue
        if kind == "wgrad6" and not K.wgrad_x6_supported(k, s, p, hi, hi):
            continue
        best = (1e9, -1)
        for cfg in cfgs:
            if kind == "fwd":
                fn = lambda: K.conv_fwd(K.full(x), wp, scale, shift, K.full(y), k, s, p, True, cfg)
            elif kind == "fwd6":
                fn = lambda: K.conv_x6_fwd(K.full(x), wp6, scale, shift, K.full(y), k, s, p, True, cfg)
            elif kind == "fwd6s2d":       # the stem through its space-to-depth form (bninception.py: stem_s2d)
                fn = lambda: K.conv_x6_fwd_rect(K.full(xs2d), wps2d, scale, shift, K.full(y), 4, 4, 2, 2, True, cfg)
            elif kind == "wgrad6s2d":
                ws = torch.empty(K.wgrad_x6_workspace_bytes(n, 4 * cin, cout, ho, ho, 4, cfg) // 4, device=dev)
                fn = lambda: K.conv_wgrad_x6(K.full(g), K.full(xs2d), dw2d, db, 4, 2, ws, cfg)
            elif kind == "dgrad6":
                fn = lambda: K.conv_x6_dgrad(K.full(g), wt6, K.full(dx), k, p, False, cfg)
            elif kind == "wgrad6":
                ws = torch.empty(K.wgrad_x6_workspace_bytes(n, cin, cout, hi, hi, k, cfg) // 4, device=dev)
                fn = lambda: K.conv_wgrad_x6(K.full(g), K.full(x), dw, db, k, p, ws, cfg)
            elif kind == "dgrad":
                fn = lambda: K.conv_dgrad(K.full(g), wt, K.full(dx), k, s, p, False, cfg, wt_layout=lay)
            else:
                ws = torch.empty(K.wgrad_workspace_bytes(n, cin, cout, ho, ho, k, cfg) // 4, device=dev)
                fn = lambda: K.conv_wgrad(K.full(g), K.full(x), dw, db, k, s, p, ws, cfg)
            ms = timeit(fn)
            if ms < best[0]:
                best = (ms, cfg)
        key = "%s|%d|%d|%d|%d|%d" % (kind, cin, cout, k, s, hi)
        table[key] = best[1]
        times[key] = round(best[0], 4)
        res[kind] = (best[1], best[0], flops / best[0] / 1e9)
    report.append((lid, cin, cout, k, s, ho, res))
    print(lid, cin, cout, k, s, ho, {kk: "cfg%d %.3fms %.1fTF" % v for kk, v in res.items()}, flush=True)
# rectangular-tap launches (forward, dgrad as a forward correlation, runtime-tap wgrad): kinds "<dir>6r<kh>x<kw>"
for (cin, cout, kh, kw, ph, pw, hi), lid in sorted(rect_shapes.items()) if (arch == "InceptionV3" and not os.environ.get("AUTOTUNE_HIN")) else []:
    guard = K.wgrad_x6_rect_guard_floats(ph, pw, hi)
    x = K.guarded_empty((n, cin, hi, hi), dev, guard).normal_()
    g = K.guarded_empty((n, cout, hi, hi), dev, guard).normal_()
    K.attach_amax(x, K.tensor_amax(x))
    K.attach_amax(g, K.tensor_amax(g))
    w = torch.randn(cout, cin, kh, kw, device=dev) * 0.05
    y, dx = torch.empty_like(g), torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
    scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    wp, wt = K.pack_weights_rect(w), K.pack_dgrad_rect(w)
    res = {}
    for kind, cfgs in (("fwd6r", [1, 2, 3, 5, 6]), ("dgrad6r", [1, 2, 3, 5, 6]), ("wgrad6r", [0, 2, 3, 5, 6])):
        best = (1e9, -1)
        for cfg in cfgs:
            if kind == "fwd6r":
                fn = lambda: K.conv_x6_fwd_rect(K.full(x), wp, scale, shift, K.full(y), kh, kw, ph, pw, True, cfg)
            elif kind == "dgrad6r":
                fn = lambda: K.conv_x6_dgrad_rect(K.full(g), wt, K.full(dx), kh, kw, ph, pw, False, cfg)
            else:
                ws = torch.empty(K.wgrad_x6_rect_workspace_bytes(n, cin, cout, hi, hi, kh, kw, cfg) // 4, device=dev)
                fn = lambda: K.conv_wgrad_x6_rect(K.full(g), K.full(x), dw, db, kh, kw, ph, pw, ws, cfg)
            ms = timeit(fn)
            if ms < best[0]:
                best = (ms, cfg)
        key = "%s%dx%d|%d|%d|%d|%d|%d" % (kind, kh, kw, cin, cout, 0, 1, hi)
        table[key] = best[1]
        times[key] = round(best[0], 4)
        res[kind] = "cfg%d %.3fms %.1fTF" % (best[1], best[0], 2.0 * n * hi * hi * cout * cin * kh * kw / best[0] / 1e9)
    print(lid, cin, cout, "%dx%d" % (kh, kw), hi, res, flush=True)
n_rec = prev.get("n_images", n) if arch != "BNInception" else n      # (the table's batch-size gate stays BN-Inception's)
json.dump({"n_images": n_rec, "tiles": table, "ms": times}, open(out_path, "w"), indent=0, sort_keys=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"n_images": n_rec, "tiles": table, "ms": times}, open(os.path.join(ROOT, "gpurun_out", "tuned_tiles.json"), "w"),
          indent=0, sort_keys=True)
print("wrote", out_path)
