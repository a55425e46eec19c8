"""Per-layer variants of the GROUPED weight-gradient launches (ssn_conv_wgrad_pl_group) for the BN-Inception plan at the bench
batch, tuned IN the group: the whole group of the plan's non-nine-tap problems is timed (its launches share the GPU, so a layer's
best tile depends on what runs beside it), then one problem at a time is switched to each variant it can take and the change kept
when the group gets faster (coordinate descent, two sweeps).  Writes "wgradg|<shape key>": hint into tuned_tiles_pl.json
(read by BNInception._pl_tile("wgradg", ...)) and a log of every measurement.

    python tools/autotune_wgrad_group.py [n_images]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import planes as P  # noqa: E402

OUT = os.path.join(ROOT, "action-detection_amd", "tuned_tiles_pl.json")


def timeit(fn, reps=8, warm=2):
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
    pkg.build()
    dev = torch.device("cuda:0")
    from action_detection_amd.bninception import BNInception
    net = BNInception(in_channels=3)
    net.eval()
    plan, shapes = net._plan(torch.zeros(1, 3, 224, 224))
    g = torch.Generator().manual_seed(0)
    jobs, keys, flops = [], [], []
    cache = {}
    for op in plan:
        if op["kind"] != "conv" or op["src"] == "data":
            continue
        kh, kw = op.get("kh", op["k"]), op.get("kw", op["k"])
        ph, pw = op.get("ph", op["p"]), op.get("pw", op["p"])
        cin, cout, s = op["cin"], op["cout"], op["s"]
        hin = shapes[op["src"]][1]
        _, ho, wo = shapes[op["dst"]]
        key = "%d|%d|%d|%d|%d|%d" % (cin, cout, kh, kw, s, hin)
        if key not in cache:      # (layers of one shape share their operands: the timing does not care)
            x = torch.randn(n, cin, hin, hin, generator=g).clamp(min=0).to(dev)
            gy = (torch.randn(n, cout, ho, wo, generator=g) * 1e-3).to(dev)
            cache[key] = (P.from_f32(x), P.from_f32(gy))
        xp, gp = cache[key]
        dw, db = torch.empty(cout, cin, kh, kw, device=dev), torch.empty(cout, device=dev)
        jobs.append(P.WgradJob(P.pfull(gp), P.pfull(xp), dw, db, kh, kw, s, ph, pw))
        keys.append(key)
        flops.append(2.0 * n * ho * wo * cout * cin * kh * kw)
    nine = [i for i, j in enumerate(jobs) if (j.kh, j.kw, j.stride, j.pad_h, j.pad_w) == (3, 3, 1, 1, 1)]
    rest = [i for i in range(len(jobs)) if i not in nine]
    log = []

    def say(msg):
        print(msg, flush=True)
        log.append(msg)

    def run(idx):
        sub = [jobs[i] for i in idx]
        ws_b, tb_b, _ = P.wgrad_group_plan(sub)
        ws = torch.empty(ws_b // 4 + 4, device=dev)
        tb = torch.empty(tb_b, device=dev, dtype=torch.uint8)
        return timeit(lambda: P.conv_wgrad_group(sub, ws, tb))

    t9 = run(nine)
    say("nine-tap problems (%d, %.0f GFLOP): %.4f ms = %.0f TF" % (len(nine), sum(flops[i] for i in nine) / 1e9, t9,
                                                                 sum(flops[i] for i in nine) / t9 / 1e9))
    base = run(rest)
    say("other problems (%d, %.0f GFLOP), library's choice: %.4f ms = %.0f TF" % (len(rest), sum(flops[i] for i in rest) / 1e9, base,
                                                                                sum(flops[i] for i in rest) / base / 1e9))
    # whole-family baselines: every problem on one variant (where it can take it)
    for hint in (3, 8, 0, 200):
        ok = True
        for i in rest:
            one = (jobs[i].kh, jobs[i].kw, jobs[i].stride) == (1, 1, 1)
            jobs[i].hint = hint if (hint < 200 or one) else 3
        try:
            t = run(rest)
            say("  all on hint %3d: %.4f ms" % (hint, t))
        except RuntimeError as e:
            say("  all on hint %3d: %s" % (hint, e))
    for i in rest:
        jobs[i].hint = -1
    best = base
    chosen = {i: -1 for i in rest}
    for sweep in range(2):
        for i in sorted(rest, key=lambda i: -flops[i]):
            one = (jobs[i].kh, jobs[i].kw, jobs[i].stride) == (1, 1, 1)
            cands = [3, 8, 0] + ([200] if one else [])
            res = {}
            for h in cands:
                if h == chosen[i]:
                    continue
                jobs[i].hint = h
                res[h] = run(rest)
            hb = min(res, key=res.get)
            line = "sweep %d %-24s now %3d (group %.4f ms) | " % (sweep, keys[i], chosen[i], best) + " ".join(
                "%d:%.4f" % (h, v) for h, v in sorted(res.items()))
            if res[hb] < best * 0.997:
                chosen[i], best = hb, res[hb]
                line += " -> %d" % hb
            jobs[i].hint = chosen[i]
            say(line)
    final = run(rest)
    say("tuned: %.4f ms (library's choice %.4f) = %.0f TF" % (final, base, sum(flops[i] for i in rest) / final / 1e9))
    try:
        with open(OUT) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {"tiles": {}, "ms": {}, "n_images": n}
    for k in [k for k in table["tiles"] if k.startswith("wgradg|")]:
        del table["tiles"][k]
    if final < base:
        for i in rest:
            if chosen[i] >= 0:
                table["tiles"]["wgradg|" + keys[i]] = chosen[i]
    for path in (OUT, os.path.join(ROOT, "gpurun_out", "tuned_tiles_pl.json")):
        if os.path.isdir(os.path.dirname(path)):
            with open(path, "w") as f:
                json.dump(table, f, indent=0, sort_keys=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r5", "autotune_wgrad_group.txt"), "w") as f:
        f.write("\n".join(log) + "\n")


if __name__ == "__main__":
    main()
