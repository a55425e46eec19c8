"""Over-fetch of the one-tap / chunked weight-gradient family (wgrad_group1_kernel), problem by problem: every problem of the BN-Inception
plan launched ALONE through the grouped entry, then the whole family in one grid.  Under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
the dispatches of wgrad_group1_kernel appear in the order written to <out>.order; `parse` joins the two (2 x FETCH_SIZE KiB = bytes read
past the L2, see tools/pmc_summary.py) with the algorithmic operand bytes.  Without rocprof the same launches are timed with events.

    python tools/pmc_wgrad_alone.py run <out prefix> [n_images]         (optionally under rocprofv3)
    python tools/pmc_wgrad_alone.py parse <out prefix> <pmc dir>
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problems(n, dev):
    import torch
    from action_detection_amd import planes as P
    from action_detection_amd.bninception import BNInception
    net = BNInception(in_channels=3)
    net.eval()
    plan, shapes = net._plan(torch.zeros(1, 3, 224, 224))
    g = torch.Generator().manual_seed(0)
    jobs, keys, alg, cache = [], [], [], {}
    for op in plan:
        if op["kind"] != "conv" or op["src"] == "data":
            continue
        kh, kw, ph, pw = op.get("kh", op["k"]), op.get("kw", op["k"]), op.get("ph", op["p"]), op.get("pw", op["p"])
        cin, cout, s = op["cin"], op["cout"], op["s"]
        hin = shapes[op["src"]][1]
        _, ho, wo = shapes[op["dst"]]
        key = "%d|%d|%d|%d|%d|%d" % (cin, cout, kh, kw, s, hin)
        if key not in cache:
            x = torch.randn(n, cin, hin, hin, generator=g).clamp(min=0).to(dev)
            gy = (torch.randn(n, cout, ho, wo, generator=g) * 1e-3).to(dev)
            cache[key] = (P.from_f32(x), P.from_f32(gy))
        xp, gp = cache[key]
        hint = net._pl_tile("wgradg", op, n, shapes)
        jobs.append(P.WgradJob(P.pfull(gp), P.pfull(xp), torch.empty(cout, cin, kh, kw, device=dev), torch.empty(cout, device=dev),
                               kh, kw, s, ph, pw, hint=hint))
        keys.append(key)
        alg.append(4.0 * n * (cin * hin * hin + cout * ho * wo))       # both planes of both operands, once
    _, _, plan_all = P.wgrad_group_plan(jobs)
    idx = [i for i in range(len(jobs)) if plan_all[i][0] == 3]
    return P, jobs, keys, alg, plan_all, idx


def run(prefix, n):
    import torch
    import action_detection_amd as pkg
    pkg.build()
    dev = torch.device("cuda:0")
    P, jobs, keys, alg, plan_all, idx = problems(n, dev)
    order = []

    def launch(sub_idx, reps):
        sub = [jobs[i] for i in sub_idx]
        ws_b, tb_b, _ = P.wgrad_group_plan(sub)
        ws = torch.empty(ws_b // 4 + 4, device=dev)
        tb = torch.empty(tb_b, device=dev, dtype=torch.uint8)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        P.conv_wgrad_group(sub, ws, tb)
        s.record()
        for _ in range(reps):
            P.conv_wgrad_group(sub, ws, tb)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps, reps + 1

    for i in idx:
        ms, k = launch([i], 3)
        order += [{"label": keys[i], "variant": plan_all[i][1], "alg_bytes": alg[i], "ms": ms}] * k
        print("%-22s variant %d  alone %.4f ms  alg %.0f MB = %.2f TB/s" % (keys[i], plan_all[i][1], ms, alg[i] / 1e6, alg[i] / ms / 1e9), flush=True)
    ms, k = launch(idx, 3)
    order += [{"label": "family", "variant": -1, "alg_bytes": sum(alg[i] for i in idx), "ms": ms}] * k
    print("family: %.4f ms, alg %.0f MB" % (ms, sum(alg[i] for i in idx) / 1e6), flush=True)
    with open(prefix + ".order", "w") as f:
        json.dump(order, f)


def parse(prefix, pmc_dir):
    with open(prefix + ".order") as f:
        order = json.load(f)
    rows = []
    for path in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if ("wgrad_group1_kernel" in row["Kernel_Name"] or "wgrad_gang1_kernel" in row["Kernel_Name"]) and row["Counter_Name"] == "FETCH_SIZE":
                    rows.append((int(row.get("Dispatch_Id") or len(rows)), float(row["Counter_Value"])))
    rows.sort()
    print("%d dispatches in the counter file, %d expected" % (len(rows), len(order)))
    agg = {}
    for (_, v), o in zip(rows, order):
        a = agg.setdefault((o["label"], o["variant"]), [0.0, 0, o])
        a[0] += 2.0 * v * 1024
        a[1] += 1
    for (label, variant), (tot, k, o) in agg.items():
        b = tot / k
        print("%-22s variant %2d  read past L2 %7.0f MB = %.2f x algorithmic (%5.0f MB)   %.4f ms -> %.2f TB/s of L2 misses"
              % (label, variant, b / 1e6, b / o["alg_bytes"], o["alg_bytes"] / 1e6, o["ms"], b / o["ms"] / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 288)
    else:
        parse(sys.argv[2], sys.argv[3])
