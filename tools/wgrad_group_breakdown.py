"""Marginal cost of every problem of the grouped weight-gradient launches in its group (BN-Inception plan, bench batch): the group
is timed whole and with one problem left out; the difference is what that problem costs BESIDE the others (its launch shares the GPU
with them).  Printed next to the problem's algorithmic flops: TF per problem, per family.

    python tools/wgrad_group_breakdown.py [n_images] [BNInception|InceptionV3]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import planes as P  # noqa: E402


def timeit(fn, reps=10, warm=2):
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
    arch = sys.argv[2] if len(sys.argv) > 2 else "BNInception"
    if arch == "InceptionV3":
        from action_detection_amd.inceptionv3 import InceptionV3
        net = InceptionV3()
        net.eval()
        plan, shapes = net._plan(torch.zeros(1, 3, 299, 299))
    else:
        from action_detection_amd.bninception import BNInception
        net = BNInception(in_channels=3)
        net.eval()
        plan, shapes = net._plan(torch.zeros(1, 3, 224, 224))
    g = torch.Generator().manual_seed(0)
    jobs, keys, flops, cache = [], [], [], {}
    for op in plan:
        if op["kind"] != "conv" or (op["src"] == "data" and arch != "InceptionV3"):
            continue
        kh, kw, ph, pw = op.get("kh", op["k"]), op.get("kw", op["k"]), op.get("ph", op["p"]), op.get("pw", op["p"])
        cin, cout, s = op["cin"], op["cout"], op["s"]
        hin = shapes[op["src"]][1] if op["src"] != "data" else 299
        _, ho, wo = shapes[op["dst"]]
        key = "%d|%d|%d|%d|%d|%d" % (cin, cout, kh, kw, s, hin)
        if key not in cache:
            x = torch.randn(n, cin, hin, hin, generator=g).clamp(min=0).to(dev)
            gy = (torch.randn(n, cout, ho, wo, generator=g) * 1e-3).to(dev)
            cache[key] = (P.from_f32(x), P.from_f32(gy))
        xp, gp = cache[key]
        jobs.append(P.WgradJob(P.pfull(gp), P.pfull(xp), torch.empty(cout, cin, kh, kw, device=dev), torch.empty(cout, device=dev),
                               kh, kw, s, ph, pw))
        keys.append(key)
        flops.append(2.0 * n * ho * wo * cout * cin * kh * kw)
    _, _, plan_all = P.wgrad_group_plan(jobs)
    fam_name = {0: "nine-tap rows<=14", 1: "nine-tap rows<=30", 2: "nine-tap rows<=56", 3: "one-tap / chunked", 4: "stem"}

    def run(idx):
        sub = [jobs[i] for i in idx]
        ws_b, tb_b, _ = P.wgrad_group_plan(sub)
        ws = torch.empty(ws_b // 4 + 4, device=dev)
        tb = torch.empty(tb_b, device=dev, dtype=torch.uint8)
        return timeit(lambda: P.conv_wgrad_group(sub, ws, tb))

    for fam in sorted({p[0] for p in plan_all}):
        idx = [i for i in range(len(jobs)) if plan_all[i][0] == fam]
        t_all = run(idx)
        fl = sum(flops[i] for i in idx)
        print("family %d (%s): %d problems, %.0f GFLOP, %.4f ms = %.0f TF (incl. table writes and the reduction)"
              % (fam, fam_name[fam], len(idx), fl / 1e9, t_all, fl / t_all / 1e9), flush=True)
        if len(idx) < 2:
            continue
        for i in sorted(idx, key=lambda i: -flops[i]):
            t = run([j for j in idx if j != i])
            d = t_all - t
            print("   %-22s variant %d splits %3d x %4d units  %6.1f GFLOP  marginal %.4f ms = %4.0f TF  (at 475 TF: %.4f ms)"
                  % (keys[i], plan_all[i][1], plan_all[i][2], plan_all[i][3], flops[i] / 1e9, d, flops[i] / max(d, 1e-6) / 1e9,
                     flops[i] / 475e9), flush=True)


if __name__ == "__main__":
    main()
