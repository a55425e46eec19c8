"""profiles/r6_summary.md from the evidence run of tools/gpu_r6_evidence.sh (gpurun_out/r6ev/): the step by kernel family (rocprofv3 kernel stats of
the eager single-stream run), the bench lines, the PMC summary.

    python tools/r6_summary.py [steps of the profiled run = 63]
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r6ev")
FRAMES = 288
FWD = 4.0632e9 * FRAMES                      # algorithmic flops of the 69 convolutions, forward
DGRAD = FWD - 2 * 118.0e6 * FRAMES * 1.0     # ... minus conv1's data gradient (118.0 MMAC per frame)
FAMILIES = [
    # (by kernel name the two directions cannot be told apart: the stride-2 data gradients run as four parity-class launches of the
    # FORWARD instantiation; bench.json's roofline_detail has the split from per-launch events)
    ("conv forward + dgrad (conv_pl_kernel, conv_pl9_kernel)", r"conv_pl9?_kernel<", FWD + DGRAD),
    ("wgrad: nine-tap, rows <= 14", r"wgrad_group9_kernel<6,", None),
    ("wgrad: nine-tap, rows <= 30", r"wgrad_group9_kernel<8,", None),
    ("wgrad: nine-tap, rows <= 56 (conv2)", r"wgrad_group9_kernel<12,", None),
    ("wgrad: one-tap / chunked 1x1", r"wgrad_group1_kernel", None),
    ("wgrad: stem", r"wgrad_group_stem_kernel", None),
    ("wgrad: reduction + table writes + 7x7 gather", r"wgrad_reduce|wgrad_group_write|s2d_weight_bwd", None),
    ("max pools forward", r"pl_maxpool_fwd", None),
    ("max pools backward", r"pl_maxpool_bwd", None),
    ("average pools (+ affine), ReLU/BN backward, global pool, channel sums", r"pl_avgpool|pl_relu_bn|pl_gap|pl_channel_sum", None),
    ("frames -> planes, scales, range check", r"pl_from_f32|tensor_amax|scales_update|range_check|s2d_kernel", None),
    ("weight packing, BN fold", r"pack_x6|bn_fold", None),
    ("STPP / heads / losses / dropout", r"heads_|ce_|completeness|cw_smooth|stpp|dropout|row_|total_loss|label_select", None),
    ("SGD", r"sgd_", None),
]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 63
    rows = list(csv.DictReader(open(os.path.join(O, "kernel_stats_eager.csv"))))
    acc = {name: [0.0, 0] for name, _, _ in FAMILIES}
    other = [0.0, 0]
    others = {}
    for r in rows:
        nm = r["Name"]
        for name, rx, _ in FAMILIES:
            if re.search(rx, nm):
                acc[name][0] += float(r["TotalDurationNs"])
                acc[name][1] += int(r["Calls"])
                break
        else:
            other[0] += float(r["TotalDurationNs"])
            other[1] += int(r["Calls"])
            others[nm[:60]] = others.get(nm[:60], 0) + float(r["TotalDurationNs"])
    total = sum(v[0] for v in acc.values()) + other[0]
    out = ["# Round 6: the step by kernel family", "",
           "rocprofv3 --kernel-trace --stats of `bench.py --no-graph --single-stream --steps 60 --warmup 3` (%d profiled steps, one stream:"
           % steps, "every kernel alone on the GPU; the replayed hipGraph overlaps the two branch lanes and runs ~8 %% shorter than this sum).",
           "Algorithmic flops: 2 x MACs of the 69 convolutions (SURVEY 8d), forward %.1f GFLOP, dgrad %.1f, wgrad %.1f per step; TF ="
           % (FWD / 1e9, DGRAD / 1e9, FWD / 1e9), "those / time; frac = TF / 833.3 (three f16 MFMA products per multiply).", "",
           "| family | launches / step | ms / step | share | TF | frac of 833 |", "|---|---|---|---|---|---|"]
    wg_ms = 0.0
    for name, _, fl in FAMILIES:
        ns, calls = acc[name]
        ms = ns / 1e6 / steps
        if name.startswith("wgrad"):
            wg_ms += ms
        tf = ("%.0f" % (fl / (ms * 1e-3) / 1e12)) if (fl and ms > 0) else ""
        fr = ("%.3f" % (fl / (ms * 1e-3) / 1e12 / 833.3)) if (fl and ms > 0) else ""
        out.append("| %s | %.1f | %.3f | %.1f %% | %s | %s |" % (name, calls / steps, ms, 100 * ns / total, tf, fr))
    out.append("| everything else (torch micro-kernels, copies) | %.1f | %.3f | %.1f %% | | |" % (other[1] / steps, other[0] / 1e6 / steps,
                                                                                                100 * other[0] / total))
    out.append("| **sum** | | **%.3f** | | | |" % (total / 1e6 / steps))
    out += ["", "All weight-gradient rows together: %.3f ms per step = %.0f TF = %.3f of 833 (round 5: 4.69 ms = 250 TF)."
            % (wg_ms, FWD / (wg_ms * 1e-3) / 1e12, FWD / (wg_ms * 1e-3) / 1e12 / 833.3), ""]
    try:
        d = json.loads([l for l in open(os.path.join(O, "bench.json")) if l.startswith("{")][-1])
        rd = d.get("roofline_detail", {})
        out += ["Per-launch HIP events of the default run (`bench.json`: `roofline_detail`; same kernels, eager, one stream, right after the",
                "timed region): " + "; ".join("%s %.3f ms = %.0f TF" % (k, v["ms_per_step"], v["tflops"]) for k, v in rd.items()
                                               if isinstance(v, dict) and k.endswith("_all")) + ".", ""]
    except (OSError, IndexError, ValueError, KeyError):
        pass
    for f, title in (("bench.json", "default line (config 2)"), ("bench_400steps.json", "400 timed steps (sustained)"),
                     ("bench_flow.json", "Flow (config 3)"), ("bench_dist1_separate.json", "1-rank RCCL, separate collectives"),
                     ("bench_dist1_overlapped.json", "1-rank RCCL, overlapped collectives"),
                     ("bench_train_inceptionv3.json", "Inception-v3 training, 2 videos"),
                     ("bench_dense_inceptionv3.json", "dense test, Inception-v3"), ("bench_dense_bninception.json", "dense test, BN-Inception")):
        try:
            d = json.loads([l for l in open(os.path.join(O, f)) if l.startswith("{")][-1])
        except (OSError, IndexError, ValueError):
            continue
        line = "* %s: **%.1f %s**, %.3f ms per step" % (title, d["value"], d["unit"], d["ms_per_step"])
        if "roofline" in d and d["roofline"]:
            r = d["roofline"]
            line += "; roofline %s %.1f / %.1f %s = %.3f" % (r.get("bound"), r.get("achieved", 0), r.get("peak", 0), r.get("unit"), r.get("frac", 0))
        if d.get("cpu_baseline"):
            c = d["cpu_baseline"]
            line += "; cpu_baseline %.2f %s on %s cores (%s)" % (c["value"], c["unit"], c["cores"], c["kind"])
        out.append(line)
    try:
        pm = json.load(open(os.path.join(O, "pmc_summary.json")))
        out += ["", "PMC (tools/gpu_pmc.sh, separate passes; `pmc_summary.json`):", "",
                "| family | launches sampled | MFMA pipe busy | non-MFMA instr per MFMA | HBM MB per launch |", "|---|---|---|---|---|"]
        for fam, v in pm.items():
            if not isinstance(v, dict) or "launches_sampled" not in v:
                continue
            out.append("| %s | %s | %s | %s | %s |" % (fam, v.get("launches_sampled"), v.get("mfma_pipe_busy_frac", ""),
                                                      v.get("non_mfma_insts_per_mfma", v.get("valu_salu_per_mfma", "")),
                                                      ("%.0f" % (v["hbm_bytes_per_launch"] / 1e6)) if v.get("hbm_bytes_per_launch") else ""))
    except (OSError, ValueError):
        pass
    if others:
        out += ["", "(largest 'everything else' kernels: " + "; ".join("%s %.3f ms" % (k, v / 1e6 / steps) for k, v in
                                                                        sorted(others.items(), key=lambda kv: -kv[1])[:6]) + ")"]
    boxes = []
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.startswith("r6_bench_box") or name == "r6_bench_final.json":
            try:
                d = json.loads([l for l in open(os.path.join(ROOT, "profiles", name)) if l.startswith("{")][-1])
                boxes.append((d["ms_per_step"], d["value"], name))
            except (OSError, ValueError, IndexError, KeyError):
                pass
    if boxes:
        boxes.sort()
        med = boxes[len(boxes) // 2][0] if len(boxes) % 2 else 0.5 * (boxes[len(boxes) // 2 - 1][0] + boxes[len(boxes) // 2][0])
        out += ["", "Boxes of the round (default `bench.py` line, 20 timed steps, hipGraph replay; `r6_bench_final.json` and `r6_bench_boxK4.json` are the CLOSING tree, `boxK` / `K2` / `K3` trees before the last two changes of DESIGN 5.2 (-0.4 ms); `r6_bench_final.json` is the evidence box "
                "every other file of this summary comes from): " + "; ".join("`%s` %.3f ms = %.1f proposals/s" % (n, ms, v) for ms, v, n in boxes)
                + ".  Median %.2f ms." % med]
    with open(os.path.join(ROOT, "profiles", "r6_summary.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
