"""Where the convolution launches stand against the ceiling the clock control measured, launch shape by launch shape.

Reads the measured per-launch times of the tile table (action-detection_amd/tuned_tiles_pl.json: tools/autotune_pl.py on an MI355X at
the bench batch) and prices, for the forward and data-gradient launches of the planes kernels, three separable losses against
CEILING_TF (the algorithmic rate of conv_pl's loop body at the clock the chip sustains under it):

  padding       MFMAs issued on rows / pixels / channels that the tile shape adds (M to BM, pixels to BN, channels to 16)
  last round    workgroup slots left empty in the last round of the launch (512 slots = 2 workgroups x 256 CUs; 1 for the big tiles)
  residual      everything else: per-workgroup prologue / epilogue, waits, clock below the control's

    python tools/layer_efficiency.py [fwd|dgrad|wgrad] [--top N]      (wgrad: measured time against the ceiling only)
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CEILING_TF = 475.0
# tile shapes of conv_pl.hip (kPlBM / kPlBN) and workgroups per CU of each
BM = [128, 64, 128, 64, 192, 256, 128, 96, 160, 32, 64, 192]
BN = [128, 128, 64, 64, 128, 128, 256, 128, 128, 128, 256, 64]
PER_CU = [2, 2, 2, 2, 2, 1, 1, 2, 2, 2, 2, 2]


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "fwd"
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    table = json.load(open(os.path.join(ROOT, "action-detection_amd", "tuned_tiles_pl.json")))
    n = table["n_images"]
    sys.path.insert(0, ROOT)
    import torch
    from action_detection_amd.bninception import BNInception
    net = BNInception(in_channels=3)
    net.eval()                           # frozen BatchNorm: the plan of the benchmarked configuration
    plan, shapes = net._plan(torch.zeros(1, 3, 224, 224))
    rows = []
    for op in plan:
        if op["kind"] != "conv" or op["src"] == "data":      # (the stem runs as 4x4 taps on the space-to-depth input)
            continue
        kh, kw = op.get("kh", op["k"]), op.get("kw", op["k"])
        cin, cout, s = op["cin"], op["cout"], op["s"]
        hin = shapes[op["src"]][1]
        ho = shapes[op["dst"]][1]
        key = "%s|%d|%d|%d|%d|%d|%d" % (kind, cin, cout, kh, kw, s, hin)
        if key not in table["tiles"] or (kind == "dgrad" and s == 2):      # (stride-2 dgrads: four parity-class launches)
            continue
        tile, ms = table["tiles"][key], table["ms"][key]
        if kind == "wgrad":              # (split-K kernels: no tile-padding / round model here, time against the ceiling only)
            flop = 2.0 * n * ho * ho * cin * cout * kh * kw
            t_ideal = flop / (CEILING_TF * 1e12) * 1e3
            rows.append(dict(key=key, tile=tile, ms=ms, tf=flop / ms / 1e9, ideal=t_ideal, pad=0.0, last=0.0, resid=ms - t_ideal, wgs=0,
                             rounds=0))
            continue
        if kind == "fwd":
            m, c, pix = cout, cin, n * ho * ho
        else:                            # dgrad: rows = input channels, reduction over output channels, pixels of the INPUT
            m, c, pix = cin, cout, n * hin * hin
        t = tile - 32 if tile >= 32 else tile
        flop = 2.0 * n * ho * ho * cin * cout * kh * kw
        mt, pt = math.ceil(m / BM[t]), math.ceil(pix / BN[t])
        issued = 2.0 * (mt * BM[t]) * (pt * BN[t]) * (math.ceil(c / 16) * 16) * kh * kw
        slots = 256 * PER_CU[t]
        wgs = mt * pt
        rounds = math.ceil(wgs / slots)
        t_ideal = flop / (CEILING_TF * 1e12) * 1e3
        t_pad = issued / (CEILING_TF * 1e12) * 1e3
        t_round = t_pad * (rounds * slots) / wgs
        rows.append(dict(key=key, tile=tile, ms=ms, tf=flop / ms / 1e9, ideal=t_ideal, pad=t_pad - t_ideal, last=t_round - t_pad,
                         resid=ms - t_round, wgs=wgs, rounds=rounds))
    tot = {f: sum(r[f] for r in rows) for f in ("ms", "ideal", "pad", "last", "resid")}
    print("%s launches of the BN-Inception plan (%d launches, %d images; per-launch times of the tile table): %.2f ms measured; at %.0f TF "
          "they would take %.2f ms" % (kind, len(rows), n, tot["ms"], CEILING_TF, tot["ideal"]))
    print("  tile padding %.2f ms (%.0f %%), empty slots of the last round %.2f ms (%.0f %%), residual %.2f ms (%.0f %%)"
          % (tot["pad"], 100 * tot["pad"] / tot["ms"], tot["last"], 100 * tot["last"] / tot["ms"], tot["resid"],
             100 * tot["resid"] / tot["ms"]))
    print("%-26s %5s %8s %7s | %7s %7s %7s %7s | %6s %6s" % ("shape (cin|cout|k|k|s|h)", "tile", "ms", "TF", "ideal", "pad", "last", "resid",
                                                          "wgs", "rounds"))
    for r in sorted(rows, key=lambda r: -(r["ms"] - r["ideal"]))[:top]:
        print("%-26s %5d %8.4f %7.1f | %7.4f %7.4f %7.4f %7.4f | %6d %6d" % (r["key"][len(kind) + 1:], r["tile"], r["ms"], r["tf"], r["ideal"],
                                                                          r["pad"], r["last"], r["resid"], r["wgs"], r["rounds"]))


if __name__ == "__main__":
    main()
