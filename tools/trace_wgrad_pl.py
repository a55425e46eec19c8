"""Where does a weight-gradient launch on planes tensors spend its time?  Per-block phase stamps of wgrad_pl_kernel /
wgrad_pl9_kernel (csrc/wgrad_pl.hip, the `trace` pointer): prologue (setup + first operand fetch issued), k-loop, epilogue
(partial-slab stores), against the 100 MHz real-time counter; per CU: how many blocks were co-resident and for how long.

    python tools/trace_wgrad_pl.py          (on the MI355X)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import _lib, planes as P  # noqa: E402

pkg.build()
ABL = os.path.join(ROOT, "tools", ".trace", "libssn_hip_ablate.so")     # tools/build_ablate_lib.sh: the ablation switches exist there only
if os.path.exists(ABL):
    _lib.use_library_for_testing(_lib.SsnLibrary(ABL))
lib = _lib.get_lib()
dev = torch.device("cuda:0")
n = 288
# name, cin, cout, k, h, tile configs (100+: nine-tap, 200+: chunked 1x1, else one-tap)
CASES = [("4d_d3x3_2", 192, 192, 3, 14, [100, 103]), ("3b_3x3", 64, 96, 3, 28, [100, 103]), ("3a_d3x3_2", 96, 96, 3, 28, [100, 103]),
         ("conv2_3x3", 64, 192, 3, 56, [100, 103]), ("5a_3x3", 192, 320, 3, 7, [100, 103]), ("3a_block_in", 192, 192, 1, 28, [1, 3, 6]),
         ("4a_block_in", 576, 512, 1, 14, [3, 6, 7]), ("5b_block_in", 1024, 736, 1, 7, [8, 3])]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, cin, cout, k, h, tiles in CASES:
    p = k // 2
    x = torch.randn(n, cin, h, h, device=dev).clamp(min=0)
    g = torch.randn(n, cout, h, h, device=dev)
    xp, gp = P.from_f32(x), P.from_f32(g)
    dw = torch.empty(cout, cin, k, k, device=dev)
    db = torch.empty(cout, device=dev)
    flops = 2.0 * n * h * h * cout * cin * k * k
    for tile in tiles:
        ws = torch.empty(P.wgrad_workspace_bytes(n, cin, cout, h, h, k, k, tile) // 4, device=dev)
        fn = lambda: P.conv_wgrad(P.pfull(gp), P.pfull(xp), dw, db, k, k, 1, p, p, ws, tile)  # noqa: E731
        t_full = timeit(fn)
        abl = []
        if os.path.exists(ABL):
            for flags, label in ((1, "fetch nothing"), (4, "no fetch instructions"), (2, "X reads of every 3rd step only"), (6, "both")):
                lib.cdll.ssn_conv_wgrad_pl_debug_flags(flags)
                abl.append("%s %.4f" % (label, timeit(fn)))
            lib.cdll.ssn_conv_wgrad_pl_debug_flags(0)
        nmax = 1 << 16
        tr = torch.zeros(nmax * 8, dtype=torch.int64, device=dev)
        lib.cdll.ssn_conv_wgrad_pl_debug_trace(ctypes.c_void_p(tr.data_ptr()))
        fn()
        torch.cuda.synchronize()
        lib.cdll.ssn_conv_wgrad_pl_debug_trace(ctypes.c_void_p(0))
        t = tr.cpu().numpy().reshape(nmax, 8)
        t = t[t[:, 0] > 0]
        cu = (t[:, 5] & 0xF) * 256 + ((t[:, 4] >> 8) & 0xFF)
        rt0, rt1 = t[:, 6].astype(np.float64), t[:, 7].astype(np.float64)
        life = rt1 - rt0
        big = life > 0
        tick_ghz = float(((t[big, 3] - t[big, 0]) / life[big]).mean() * 0.1)
        kern_us = float(rt1.max() - rt0.min()) * 0.01
        first_us = (rt0 - rt0.min()) * 0.01
        peaks, busy1, busy2, nper = [], [], [], []
        for c in np.unique(cu):
            sel = np.where(cu == c)[0]
            ev = sorted([(rt0[i], 1) for i in sel] + [(rt1[i], -1) for i in sel])
            live = peak = 0
            b1 = b2 = 0.0
            last = ev[0][0]
            for when, d in ev:
                if live >= 1:
                    b1 += when - last
                if live >= 2:
                    b2 += when - last
                last = when
                live += d
                peak = max(peak, live)
            peaks.append(peak); busy1.append(b1 * 0.01); busy2.append(b2 * 0.01); nper.append(len(sel))
        peaks, busy1, busy2, nper = map(np.array, (peaks, busy1, busy2, nper))
        print("%s tile %d: %.4f ms (wgrad kernel + reduce) = %.1f TF; %d blocks on %d CUs (%d..%d per CU), tick %.3f GHz; kernel %.1f us "
              "by real-time stamps; block starts: median %.1f us, last %.1f us; per CU: peak co-resident mean %.2f max %d, >=1 block "
              "%.1f us, >=2 blocks %.1f us" % (name, tile, t_full, flops / t_full / 1e9, len(t), len(peaks), nper.min(), nper.max(),
                                               tick_ghz, kern_us, float(np.median(first_us)), float(first_us.max()), peaks.mean(),
                                               peaks.max(), busy1.mean(), busy2.mean()))
        print("    per block (ticks): prologue %.0f  loop %.0f  epilogue %.0f  total %.0f = %.1f us" % (
            (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 0]).mean(),
            float(life.mean()) * 0.01), flush=True)
        if abl:
            print("    ablations (ms): " + " | ".join(abl), flush=True)
