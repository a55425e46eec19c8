#!/usr/bin/env python
"""Ablate the conv kernel's phases on a few layers (GPU): which part keeps the MFMA pipe from peak?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import action_detection_amd as pkg
from action_detection_amd import kernels as K, _lib
pkg.build()
lib = _lib.get_lib()
dev = torch.device("cuda:0")
n = 288
cases = [("conv2_3x3", 64, 192, 3, 1, 1, 56), ("4c_3x3", 128, 160, 3, 1, 1, 14), ("4a_1x1", 576, 224, 1, 1, 0, 14),
         ("5a_3x3", 192, 320, 3, 1, 1, 7), ("3a_3x3", 64, 64, 3, 1, 1, 28)]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best
for name, cin, cout, k, s, p, h in cases:
    x = torch.randn(n, cin, h, h, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    y = torch.empty(n, cout, h, h, device=dev); sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    wp = K.pack_weights(w, False)
    flops = 2.0 * n * h * h * cout * cin * k * k
    for cfg in (0, 2, 3, 1):
        row = []
        for flags, label in ((0, "full"), (1, "noGload"), (3, "noGload+noLDSst"), (7, "+nobarrier")):
            lib.cdll.ssn_conv_debug_flags(flags)
            ms = timeit(lambda: K.conv_fwd(K.full(x), wp, sc, sh, K.full(y), k, s, p, True, cfg))
            row.append("%s %.3fms %.0fTF" % (label, ms, flops / ms / 1e9))
        lib.cdll.ssn_conv_debug_flags(0)
        print(name, "cfg%d" % cfg, " | ".join(row), flush=True)
