#!/usr/bin/env python
"""Wave-priority experiments on the x6 conv kernel (GPU): dbg bit 5 = static priority by wave slot, bit 6 = per-block."""
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
cases = [("3b_d3x3_2", 96, 96, 3, 1, 1, 28, 2), ("4b_d3x3_2", 128, 128, 3, 1, 1, 14, 5), ("conv2_3x3", 64, 192, 3, 1, 1, 56, 6),
         ("4c_red", 576, 256, 1, 1, 0, 14, 5), ("4d_d3x3_2", 192, 192, 3, 1, 1, 14, 2), ("5b_3x3", 192, 320, 3, 1, 1, 7, 2),
         ("3c_red", 320, 192, 1, 1, 0, 28, 2), ("4b_d3x3_2", 128, 128, 3, 1, 1, 14, 0)]
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best
for name, cin, cout, k, s, p, h, cfg in cases:
    x = K.guarded_empty((n, cin, h, h), dev).normal_(); w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    y = torch.empty(n, cout, h, h, device=dev); sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    (wp,) = K.pack_weights_multi([([w], 0)], x6=True)
    flops = 2.0 * n * h * h * cout * cin * k * k
    row = []
    for flags, label in ((0, "base"), (32, "static"), (64, "perblock"), (96, "both")):
        lib.cdll.ssn_conv_x6_debug_flags(flags)
        ms = timeit(lambda: K.conv_x6_fwd(K.full(x), wp, sc, sh, K.full(y), k, s, p, True, cfg))
        row.append("%s %.3f %.0fTF" % (label, ms, flops / ms / 1e9))
    lib.cdll.ssn_conv_x6_debug_flags(0)
    print(name, "cfg%d" % cfg, " | ".join(row), flush=True)
