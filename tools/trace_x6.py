#!/usr/bin/env python
"""Per-block phase timeline of the x6 conv kernel (GPU): prologue / main loop / epilogue cycles and blocks per CU."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import action_detection_amd as pkg
from action_detection_amd import kernels as K, _lib
pkg.build()
TRACE_LIB = os.path.join(ROOT, "tools", ".trace", "libssn_hip_trace.so")
if os.path.exists(TRACE_LIB):      # built by tools/build_trace_lib.sh with -DX6_PHASE_TRACE
    _lib.use_library_for_testing(_lib.SsnLibrary(TRACE_LIB))
lib = _lib.get_lib()
dev = torch.device("cuda:0")
n = 288
cases = [("conv2_3x3", 64, 192, 3, 1, 1, 56, 19), ("conv2_3x3", 64, 192, 3, 1, 1, 56, 6), ("3b_d3x3_2", 96, 96, 3, 1, 1, 28, 2),
         ("4d_d3x3_2", 192, 192, 3, 1, 1, 14, 19), ("4d_d3x3_2", 192, 192, 3, 1, 1, 14, 5), ("4c_red", 576, 256, 1, 1, 0, 14, 5),
         ("3a_1x1", 192, 64, 1, 1, 0, 28, 6), ("5b_3x3", 192, 320, 3, 1, 1, 7, 2)]
TILES = {0: (128, 128), 1: (64, 128), 2: (96, 128), 3: (64, 64), 4: (32, 128), 5: (128, 128), 6: (64, 128), 7: (128, 64),
         8: (128, 256), 9: (64, 256), 10: (96, 256), 11: (64, 128), 12: (160, 256), 13: (128, 128), 14: (64, 256),
         15: (128, 256), 16: (128, 256), 17: (96, 256), 18: (160, 256), 19: (192, 256)}
DESYNC = [0]
for (name, cin, cout, k, s, p, h, cfg), des in [(c, d) for c in cases for d in DESYNC]:
    lib.cdll.ssn_conv_x6_debug_flags(des)
    x = K.guarded_empty((n, cin, h, h), dev).normal_(); w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    K.attach_amax(x, K.tensor_amax(x))
    y = torch.empty(n, cout, h, h, device=dev); sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    (wp,) = K.pack_weights_multi([([w], 0)], x6=True)
    bm, bn = TILES[cfg]
    nblk = ((n * h * h + bn - 1) // bn) * ((cout + bm - 1) // bm)
    tr = torch.zeros(nblk * 32, dtype=torch.int64, device=dev)
    fn = lambda: K.conv_x6_fwd(K.full(x), wp, sc, sh, K.full(y), k, s, p, True, cfg)
    fn(); torch.cuda.synchronize()
    lib.cdll.ssn_conv_x6_debug_trace(ctypes.c_void_p(tr.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.cdll.ssn_conv_x6_debug_trace(ctypes.c_void_p(0))
    t = tr.cpu().numpy().reshape(nblk, 32)
    t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    hw, xcc = t[:, 4], t[:, 5] & 0xF
    cu = (xcc << 16) | (hw & 0xFFF0 & ~0x30)          # drop wave/simd bits, keep cu/sh/se + xcc
    nslab = ((cin + 15) // 16) * k * k
    span = (t3.max() - t0.min())
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); fn(); e0.record(); torch.cuda.synchronize()
    print("dbg %d: %.3f ms" % (des, s0.elapsed_time(e0)), end="  ")
    print("%s cfg%d: %d blocks, %d slabs; kernel span %d cyc; per block: prologue %.0f, loop %.0f (%.0f/slab), epilogue %.0f, total %.0f"
          % (name, cfg, nblk, nslab, span, (t1 - t0).mean(), (t2 - t1).mean(), (t2 - t1).mean() / nslab, (t3 - t2).mean(),
             (t3 - t0).mean()))
    names = ["vmwait", "barA", "issue", "1st", "barB", "2nd"]
    for g in range(2 if cfg >= 8 else 1):
        ph = t[:, 8 + 8 * g: 14 + 8 * g].mean(axis=0) / nslab
        print("   group %d per slab: %s  (1st/2nd = %s)" % (g, "  ".join("%s %.0f" % (n_, v) for n_, v in zip(names, ph)),
              "front/mfma" if g == 0 else "mfma/front"))
    ncu = len(np.unique(cu))
    first = np.argsort(t0)[:0]
    print("   distinct CUs %d; CUs hosting blocks 0-255: %d, blocks 256-511: %d" %
          (ncu, len(np.unique(cu[:256])), len(np.unique(cu[256:512]))))
