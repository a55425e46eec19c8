"""Where the time of conv_pl_kernel goes: ablation switches (tools/build_ablate_lib.sh) + per-block phase timestamps.

    python tools/ablate_conv_pl.py            (on the GPU box; builds the ablation library first)

Per (layer, tile): ms with nothing removed, without the B fetch, without the A fetch, without both, without the fragment
reads, without the stores; and prologue / loop / epilogue cycles per block from the timestamps."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402
from action_detection_amd import _lib, kernels as K, planes as P  # noqa: E402

ABL = os.path.join(ROOT, "tools", ".trace", "libssn_hip_ablate.so")
if not os.path.exists(ABL):      # normally cross-compiled on the build host (the .so travels to the GPU box, objects do not)
    subprocess.check_call([os.path.join(ROOT, "tools", "build_ablate_lib.sh")], stdout=subprocess.DEVNULL)
_lib.use_library_for_testing(_lib.SsnLibrary(os.path.join(ROOT, "tools", ".trace", "libssn_hip_ablate.so")))
lib = _lib.get_lib()
dev = torch.device("cuda:0")
n = 288
CASES = [("4d_d3x3_2", 192, 192, 3, 14, [0, 4, 7]), ("conv2_3x3", 64, 192, 3, 56, [4, 1]), ("4a_block_in", 576, 512, 1, 14, [0, 10]),
         ("3a_d3x3_2", 96, 96, 3, 28, [7, 0]), ("5a_3x3", 192, 320, 3, 7, [8, 0])]
MASKS = [(0, "full"), (1, "no B"), (2, "no A"), (3, "no A,B"), (4, "no reads"), (7, "no A,B,reads"), (8, "no stores")]


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
    w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    wp = K.pack_weights_multi([([w], 0)], x6=True)[0]
    xp = P.from_f32(x)
    yp = P.PlaneTensor(n, cout, h, h, dev)
    flops = 2.0 * n * h * h * cout * cin * k * k
    for tile in tiles:
        bm, bn = ctypes.c_int(), ctypes.c_int()
        lib.cdll.ssn_conv_pl_tile_shape(tile, ctypes.byref(bm), ctypes.byref(bn))
        nblk = ((n * h * h + bn.value - 1) // bn.value) * ((cout + bm.value - 1) // bm.value)
        fn = lambda: P.conv_fwd(P.pfull(xp), wp, sc, sh, P.pfull(yp), k, k, 1, p, p, True, tile)  # noqa: E731
        fn()
        yp.pool.update()
        res = []
        for mask, label in MASKS:
            lib.cdll.ssn_conv_pl_debug_flags(mask)
            res.append("%s %.4f" % (label, timeit(fn)))
        lib.cdll.ssn_conv_pl_debug_flags(0)
        tr = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        lib.cdll.ssn_conv_pl_debug_trace(ctypes.c_void_p(tr.data_ptr()))
        fn()
        torch.cuda.synchronize()
        lib.cdll.ssn_conv_pl_debug_trace(ctypes.c_void_p(0))
        t = tr.cpu().numpy().reshape(nblk, 8)
        t = t[t[:, 0] > 0]
        nslab = ((cin + 15) // 16) * k * k
        t_full = timeit(fn)
        # per CU (XCC id | SE, SH, CU bits of HW_ID): tick of the cycle counter against the 100 MHz real-time counter, how many
        # blocks were co-resident, and how much of the kernel's duration the CU held at least one / two blocks
        # (cycle counters of different XCDs are not synchronised: nothing is compared across CUs except real-time stamps)
        cu = (t[:, 5] & 0xF) * 256 + ((t[:, 4] >> 8) & 0xFF)
        rt0, rt1 = t[:, 6].astype(np.float64), t[:, 7].astype(np.float64)
        life_rt = rt1 - rt0
        big = life_rt > 0
        tick_ghz = float(((t[big, 3] - t[big, 0]) / life_rt[big]).mean() * 0.1)
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
        print("    traced blocks %d of %d on %d CUs (%d..%d per CU); cycle-counter tick %.3f GHz; kernel %.1f us by real-time stamps "
              "(%.1f us by events); block start spread: median %.1f us, last %.1f us; per CU: peak co-resident mean %.2f max %d, "
              ">=1 block %.1f us, >=2 blocks %.1f us" % (len(t), nblk, len(peaks), nper.min(), nper.max(), tick_ghz, kern_us,
                                                        t_full * 1e3, float(np.median(first_us)), float(first_us.max()),
                                                        peaks.mean(), peaks.max(), busy1.mean(), busy2.mean()))
        print("%s tile %d (%dx%d, %d blocks, %d slabs) %.1f TF | %s" % (name, tile, bm.value, bn.value, nblk, nslab,
                                                                       flops / t_full / 1e9, " | ".join(res)))
        print("    per block: prologue %.0f  loop %.0f (%.0f / slab)  epilogue %.0f  total %.0f ticks" % (
            (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).mean() / nslab, (t[:, 3] - t[:, 2]).mean(),
            (t[:, 3] - t[:, 0]).mean()), flush=True)
