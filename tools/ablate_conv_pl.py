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
        # tick rate of the cycle counter, blocks in flight (sum of block lifetimes / kernel span) and per-CU co-residency
        span = float(t[:, 3].max() - t[:, 0].min())
        conc = float((t[:, 3] - t[:, 0]).sum()) / span
        cu = (t[:, 5] & 0xF) * 4096 + ((t[:, 4] >> 8) & 0xFFF)       # XCC id | (CU, SH, SE) bits of HW_ID
        order = np.argsort(t[:, 0])
        peak = {}
        live = {}
        for i in order:
            c = int(cu[i])
            lst = [e for e in live.get(c, []) if e > t[i, 0]]
            lst.append(t[i, 3])
            live[c] = lst
            peak[c] = max(peak.get(c, 0), len(lst))
        pk = np.array(list(peak.values()))
        print("    traced blocks %d of %d; span %.0f ticks = %.4f ms -> %.2f GHz tick rate; blocks in flight %.0f (%.2f per CU of %d CUs "
              "seen); peak co-resident blocks per CU: mean %.2f max %d" % (len(t), nblk, span, t_full, span / (t_full * 1e6), conc,
                                                                           conc / len(peak), len(peak), pk.mean(), pk.max()))
        print("%s tile %d (%dx%d, %d blocks, %d slabs) %.1f TF | %s" % (name, tile, bm.value, bn.value, nblk, nslab,
                                                                       flops / t_full / 1e9, " | ".join(res)))
        print("    per block: prologue %.0f  loop %.0f (%.0f / slab)  epilogue %.0f  total %.0f cycles; kernel span %.0f" % (
            (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 2] - t[:, 1]).mean() / nslab, (t[:, 3] - t[:, 2]).mean(),
            (t[:, 3] - t[:, 0]).mean(), float(t[:, 3].max() - t[:, 0].min())), flush=True)
