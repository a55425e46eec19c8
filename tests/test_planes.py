"""Planes tensors (csrc/planes.h) and the kernels that read / write them, against float64 torch references of the same ops.

Every test runs through the host emulator build of the kernel sources (CPU tier, small shapes) and, with ``-m gpu``, through
libssn_hip.so at the backbone's layer shapes.  Tolerances are relative to the largest magnitude of the reference and are
those of the fp32-layout split kernels (two f16 terms per operand = 22 significant bits, fp32 accumulation).
"""
import pytest
import numpy as np
import torch
import torch.nn.functional as F

import action_detection_amd  # noqa: F401
from action_detection_amd import kernels as K
from action_detection_amd import planes as P


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_planes_roundtrip_and_scales(backend):
    g = torch.Generator().manual_seed(0)
    for (n, c, h, w) in [(2, 8, 5, 7), (1, 20, 4, 4), (3, 3, 6, 6)]:
        x = torch.randn(n, c, h, w, generator=g) * 37.0
        t = P.from_f32(backend.put(x))
        assert t.g == (c + 7) // 8
        s = t.scale.cpu().item()
        amax = x.abs().max().item()
        assert 2 ** 14 <= amax * s < 2 ** 15, "exact scale puts the maximum just below 2^15"
        back = P.to_f32(t).cpu()
        assert rel_err(back, x) < 2.0 ** -21
        # padding channels are zero
        raw = t.data.cpu().float().view(2, n, t.g, h * w, 8)
        if c % 8:
            assert (raw[:, :, -1, :, c % 8:] == 0).all()
    # space-to-depth view of the stem input
    x = torch.randn(2, 3, 8, 6, generator=g)
    t = P.from_f32(backend.put(x), s2d=True)
    back = P.to_f32(t).cpu()
    ref = torch.zeros(2, 12, 4, 3)
    for c in range(3):
        for a in range(2):
            for b in range(2):
                ref[:, (c * 2 + a) * 2 + b] = x[:, c, a::2, b::2]
    assert rel_err(back, ref) < 2.0 ** -21


def test_scales_update_protocol(backend):
    pool = P.SlotPool(4, backend.device)
    pool.used = 4
    pool.amax.copy_(torch.tensor([3.0, 0.0, 1e-6, 5000.0]))
    pool.scale.copy_(torch.tensor([1.0, 7.0, 1.0, 16.0]))
    pool.update()
    s = pool.scale.cpu()
    assert 2 ** 12 <= 3.0 * s[0] < 2 ** 13 and s[1] == 7.0 and 2 ** 12 <= 1e-6 * s[2] < 2 ** 13
    assert s[3] != 16.0 and 2 ** 12 <= 5000.0 * s[3] < 2 ** 13      # 5000 * 16 > 65504: re-derived, and flagged
    assert pool.flag.cpu()[0].item() == 1 and pool.flag.cpu()[1].item() == 3
    assert (pool.amax.cpu() == 0).all()
    # hysteresis: a maximum that still sits inside [2^10, 2^14) of the current scale keeps it
    pool.flag.zero_()
    pool.amax.copy_(torch.tensor([5.0, 0.0, 0.4e-6, 5000.0]))
    pool.update()
    assert torch.equal(pool.scale.cpu(), s) and pool.flag.cpu()[1].item() == 0
    # odd slots (the executor's gradient tensors) with 3 more bits of head-room: target [2^9, 2^10), kept inside [2^7, 2^11); the
    # range check flags what left the f16 range (bit 0) or fell 8 bits below the slot's target (bit 1), and modifies nothing
    pool.odd_extra_bits = 3
    pool.flag.zero_()
    pool.amax.copy_(torch.tensor([3.0, 3.0, 3.0, 3.0]))
    pool.scale.copy_(torch.tensor([1.0, 1.0, 2.0 ** 10, 2.0 ** 10]))
    pool.update()
    s = pool.scale.cpu()
    assert 2 ** 12 <= 3.0 * s[0] < 2 ** 13 and 2 ** 9 <= 3.0 * s[1] < 2 ** 10
    assert s[2] == 2.0 ** 10 and s[3] != 2.0 ** 10 and 2 ** 9 <= 3.0 * s[3] < 2 ** 10     # 3 * 2^10 = 3072: inside the even band only
    pool.flag.zero_()
    pool.amax.copy_(torch.tensor([3.0, 3.0, 0.0, 3.0]))
    pool.scale.copy_(torch.tensor([4.0, 0.5, 1.0, 2.0 ** 15]))      # 12 (< 16: drained), 1.5 (< 2: fine for an odd slot ... no), -, 98304 (clamped)
    before = (pool.amax.clone(), pool.scale.clone())
    pool.range_check()
    assert pool.flag.cpu()[0].item() == 3 and torch.equal(pool.amax, before[0]) and torch.equal(pool.scale, before[1])
    pool.flag.zero_()
    pool.scale.copy_(torch.tensor([8.0, 1.0, 1.0, 2.0 ** 10]))      # 24, 3 (>= 2), -, 3072: all in range
    pool.range_check()
    assert pool.flag.cpu()[0].item() == 0


CASES_SMALL = [
    # N, Cin, H, W, Cout, kh, kw, stride, ph, pw
    (2, 16, 9, 9, 40, 3, 3, 1, 1, 1), (1, 16, 7, 7, 96, 1, 1, 1, 0, 0), (2, 8, 10, 10, 32, 3, 3, 2, 1, 1),
    (1, 24, 6, 8, 64, 1, 1, 1, 0, 0), (2, 16, 5, 5, 72, 5, 5, 1, 2, 2), (1, 32, 8, 8, 48, 1, 7, 1, 0, 3),
    (1, 16, 9, 9, 32, 3, 3, 1, 0, 0), (1, 16, 12, 12, 64, 4, 4, 1, 2, 2), (2, 16, 9, 9, 32, 3, 3, 2, 0, 0),
    (5, 24, 7, 7, 48, 3, 3, 1, 1, 1), (3, 40, 6, 11, 72, 3, 3, 1, 1, 1),      # 3x3 / pad 1: tiles that cross images, an 8-channel tail group
    (3, 16, 5, 5, 32, 3, 3, 2, 0, 0), (5, 8, 3, 3, 16, 1, 1, 1, 0, 0),        # images of 4 / 9 output pixels: a 16-slot k-step spans several images
]
CASES_GPU = [
    (9, 64, 56, 56, 192, 3, 3, 1, 1, 1), (18, 192, 28, 28, 224, 1, 1, 1, 0, 0), (18, 128, 28, 28, 160, 3, 3, 2, 1, 1),
    (18, 576, 14, 14, 512, 1, 1, 1, 0, 0), (18, 160, 14, 14, 192, 3, 3, 1, 1, 1), (36, 1056, 7, 7, 832, 1, 1, 1, 0, 0),
    (36, 224, 7, 7, 224, 3, 3, 1, 1, 1), (4, 16, 112, 112, 64, 4, 4, 1, 2, 2), (4, 48, 35, 35, 64, 5, 5, 1, 2, 2),
    (4, 128, 17, 17, 128, 1, 7, 1, 0, 3), (4, 128, 17, 17, 128, 7, 1, 1, 3, 0), (4, 32, 37, 37, 64, 3, 3, 1, 0, 0),
    (18, 96, 28, 28, 96, 3, 3, 1, 1, 1), (5, 64, 35, 35, 96, 3, 3, 1, 1, 1),
]


HALO_TILES = [32 + c for c in (0, 1, 2, 3, 4, 7, 8, 9, 11)]
# the haloed kernel with per-image tiles (tile_cfg 48 + c): conv2's 56 x 56 layer (its image-crossing tiles do not fit the halo buffer);
# round 5 measured it on the MI355X (profiles/r5_off_code_measured.txt: dgrad -8 %, picked by the tile table) -- default tier now
HALO_PI_TILES = [48 + c for c in (0, 1, 4, 7, 8, 9)]


def _pack_fwd(w, backend):
    cout, cin, kh, kw = w.shape
    if kh == kw and kh in (1, 3):
        return K.pack_weights_multi([([backend.put(w)], 0)], x6=True)[0]
    return K.pack_weights_rect(backend.put(w))


def test_conv_pl_forward(backend):
    """conv + folded BN + ReLU on planes vs float64; every tile config on the first case."""
    g = torch.Generator().manual_seed(1)
    cases = CASES_GPU if backend.is_gpu else CASES_SMALL
    ntiles = int(action_detection_amd._lib.get_lib().cdll.ssn_conv_pl_tiles())
    for ci, (n, cin, h, wd, cout, kh, kw, s, ph, pw) in enumerate(cases):
        x = torch.randn(n, cin, h, wd, generator=g) * 3.0
        w = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        ref = F.relu(F.conv2d(x.double(), w.double(), None, s, (ph, pw)) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1))
        ho, wo = ref.shape[2], ref.shape[3]
        xp = P.from_f32(backend.put(x))
        wp = _pack_fwd(w, backend)
        c0, ctot = 16, cout + 48
        tiles = list(range(ntiles)) if ci == 0 else [-1]
        if (kh, kw, s, ph, pw) == (3, 3, 1, 1, 1):
            tiles += HALO_TILES if (ci == 0 or backend.is_gpu) else [32, 36]      # the haloed 3x3 kernel (tile 32 + c)
            tiles += HALO_PI_TILES if (ci == 0 or backend.is_gpu) else [55]
        for tile in tiles:
            if tile >= 32:      # (on the GPU cases some tiles span more slots than the halo buffer, e.g. 128 pixels across two 56 x 56 images: plain kernel)
                taken = action_detection_amd._lib.get_lib().cdll.ssn_conv_pl_halo_taken(n, h, wd, tile)
                assert taken == 1 or backend.is_gpu, (n, h, wd, tile)
            y = P.PlaneTensor(n, ctot, ho, wo, backend.device)
            y.data.fill_(7.0)
            # the delayed protocol: first pass with a guessed scale records the maximum, the update derives the scale,
            # the second pass stores with it
            for _ in range(2):
                P.conv_fwd(P.pfull(xp), wp, backend.put(scale), backend.put(shift), P.PSlice(y, c0, cout), kh, kw, s, ph, pw,
                           True, tile)
                assert abs(y.amax.cpu().item() - ref.abs().max().item()) <= 1e-4 * ref.abs().max().item()
                y.pool.update()
            got = P.to_f32(P.PSlice(y, c0, cout)).cpu()
            assert rel_err(got, ref) < 3e-6, (n, cin, h, cout, kh, kw, s, tile)
            raw = y.data.cpu().float()
            assert (raw[:, :, :c0 // 8] == 7.0).all() and (raw[:, :, (c0 + cout) // 8:] == 7.0).all(), "wrote outside its slice"


def test_conv_pl_raw_rows_and_row_gap(backend):
    """Fused block-input launch: rows >= raw_from take no affine / ReLU, rows >= row_split land row_gap channels further up."""
    g = torch.Generator().manual_seed(2)
    n, cin, h, cout = (4, 192, 28, 224) if backend.is_gpu else (1, 16, 6, 96)
    split, gap, raw_from = (64, 192, 192) if backend.is_gpu else (32, 64, 64)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    ctot = split + gap + (cout - split)
    scale = torch.rand(ctot, generator=g) + 0.5       # indexed by destination channel
    shift = torch.randn(cout, generator=g) * 0.1      # indexed by launch row
    z = F.conv2d(x.double(), w.double())
    ref = torch.zeros(n, ctot, h, h, dtype=torch.float64)
    for m in range(cout):
        d = m if m < split else m + gap
        if m < raw_from:
            ref[:, d] = F.relu(z[:, m] * scale[d].double() + shift[m].double())
        else:
            ref[:, d] = z[:, m]
    xp = P.from_f32(backend.put(x))
    wp = _pack_fwd(w, backend)
    y = P.PlaneTensor(n, ctot, h, h, backend.device).zero_()
    for _ in range(2):
        P.conv_fwd(P.pfull(xp), wp, backend.put(scale), backend.put(shift), P.PSlice(y, 0, cout), 1, 1, 1, 0, 0, True, -1,
                   raw_from=raw_from, row_split=split, row_gap=gap)
        y.pool.update()
    got = P.to_f32(y).cpu()
    assert rel_err(got, ref) < 3e-6


def test_conv_pl_dgrad(backend):
    """Data gradient (stride 1) on planes vs autograd in float64: plain, accumulating, and with the fused ReLU / BN mask."""
    g = torch.Generator().manual_seed(3)
    cases = [c for c in (CASES_GPU if backend.is_gpu else CASES_SMALL) if c[7] == 1]
    halo_done = False
    for (n, cin, h, wd, cout, kh, kw, s, ph, pw) in cases:
        x = torch.randn(n, cin, h, wd, generator=g, dtype=torch.float64).requires_grad_()
        w = torch.randn(cout, cin, kh, kw, generator=g) * 0.1
        y = F.conv2d(x, w.double(), None, 1, (ph, pw))
        gy = torch.randn(y.shape, generator=g) * 1e-3
        y.backward(gy.double())
        dref = x.grad
        rev = not (kh == kw and kh in (1, 3))
        if not rev:
            wt = K.pack_weights_multi([([backend.put(w)], 1)], x6=True)[0]
        else:
            wt = K.pack_dgrad_rect(backend.put(w))
        gp = P.from_f32(backend.put(gy))
        dx = P.PlaneTensor(n, cin, h, wd, backend.device)
        halo = (HALO_TILES if (backend.is_gpu or not halo_done) else [32, 36]) if (kh, kw, ph, pw) == (3, 3, 1, 1) else []
        if halo:
            halo = halo + (HALO_PI_TILES if backend.is_gpu else ([48, 52, 55] if not halo_done else [52]))
        halo_done = halo_done or bool(halo)      # (emulator: every haloed tile on the first 3x3 case, two of them on the others)
        for tile in [-1] + halo:
            dx.data.fill_(3.0)
            for _ in range(2):
                P.conv_dgrad(P.pfull(gp), wt, P.pfull(dx), kh, kw, ph, pw, tile_cfg=tile, taps_reversed=rev)
                dx.pool.update()
            assert rel_err(P.to_f32(dx), dref) < 3e-6, ("dgrad", n, cin, h, cout, kh, kw, tile)
        # accumulate on top of itself, then the mask: dx <- (dx + dgrad) * (act > 0) * mscale
        act = torch.randn(n, cin, h, wd, generator=g).clamp(min=0)
        msc = torch.randn(cin, generator=g)
        msc[::5] = float("nan")          # channels that are not ReLU outputs pass through
        actp = P.from_f32(backend.put(act))
        for _ in range(2):
            P.conv_dgrad(P.pfull(gp), wt, P.pfull(dx), kh, kw, ph, pw, taps_reversed=rev)      # dx = d
            P.conv_dgrad(P.pfull(gp), wt, P.pfull(dx), kh, kw, ph, pw, accumulate=True, mask=P.pfull(actp),
                         mask_scale=backend.put(msc), taps_reversed=rev)                                          # dx = mask(2 d)
            dx.pool.update()
        m = torch.where(torch.isnan(msc).view(1, -1, 1, 1), torch.ones_like(act),
                        (act > 0).float() * torch.nan_to_num(msc).view(1, -1, 1, 1)).double()
        assert rel_err(P.to_f32(dx), 2 * dref * m) < 4e-6, ("dgrad mask", n, cin, h, cout, kh, kw)


def test_conv_pl_wgrad(backend):
    """Weight + bias gradient on planes (LDS transpose reads) vs autograd in float64: every stride / tap shape, every tile."""
    g = torch.Generator().manual_seed(4)
    cases = CASES_GPU if backend.is_gpu else CASES_SMALL
    ntiles = int(action_detection_amd._lib.get_lib().cdll.ssn_conv_wgrad_pl_tiles())
    for ci, (n, cin, h, wd, cout, kh, kw, s, ph, pw) in enumerate(cases):
        x = torch.randn(n, cin, h, wd, generator=g)
        w = (torch.randn(cout, cin, kh, kw, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
        b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x.double(), w, b, s, (ph, pw))
        gy = torch.randn(y.shape, generator=g) * 1e-3
        y.backward(gy.double())
        ho, wo = y.shape[2], y.shape[3]
        # the output gradient lives in a slice of a wider tensor, the input in a slice too
        gt = P.PlaneTensor(n, cout + 16, ho, wo, backend.device).zero_()
        P.from_f32(backend.put(gy), P.PSlice(gt, 8, cout))
        xt = P.PlaneTensor(n, cin + 8, h, wd, backend.device).zero_()
        P.from_f32(backend.put(x), P.PSlice(xt, 8, cin))
        tiles = list(range(ntiles)) if ci in (0, 2) else [-1]
        if (kh, kw, s, ph, pw) == (3, 3, 1, 1, 1):
            tiles += [100, 101, 102, 103]     # the nine-tap kernel's tiles
        if (kh, kw, s, ph, pw) == (1, 1, 1, 0, 0):
            tiles += [200, 201, 202, 203]     # the chunked 1x1 kernel's tiles
        for tile in tiles:
            print("  wgrad case", (n, cin, h, wd, cout, kh, kw, s, ph, pw), "tile", tile, flush=True)
            ws = backend.put(torch.empty(P.wgrad_workspace_bytes(n, cin, cout, ho, wo, kh, kw, tile) // 4))
            dw, db = backend.put(torch.full(w.shape, 9.0)), backend.put(torch.full((cout,), 9.0))
            P.conv_wgrad(P.PSlice(gt, 8, cout), P.PSlice(xt, 8, cin), dw, db, kh, kw, s, ph, pw, ws, tile)
            assert rel_err(dw, w.grad) < 5e-6, ("wgrad", n, cin, h, cout, kh, kw, s, tile)
            assert rel_err(db, b.grad) < 5e-6, ("bias", n, cin, h, cout, kh, kw, s, tile)


def test_conv_pl_wgrad_group(backend):
    """ssn_conv_wgrad_pl_group: ALL problems of the case list as ONE grouped call (one launch per kernel family + one reduction) vs autograd in
    float64 -- every family (nine taps on rows of <= 14 / <= 30 / <= 56 pixels, one-tap bodies with any taps / stride / padding,
    chunked 1x1), operands in slices of wider tensors, a problem without a bias gradient, a fused block-input problem (row gap);
    then every variant forced by hint on the problems that take it; a second call must reproduce the first bit by bit."""
    g = torch.Generator().manual_seed(41)
    cases = list(CASES_GPU if backend.is_gpu else CASES_SMALL)
    # rows of 28 / 56 pixels for the XP = 8 / 12 nine-tap families (emulator: narrow slices of such rows keep it affordable)
    cases += [] if backend.is_gpu else [(1, 8, 5, 28, 16, 3, 3, 1, 1, 1), (1, 8, 3, 56, 16, 3, 3, 1, 1, 1), (2, 72, 4, 4, 136, 1, 1, 1, 0, 0),
                                        (2, 16, 9, 6, 40, 7, 1, 1, 3, 0)]      # (7 x 1: the seven-tap column family)
    # the space-to-depth stem (4x4 taps, two padding pixels in front and ONE behind: outputs = inputs): 12 (RGB) / 40 (flow) real
    # channels, an output-channel count that is not a multiple of 64, a row length that is not a multiple of 4
    # (emulator: the third case is 1452 padded slots in ONE share -- the X ring of 1024 slots wraps)
    stem = [(4, 12, 112, 112, 64), (2, 40, 112, 112, 64)] if backend.is_gpu else [(2, 12, 6, 8, 64), (1, 40, 5, 7, 72), (3, 12, 20, 20, 64)]
    n_plain = len(cases)
    cases += [(n, cin, h, wd, cout, 4, 4, 1, 2, 2) for (n, cin, h, wd, cout) in stem]
    jobs, want, keep = [], [], []
    for ci, (n, cin, h, wd, cout, kh, kw, s, ph, pw) in enumerate(cases):
        x = torch.randn(n, cin, h, wd, generator=g)
        w = (torch.randn(cout, cin, kh, kw, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
        b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x.double(), w, b, s, (ph, pw))
        if ci >= n_plain:
            y = y[:, :, :h, :wd]
        gy = torch.randn(y.shape, generator=g) * 1e-3
        y.backward(gy.double())
        ho, wo = y.shape[2], y.shape[3]
        gap = 16 if (ci == 1 and cout > 32) else 0      # rows >= 32 of this problem's dY sit 16 channels further up (fused block input)
        gt = P.PlaneTensor(n, cout + 16 + gap, ho, wo, backend.device).zero_()
        if gap:
            P.from_f32(backend.put(gy[:, :32].contiguous()), P.PSlice(gt, 8, 32))
            P.from_f32(backend.put(gy[:, 32:].contiguous()), P.PSlice(gt, 8 + 32 + gap, cout - 32), exact=False)
            # (both slices must share one scale: the second is stored with the scale the first derived)
        else:
            P.from_f32(backend.put(gy), P.PSlice(gt, 8, cout))
        xt = P.PlaneTensor(n, cin + 8, h, wd, backend.device).zero_()
        P.from_f32(backend.put(x), P.PSlice(xt, 8, cin))
        dw = backend.put(torch.full(w.shape, 9.0))
        db = None if ci == 2 else backend.put(torch.full((cout,), 9.0))
        jobs.append(P.WgradJob(P.PSlice(gt, 8, cout), P.PSlice(xt, 8, cin), dw, db, kh, kw, s, ph, pw,
                               g_row_split=32 if gap else 0, g_row_gap=gap))
        want.append((w.grad, b.grad))
        keep.append((gt, xt))
    ws_bytes, tb_bytes, plan = P.wgrad_group_plan(jobs)
    fams = sorted({f for f, _, _, _ in plan})
    print("  group plan (family, variant, splits, units):", plan, flush=True)
    assert fams == [0, 1, 2, 3, 4, 5, 6], fams             # every kernel family has a problem (5 / 6: 1 x 7 and 7 x 1 on seven taps)
    assert all(f == 4 for f, _, _, _ in plan[n_plain:]) and len(plan) - n_plain == len(stem)
    assert {v for f, v, _, _ in plan if f == 3} >= {2, 3} or backend.is_gpu
    P.conv_wgrad_group(jobs)
    first = [(j.dw.clone(), None if j.db is None else j.db.clone()) for j in jobs]
    for i, (j, (dwr, dbr)) in enumerate(zip(jobs, want)):
        assert rel_err(j.dw, dwr) < 5e-6, ("group wgrad", i, cases[i], plan[i])
        if j.db is not None:
            assert rel_err(j.db, dbr) < 5e-6, ("group bias", i, cases[i], plan[i])
    # same buffers, caller-provided workspace / table, poisoned destinations: bit-identical
    ws = backend.put(torch.full((ws_bytes // 4 + 4,), float("nan")))
    tb = backend.put(torch.zeros(tb_bytes, dtype=torch.uint8))
    for j in jobs:
        j.dw.fill_(3.0)
        if j.db is not None:
            j.db.fill_(3.0)
    P.conv_wgrad_group(jobs, ws, tb)
    for i, (j, (dw0, db0)) in enumerate(zip(jobs, first)):
        assert torch.equal(j.dw.cpu(), dw0.cpu()), i
        assert j.db is None or torch.equal(j.db.cpu(), db0.cpu()), i
    # every compiled variant, forced by hint, on the problems that take it (a hint the problem cannot take is an error)
    for hint in (0, 3, 8, 200, 100):
        sub, idx = [], []
        for i, (j, c) in enumerate(zip(jobs, cases)):
            n, cin, h, wd, cout, kh, kw, s, ph, pw = c
            one_by_one = (kh, kw, s, ph, pw) == (1, 1, 1, 0, 0)
            ok = {200: one_by_one, 100: (kh, kw, s, ph, pw) == (3, 3, 1, 1, 1)}.get(hint, True)
            if ok and (backend.is_gpu or i < 6 or hint >= 100):
                j.hint = hint
                j.dw.fill_(5.0)
                sub.append(j)
                idx.append(i)
        if not sub:
            continue
        P.conv_wgrad_group(sub)
        for j, i in zip(sub, idx):
            assert rel_err(j.dw, want[i][0]) < 5e-6, ("hint", hint, i, cases[i])
            if j.db is not None:
                assert rel_err(j.db, want[i][1]) < 5e-6, ("hint bias", hint, i, cases[i])
            j.hint = -1
    jobs[2].hint = 200                                        # a 3x3 / stride-2 problem cannot take the chunked 1x1 body
    with pytest.raises(RuntimeError):
        P.conv_wgrad_group([jobs[2]])


def test_planes_im2col(backend):
    """ssn_pl_im2col against F.unfold: 3 / 10 channels, 3x3 / 2 unpadded (Inception-v3's first layer), 3x3 / 1 padded, 5x5 / 2; and the
    weight gradient of such a layer as a 1x1 problem on it (= dW of the k x k convolution, flattened)."""
    g = torch.Generator().manual_seed(77)
    for (n, c, h, w, k, s_, p_) in [(2, 3, 11, 9, 3, 2, 0), (1, 10, 8, 8, 3, 1, 1), (2, 3, 12, 12, 5, 2, 2)]:
        x = torch.randn(n, c, h, w, generator=g)
        xp = P.from_f32(backend.put(x))
        ho, wo = (h + 2 * p_ - k) // s_ + 1, (w + 2 * p_ - k) // s_ + 1
        y = P.im2col(P.PSlice(xp, 0, c), k, k, s_, p_, p_, ho, wo)
        ref = F.unfold(P.to_f32(P.PSlice(xp, 0, c)).cpu(), k, padding=p_, stride=s_).view(n, c * k * k, ho, wo)
        assert torch.equal(P.to_f32(P.PSlice(y, 0, c * k * k)).cpu(), ref), (n, c, h, w, k, s_, p_)
        assert y.g * 8 == (c * k * k + 7) // 8 * 8
        if y.g * 8 > c * k * k:       # the channels that pad K to a multiple of 8 hold zeros
            assert float(P.to_f32(P.PSlice(y, 0, y.g * 8)).cpu()[:, c * k * k:].abs().max()) == 0.0
        cout = 32
        wt = (torch.randn(cout, c, k, k, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
        out = F.conv2d(x.double(), wt, None, s_, p_)
        gy = torch.randn(out.shape, generator=g) * 1e-3
        out.backward(gy.double())
        gp = P.from_f32(backend.put(gy))
        dw, db = backend.put(torch.full((cout, c, k, k), 9.0)), backend.put(torch.full((cout,), 9.0))
        P.conv_wgrad_group([P.WgradJob(P.pfull(gp), P.PSlice(y, 0, c * k * k), dw.view(cout, c * k * k, 1, 1), db, 1, 1, 1, 0, 0)])
        assert rel_err(dw, wt.grad) < 5e-6 and rel_err(db, gy.double().sum((0, 2, 3))) < 5e-6


def test_wgrad_deferred_reduce_multi(backend):
    """The split-K slabs of several layers (one-tap, nine-tap, chunked 1x1) left in their own workspace regions and reduced by ONE
    launch (ssn_wgrad_reduce_multi) give bit-identical dW / db to the per-layer reductions."""
    g = torch.Generator().manual_seed(14)
    layers = [(2, 16, 9, 40, 3, 1, 1, -1), (1, 24, 6, 64, 1, 1, 0, -1), (2, 8, 10, 32, 3, 2, 1, -1), (1, 16, 7, 96, 1, 1, 0, 200),
              (2, 16, 9, 40, 3, 1, 1, 0)]
    if backend.is_gpu:
        layers = [(9, 64, 56, 192, 3, 1, 1, -1), (18, 576, 14, 512, 1, 1, 0, -1), (18, 128, 28, 160, 3, 2, 1, -1),
                  (18, 192, 28, 224, 1, 1, 0, 201), (18, 160, 14, 192, 3, 1, 1, -1)]
    want, entries, keep = [], [], []
    for (n, cin, h, cout, k, s, pad, tile) in layers:
        x = torch.randn(n, cin, h, h, generator=g)
        ho = (h + 2 * pad - k) // s + 1
        gy = torch.randn(n, cout, ho, ho, generator=g) * 1e-3
        gp, xp = P.from_f32(backend.put(gy)), P.from_f32(backend.put(x))
        nbytes = P.wgrad_workspace_bytes(n, cin, cout, ho, ho, k, k, tile)
        dw0, db0 = backend.put(torch.empty(cout, cin, k, k)), backend.put(torch.empty(cout))
        P.conv_wgrad(P.pfull(gp), P.pfull(xp), dw0, db0, k, k, s, pad, pad, backend.put(torch.empty(nbytes // 4)), tile)
        want.append((dw0, db0))
        dw1, db1 = backend.put(torch.full((cout, cin, k, k), 7.0)), backend.put(torch.full((cout,), 7.0))
        ws = backend.put(torch.empty(nbytes // 4))
        P.conv_wgrad(P.pfull(gp), P.pfull(xp), dw1, db1 if len(entries) != 1 else None, k, k, s, pad, pad, ws, tile, defer=entries)
        keep.append((dw1, db1, gp, xp, ws))
    assert len(entries) == len(layers) and all(float(k_[0].flatten()[0]) == 7.0 for k_ in keep)      # nothing reduced yet
    P.wgrad_reduce_multi(entries)
    for i, ((dw0, db0), (dw1, db1, _, _, _)) in enumerate(zip(want, keep)):
        assert torch.equal(dw0.cpu(), dw1.cpu()), i
        assert torch.equal(db0.cpu(), db1.cpu()) if i != 1 else bool((db1 == 7.0).all()), i      # (entry 1 asked for no bias gradient)


def test_wgrad_reduce_tap_major(backend):
    """Split-K slabs in the nine-tap kernel's tap-major column order are summed and permuted back to dW[m][ci][tap]."""
    g = torch.Generator().manual_seed(9)
    for (m, cin, taps, splits) in [(5, 7, 9, 3), (64, 40, 9, 11), (3, 4, 1, 8)]:
        k = cin * taps
        part = torch.randn(splits, m, k + 1, generator=g)
        dw, db = backend.put(torch.full((m, cin, taps), 9.0)), backend.put(torch.full((m,), 9.0))
        K.wgrad_reduce(backend.put(part), dw, db, splits, taps)
        s = part.double().sum(0)
        ref = s[:, :k].view(m, taps, cin).permute(0, 2, 1) if taps > 1 else s[:, :k].view(m, cin, 1)
        assert rel_err(dw, ref) < 1e-6 and rel_err(db, s[:, k]) < 1e-6, (m, cin, taps, splits)
        K.wgrad_reduce(backend.put(part), dw, None, splits, taps)         # (no bias gradient wanted)
        assert rel_err(dw, ref) < 1e-6


def _two_pass(fn, t):
    """run a producer twice around a scale update (the delayed-scale protocol), leave the result of the second pass"""
    fn()
    t.pool.update()
    fn()


def test_conv_pl_dgrad_stride2(backend):
    g = torch.Generator().manual_seed(5)
    cases = [(4, 128, 28, 160, 1), (4, 96, 35, 96, 0)] if backend.is_gpu else [(2, 16, 8, 32, 1), (1, 16, 9, 40, 0), (1, 16, 10, 32, 0)]
    for (n, cin, h, cout, pad) in cases:
        x = torch.randn(n, cin, h, h, generator=g, dtype=torch.float64).requires_grad_()
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        y = F.conv2d(x, w.double(), None, 2, pad)
        gy = torch.randn(y.shape, generator=g) * 1e-2
        y.backward(gy.double())
        wt = K.pack_dgrad_s2(backend.put(w))
        gp = P.from_f32(backend.put(gy))
        act = torch.randn(n, cin, h, h, generator=g).clamp(min=0)
        msc = torch.rand(cin, generator=g) + 0.5
        actp = P.from_f32(backend.put(act))
        dx = P.PlaneTensor(n, cin, h, h, backend.device).zero_()
        _two_pass(lambda: P.conv_dgrad_s2(P.pfull(gp), wt, P.pfull(dx), pad, mask=P.pfull(actp), mask_scale=backend.put(msc)), dx)
        ref = x.grad * (act > 0).double() * msc.double().view(1, -1, 1, 1)
        assert rel_err(P.to_f32(dx), ref) < 4e-6, (n, cin, h, cout, pad)


def test_planes_pools(backend):
    """max pool (ceil mode, argmax routing), average pool behind a projection, global pool, ReLU/BN backward, channel sums."""
    g = torch.Generator().manual_seed(6)
    n, c = (6, 64) if backend.is_gpu else (2, 16)
    # (57 / 30 / 29: rows of several cache lines, with and without padding; also between slices of wider tensors)
    for (h, k, s, pad) in [(12, 3, 2, 0), (7, 3, 1, 1), (9, 3, 2, 0), (10, 3, 2, 1), (11, 3, 2, 1), (57, 3, 2, 0), (30, 3, 2, 1), (29, 3, 2, 0)]:
        x = torch.randn(n, c, h, h, generator=g).clamp(min=0)          # post-ReLU: many exact-zero ties
        xd = x.double().requires_grad_()
        ref, idx = F.max_pool2d(xd, k, s, pad, ceil_mode=True, return_indices=True)
        ho = ref.shape[2]
        xp = P.from_f32(backend.put(x))
        y = P.PlaneTensor(n, c, ho, ho, backend.device)
        am = backend.put(torch.zeros((n, c // 8, ho * ho, 8), dtype=torch.uint8))
        _two_pass(lambda: P.maxpool_fwd(P.pfull(xp), P.pfull(y), am, k, s, pad), y)
        assert rel_err(P.to_f32(y), ref) < 2.0 ** -21
        if h >= 24:      # the same pool between slices of wider tensors: bit-identical planes and argmax
            xw = P.PlaneTensor(n, c + 16, h, h, backend.device).zero_()
            P.from_f32(backend.put(x), P.PSlice(xw, 8, c))
            yw = P.PlaneTensor(n, c + 8, ho, ho, backend.device).zero_()
            am2 = backend.put(torch.zeros((n, c // 8, ho * ho, 8), dtype=torch.uint8))
            _two_pass(lambda: P.maxpool_fwd(P.PSlice(xw, 8, c), P.PSlice(yw, 0, c), am2, k, s, pad), yw)
            assert torch.equal(am2.cpu(), am.cpu()), ("argmax through slices", h)
            assert rel_err(P.to_f32(P.PSlice(yw, 0, c)), ref) < 2.0 ** -21
        gy = torch.randn(ref.shape, generator=g)
        ref.backward(gy.double())
        gp = P.from_f32(backend.put(gy))
        dx = P.PlaneTensor(n, c, h, h, backend.device)
        _two_pass(lambda: P.maxpool_bwd(P.pfull(gp), am, P.pfull(dx), k, s, pad), dx)
        assert rel_err(P.to_f32(dx), xd.grad) < 2.0 ** -20, ("maxpool bwd", h, k, s, pad)
        # accumulate + mask
        msc = torch.rand(c, generator=g) + 0.5
        msc[3] = float("nan")
        base = torch.randn(n, c, h, h, generator=g)
        P.from_f32(backend.put(base), dx)
        dx.pool.scale.mul_(0.25)        # leave head-room for the sum (the executor's scale comes from the last step's maximum)
        P.from_f32(backend.put(base), dx, exact=False)
        P.maxpool_bwd(P.pfull(gp), am, P.pfull(dx), k, s, pad, accumulate=True, mask=P.pfull(xp), mask_scale=backend.put(msc))
        m = torch.where(torch.isnan(msc).view(1, -1, 1, 1), torch.ones_like(x), (x > 0).float() * torch.nan_to_num(msc).view(1, -1, 1, 1))
        assert rel_err(P.to_f32(dx), (base.double() + xd.grad) * m.double()) < 2.0 ** -19
        # the gradient written as fp32 NCHW (the stem's pool: its only reader is a weight gradient on the fp32-layout kernels),
        # with the producer's ReLU / BN mask fused; even widths take the paired 8-byte stores of the 2 x 2 block kernel
        dx32 = K.attach_amax(backend.put(torch.full((n, c, h, h), 7.0)))
        P.maxpool_bwd(P.pfull(gp), am, dx32, k, s, pad, mask=P.pfull(xp), mask_scale=backend.put(msc))
        assert rel_err(dx32, xd.grad * m.double()) < 2.0 ** -20, ("maxpool bwd fp32", h, k, s, pad)
        assert abs(float(dx32._ssn_amax) - float((xd.grad * m.double()).abs().max())) <= 1e-6 * float(xd.grad.abs().max())
        if (k, s) == (3, 2):
            # the same mask read from the POOLED activation (the window's maximum is the element its gradient goes to): planes and
            # fp32 output, identical to the input-mask result
            dxp = P.PlaneTensor(n, c, h, h, backend.device)
            _two_pass(lambda: P.maxpool_bwd(P.pfull(gp), am, P.pfull(dxp), k, s, pad, mask=P.pfull(y), mask_scale=backend.put(msc),
                                            mask_pooled=True), dxp)
            assert rel_err(P.to_f32(dxp), xd.grad * m.double()) < 2.0 ** -20, ("maxpool bwd pooled mask", h, k, s, pad)
            dx32.fill_(7.0)
            P.maxpool_bwd(P.pfull(gp), am, dx32, k, s, pad, mask=P.pfull(y), mask_scale=backend.put(msc), mask_pooled=True)
            assert rel_err(dx32, xd.grad * m.double()) < 2.0 ** -20, ("maxpool bwd pooled mask fp32", h, k, s, pad)
    # average pool behind the projection + its backward stencil
    h = 7
    z = torch.randn(n, c, h, h, generator=g)
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    ref = F.relu(F.avg_pool2d(z.double(), 3, 1, 1, count_include_pad=True) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    zp = P.from_f32(backend.put(z))
    y = P.PlaneTensor(n, c + 8, h, h, backend.device).zero_()
    _two_pass(lambda: P.avgpool_affine(P.pfull(zp), P.PSlice(y, 8, c), backend.put(sc), backend.put(sh), True, 3, 1), y)
    assert rel_err(P.to_f32(P.PSlice(y, 8, c)), ref) < 1e-6
    dz = P.PlaneTensor(n, c, h, h, backend.device)
    _two_pass(lambda: P.avgpool_affine(P.pfull(zp), P.pfull(dz), None, None, False, 3, 1), dz)
    assert rel_err(P.to_f32(dz), F.avg_pool2d(z.double(), 3, 1, 1, count_include_pad=True)) < 1e-6
    # relu / bn backward in place, channel sums, global pool
    act = torch.randn(n, c, h, h, generator=g).clamp(min=0)
    msc = torch.randn(c, generator=g)
    gz = P.from_f32(backend.put(z))
    P.relu_bn_bwd(P.pfull(gz), P.pfull(P.from_f32(backend.put(act))), backend.put(msc))
    refm = z.double() * (act > 0).double() * msc.double().view(1, -1, 1, 1)
    assert rel_err(P.to_f32(gz), refm) < 1e-6
    out = backend.put(torch.empty(c))
    ws = backend.put(torch.empty(P.channel_sum_workspace_bytes(c) // 4))
    P.channel_sum(P.pfull(gz), out, ws)
    assert rel_err(out, refm.sum(dim=(0, 2, 3))) < 2e-6
    # several slices in one pair of launches: bit-identical to the single-slice calls
    wide = P.from_f32(backend.put(torch.randn(n, c + 24, 5, 5, generator=g)))
    ent = [(P.pfull(gz), backend.put(torch.empty(c))), (P.PSlice(wide, 8, 16), backend.put(torch.empty(16))),
           (P.PSlice(wide, 24, c), backend.put(torch.empty(c)))]
    wsm = backend.put(torch.empty(P.channel_sum_workspace_bytes(2 * c + 16) // 4))
    P.channel_sum_multi(ent, wsm)
    for sl, got in ent:
        one = backend.put(torch.empty(sl.c))
        P.channel_sum(sl, one, ws)
        assert torch.equal(got, one)
    feat = backend.put(torch.empty(n, c))
    P.gap_fwd(P.pfull(zp), feat)
    assert rel_err(feat, z.double().mean(dim=(2, 3))) < 1e-6
    # 21 channel groups of a wider tensor (a workgroup takes 16: one full, one of five), 7 x 7 pixels; the sums run in pixel order in
    # fp32: the host repeats them on the stored values and must get the same bits
    cw = 21 * 8
    zw = torch.randn(n, cw + 16, 7, 7, generator=g)
    zwp = P.from_f32(backend.put(zw))
    featw = backend.put(torch.empty(n, cw))
    P.gap_fwd(P.PSlice(zwp, 8, cw), featw)
    assert rel_err(featw, zw[:, 8:8 + cw].double().mean(dim=(2, 3))) < 1e-6
    sc = np.float32(zwp.scale.cpu().item())
    stored = (P.to_f32(P.PSlice(zwp, 8, cw)).cpu().numpy() * sc).reshape(n, cw, 49)      # (power-of-two scale: exact)
    acc = np.zeros((n, cw), dtype=np.float32)
    for q in range(49):
        acc = (acc + stored[:, :, q]).astype(np.float32)
    want = acc * (np.float32(1.0) / (np.float32(49.0) * sc))
    assert np.array_equal(featw.cpu().numpy(), want), "global average pool: sums in pixel order"
    dfeat = torch.randn(n, c, generator=g)
    dxp = P.PlaneTensor(n, c, h, h, backend.device)
    actp = P.from_f32(backend.put(act))
    _two_pass(lambda: P.gap_bwd(backend.put(dfeat), P.pfull(dxp), mask=P.pfull(actp), mask_scale=backend.put(msc)), dxp)
    refg = (dfeat.double() / (h * h)).view(n, c, 1, 1) * (act > 0).double() * msc.double().view(1, -1, 1, 1)
    assert rel_err(P.to_f32(dxp), refg) < 1e-6


def test_planes_bn_train(backend):
    """Training-mode BatchNorm2d + ReLU on planes slices (csrc/planes_bn.hip) against float64 autograd of F.batch_norm: statistics,
    running statistics, output, dgamma / dbeta, dz as planes and as fp32 NCHW; slices inside wider tensors."""
    g = torch.Generator().manual_seed(11)
    shapes = [(6, 64, 28), (3, 24, 9)] if backend.is_gpu else [(3, 16, 6), (2, 24, 5)]
    for (n, c, h) in shapes:
        for relu in (True, False):
            # |mean| several sigma on some channels (the pivoted variance), a conv bias that only the running mean sees.  (A planes
            # tensor carries 22 bits below ITS maximum: the bounds below are relative to that, as for every kernel of this layout)
            z = torch.randn(n, c, h, h, generator=g) * (torch.rand(c, generator=g) * 3 + 0.5).view(1, -1, 1, 1) \
                + (torch.randn(c, generator=g) * 4).view(1, -1, 1, 1)
            bias = torch.randn(c, generator=g)
            gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
            gamma[1] = -gamma[1]
            rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
            eps, mom = 1e-5, 0.1
            zd = z.double().requires_grad_()
            gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
            rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
            pre = F.batch_norm(zd + bias.double().view(1, -1, 1, 1), rm_ref, rv_ref, gd, bd, True, mom, eps)
            ref = F.relu(pre) if relu else pre
            # the slice sits at channel 8 of a wider planes tensor (image stride != slice size)
            zt = P.PlaneTensor(n, c + 16, h, h, backend.device).zero_()
            zs = P.PSlice(zt, 8, c)
            zfull = torch.zeros(n, c + 16, h, h)
            zfull[:, 8:8 + c] = z
            P.from_f32(backend.put(zfull), zt)
            mean, invstd = backend.put(torch.empty(c)), backend.put(torch.empty(c))
            rmd, rvd = backend.put(rm.clone()), backend.put(rv.clone())
            ws = backend.put(torch.empty(P.bn_train_workspace_bytes(c) // 4))
            P.bn_train_stats(zs, backend.put(bias), mean, invstd, rmd, rvd, eps, mom, ws)
            bm = z.double().mean(dim=(0, 2, 3))
            bv = z.double().var(dim=(0, 2, 3), unbiased=False)
            assert (mean.cpu().double() - bm).abs().max() < z.abs().max() * 2.0 ** -21
            assert rel_err(invstd, 1.0 / (bv + eps).sqrt()) < 5e-6
            assert rel_err(rmd, rm_ref) < 1e-6 and rel_err(rvd, rv_ref) < 2e-6
            yt = P.PlaneTensor(n, c + 8, h, h, backend.device).zero_()
            ys = P.PSlice(yt, 0, c)
            _two_pass(lambda: P.bn_train_apply(zs, ys, mean, invstd, backend.put(gamma), backend.put(beta), relu), yt)
            assert rel_err(P.to_f32(ys), ref) < 1e-5, ("bn apply", n, c, h, relu)
            # backward
            gy = torch.randn(ref.shape, generator=g) * 1e-2
            ref.backward(gy.double())
            gp = P.from_f32(backend.put(gy))
            dgm, dbt = backend.put(torch.empty(c)), backend.put(torch.empty(c))
            dzt = P.PlaneTensor(n, c + 8, h, h, backend.device).zero_()
            dzs = P.PSlice(dzt, 8, c)
            _two_pass(lambda: P.bn_train_bwd(P.pfull(gp), ys, zs, mean, invstd, backend.put(gamma), dgm, dbt, dzs, ws, relu), dzt)
            assert rel_err(dgm, gd.grad) < 1e-5 and rel_err(dbt, bd.grad) < 1e-5, ("bn bwd params", n, c, h, relu)
            assert rel_err(P.to_f32(dzs), zd.grad) < 2e-5, ("bn bwd dz", n, c, h, relu)
            dz32 = K.attach_amax(backend.put(torch.full((n, c, h, h), 7.0)))
            P.bn_train_bwd(P.pfull(gp), ys, zs, mean, invstd, backend.put(gamma), dgm, dbt, dz32, ws, relu)
            assert rel_err(dz32, zd.grad) < 2e-5, ("bn bwd dz fp32", n, c, h, relu)
            assert abs(float(dz32._ssn_amax) - float(zd.grad.abs().max())) <= 1e-4 * float(zd.grad.abs().max())


def test_batched_weight_packing_is_bit_identical(backend):
    """kernels.PackBatch: the packing calls of a pass recorded and issued in three launches through a device-resident plan give the
    very bytes the separate calls give -- fused pairs, dgrad operands, stride-2 parity sections, rectangular taps, the stem's
    space-to-depth operand, more entries than one kernel-argument table holds (40) -- and keep doing so when the batch repeats with
    the same buffers (plan not rewritten) and after the weights changed in place."""
    g = torch.Generator().manual_seed(21)

    def rnd(*shape):
        return backend.put(torch.randn(*shape, generator=g) * 0.1)
    sq = [([rnd(24, 16, 3, 3)], 0), ([rnd(16, 8, 1, 1), rnd(8, 8, 1, 1)], 0), ([rnd(32, 16, 1, 1), rnd(16, 16, 1, 1), rnd(8, 16, 1, 1)], 0)]
    sq += [([rnd(8 + 8 * (i % 3), 16, 1, 1)], i % 2) for i in range(45)]
    dg = [([rnd(24, 16, 3, 3)], 1), ([rnd(16, 24, 1, 1), rnd(8, 24, 1, 1)], 1)]
    rect = [rnd(16, 8, 1, 7), rnd(8, 16, 5, 5), rnd(24, 8, 3, 1)]
    s2 = rnd(16, 24, 3, 3)
    stem = rnd(16, 3, 7, 7)

    def run_all():
        outs = list(K.pack_weights_multi(sq, x6=True)) + list(K.pack_weights_multi(dg, x6=True))
        outs += list(K.pack_rect_multi(rect)) + list(K.pack_rect_multi(rect, dgrad=True))
        outs += [K.pack_dgrad_s2(s2), K.pack_dgrad_rect(rect[0]), K.pack_weights_rect(K.s2d_weights(stem))]
        return outs
    ref = [t.clone() for t in run_all()]
    state = {}
    for rep in range(3):
        with K.PackBatch(state, backend.device):
            got = run_all()
        assert len(got) == len(ref)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), ("entry", i, "repeat", rep)
        if rep == 0:
            ptrs = [t.data_ptr() for t in got]
        else:
            assert ptrs == [t.data_ptr() for t in got], "operand buffers must repeat (the plan is keyed on them)"
        for t in got:
            t.fill_(7.0)      # (a repeat must rewrite everything)
    # weights updated in place (an optimizer step): same plan, new contents
    for ws, _ in sq + dg:
        for w in ws:
            w.mul_(-1.5)
    s2.add_(0.01)
    ref2 = [t.clone() for t in run_all()]
    with K.PackBatch(state, backend.device):
        got = run_all()
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(got, ref2))
    # an exception inside the block drops the batch, the next one works
    with pytest.raises(ZeroDivisionError):
        with K.PackBatch(state, backend.device):
            K.pack_dgrad_s2(s2)
            1 / 0
    with K.PackBatch({}, backend.device):
        one = K.pack_dgrad_s2(s2)
    assert torch.equal(one.view(torch.int32), K.pack_dgrad_s2(s2).view(torch.int32))


def test_conv_pl_halo_per_image_tiles_at_56(emu):
    """The layer the per-image tiling of the haloed kernel is for: 3x3 / pad 1 at 56 x 56 (conv2), where a 128-pixel tile that crosses an
    image does not fit the halo buffer (tile_cfg 32 + c falls back to the plain kernel) and a tile inside one image does (48 + c)."""
    lib = action_detection_amd._lib.get_lib()
    assert lib.cdll.ssn_conv_pl_halo_taken(2, 56, 56, 32) == 0 and lib.cdll.ssn_conv_pl_halo_taken(2, 56, 56, 48) == 1      # (N = 2: a crossing exists)
    g = torch.Generator().manual_seed(31)
    n, cin, cout, h = 1, 16, 40, 56
    x = torch.randn(n, cin, h, h, generator=g, dtype=torch.float64).requires_grad_()
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    y = F.conv2d(x, w.double(), None, 1, 1)
    gy = torch.randn(y.shape, generator=g) * 1e-3
    y.backward(gy.double())
    dev = torch.device("cpu")
    xp = P.from_f32(x.detach().float())
    wp = _pack_fwd(w, type("B", (), {"put": staticmethod(lambda t: t)}))
    wt = K.pack_weights_multi([([w], 1)], x6=True)[0]
    gp = P.from_f32(gy)
    one, zero = torch.ones(cout), torch.zeros(cout)
    for tile in (52,):
        yt = P.PlaneTensor(n, cout, h, h, dev)
        _two_pass(lambda: P.conv_fwd(P.pfull(xp), wp, one, zero, P.pfull(yt), 3, 3, 1, 1, 1, False, tile), yt)
        assert rel_err(P.to_f32(yt), y) < 3e-6, ("fwd", tile)
        dx = P.PlaneTensor(n, cin, h, h, dev)
        _two_pass(lambda: P.conv_dgrad(P.pfull(gp), wt, P.pfull(dx), 3, 3, 1, 1, tile_cfg=tile), dx)
        assert rel_err(P.to_f32(dx), x.grad) < 3e-6, ("dgrad", tile)
