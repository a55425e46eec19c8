"""Per-kernel parity: every C-ABI entry point against a plain torch-CPU fp32 reference of the same op.

Each test runs twice: through the host emulator build of the kernel sources (CPU tier, small
shapes) and, with ``-m gpu``, through libssn_hip.so on the MI355X (larger shapes).  Tolerances are
stated per test; integer outputs (argmax routing, STPP segment assignment, OHEM selection) are exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import action_detection_amd  # noqa: F401
from action_detection_amd import kernels as K


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


CONV_CASES_SMALL = [
    # N, Cin, H, Cout, k, s, p, tile
    (2, 8, 9, 40, 3, 1, 1, 3), (1, 3, 20, 64, 7, 2, 3, 1), (2, 16, 7, 96, 1, 1, 0, 2), (3, 8, 10, 33, 3, 2, 1, 4),
    (1, 16, 8, 130, 1, 1, 0, 0), (1, 4, 12, 160, 3, 1, 1, 5), (2, 4, 6, 128, 3, 1, 1, 6), (2, 4, 6, 100, 3, 1, 1, 7),
    (2, 8, 9, 40, 3, 1, 1, -1),
]
CONV_CASES_GPU = [
    (9, 3, 224, 64, 7, 2, 3, -1), (9, 64, 56, 192, 3, 1, 1, -1), (18, 192, 28, 64, 1, 1, 0, -1),
    (18, 128, 28, 160, 3, 2, 1, -1), (18, 576, 14, 224, 1, 1, 0, -1), (18, 160, 14, 192, 3, 1, 1, -1),
    (36, 256, 14, 256, 3, 2, 1, -1), (36, 1056, 7, 352, 1, 1, 0, -1), (36, 224, 7, 224, 3, 1, 1, -1),
    (5, 96, 28, 96, 3, 1, 1, 0), (5, 96, 28, 96, 3, 1, 1, 1), (5, 96, 28, 96, 3, 1, 1, 2), (5, 96, 28, 96, 3, 1, 1, 3),
    (5, 96, 28, 96, 3, 1, 1, 4), (5, 96, 28, 160, 3, 1, 1, 5), (5, 96, 28, 96, 3, 1, 1, 6), (5, 96, 28, 96, 3, 1, 1, 7),
    (3, 10, 224, 64, 7, 2, 3, -1),
]


def conv_cases(backend):
    return CONV_CASES_GPU if backend.is_gpu else CONV_CASES_SMALL


def test_conv_bn_relu_fwd(backend):
    """cuDNN conv+BN(eval)+ReLU replacement; tolerance 2e-5 relative (fp32 summation order only)."""
    g = torch.Generator().manual_seed(0)
    for (n, cin, h, cout, k, s, p, tile) in conv_cases(backend):
        x = torch.randn(n, cin, h, h, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        ref = F.relu(F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        ho = ref.shape[2]
        c0, ctot = 16, cout + 48      # write into a channel slice of a wider (concat) tensor
        y = torch.full((n, ctot, ho, ho), 7.0)
        yd = backend.put(y)
        K.conv_fwd(K.full(backend.put(x)), K.pack_weights(backend.put(w), False), backend.put(scale), backend.put(shift),
                   K.ChanSlice(yd, c0, cout), k, s, p, True, tile)
        got = yd.cpu()
        assert rel_err(got[:, c0:c0 + cout], ref) < 2e-5, (n, cin, h, cout, k, s, tile)
        assert (got[:, :c0] == 7.0).all() and (got[:, c0 + cout:] == 7.0).all(), "wrote outside its slice"


def test_conv_dgrad_and_wgrad(backend):
    """cuDNN dgrad / wgrad(+bias) replacement vs torch autograd; tolerance 5e-5 relative."""
    g = torch.Generator().manual_seed(1)
    for (n, cin, h, cout, k, s, p, tile) in conv_cases(backend):
        x = torch.randn(n, cin, h, h, generator=g).requires_grad_()
        w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).requires_grad_()
        b = torch.zeros(cout, requires_grad=True)
        y = F.conv2d(x, w, b, s, p)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        ho = y.shape[2]
        gd, xd, wd = backend.put(gy), backend.put(x.detach()), backend.put(w.detach())
        if k != 7:
            for layout in sorted({1, K.dgrad_layout(k, s, p, h, h)}):   # plain and (where it applies) parity mode
                wt = K.pack_weights(wd, layout)
                dx = backend.put(torch.full(x.shape, 0.5))
                K.conv_dgrad(K.full(gd), wt, K.full(dx), k, s, p, True, tile, wt_layout=layout)
                assert rel_err(dx.cpu() - 0.5, x.grad) < 5e-5, ("dgrad", n, cin, h, cout, k, s, layout)
                K.conv_dgrad(K.full(gd), wt, K.full(dx), k, s, p, False, tile, wt_layout=layout)
                assert rel_err(dx, x.grad) < 5e-5
        ws = backend.put(torch.empty(K.wgrad_workspace_bytes(n, cin, cout, ho, ho, k) // 4))
        dw, db = backend.put(torch.empty(w.shape)), backend.put(torch.empty(cout))
        K.conv_wgrad(K.full(gd), K.full(xd), dw, db, k, s, p, ws)
        assert rel_err(dw, w.grad) < 5e-5, ("wgrad", n, cin, h, cout, k, s)
        assert rel_err(db, b.grad) < 5e-5


def test_wgrad_tiles(backend):
    g = torch.Generator().manual_seed(2)
    n, cin, h, cout, k, s, p = (6, 24, 14, 80, 3, 1, 1) if backend.is_gpu else (2, 6, 8, 40, 3, 1, 1)
    x = torch.randn(n, cin, h, h, generator=g)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    for cfg in range(7):
        ws = backend.put(torch.empty(K.wgrad_workspace_bytes(n, cin, cout, h, h, k, cfg) // 4))
        dw = backend.put(torch.empty(w.shape))
        K.conv_wgrad(K.full(backend.put(gy)), K.full(backend.put(x)), dw, None, k, s, p, ws, cfg)
        assert rel_err(dw, w.grad) < 5e-5, cfg


def test_bn_fold_and_relu_bn_bwd(backend):
    g = torch.Generator().manual_seed(3)
    c, n, h = 24, 3, 7
    bias, gamma, beta = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    mean, var = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    scale, shift = backend.put(torch.empty(c)), backend.put(torch.empty(c))
    K.bn_fold(*[backend.put(t) for t in (bias, gamma, beta, mean, var)], 1e-5, scale, shift)
    s_ref = gamma / torch.sqrt(var + 1e-5)
    assert rel_err(scale, s_ref) < 1e-6
    assert rel_err(shift, (bias - mean) * s_ref + beta) < 1e-6
    y = torch.relu(torch.randn(n, c + 8, h, h, generator=g))
    dy = torch.randn(n, c + 4, h, h, generator=g)
    dyd = backend.put(dy.clone())
    K.relu_bn_bwd(K.ChanSlice(dyd, 4, c), K.ChanSlice(backend.put(y), 8, c), backend.put(s_ref))
    ref = dy.clone()
    ref[:, 4:] = dy[:, 4:] * (y[:, 8:] > 0) * s_ref.view(1, -1, 1, 1)
    assert rel_err(dyd, ref) < 1e-6


POOLS = [("max", 3, 2, 0), ("max", 3, 1, 1), ("avg", 3, 1, 1)]


def test_pools(backend):
    """ceil_mode max/avg pools: forward exact; backward routes ties exactly like torch (post-ReLU zeros)."""
    g = torch.Generator().manual_seed(4)
    for kind, k, s, p in POOLS:
        for h in ((112, 56, 28, 14, 7) if backend.is_gpu else (12, 16, 14, 7, 5)):
            n, c = 2, 6
            x = torch.relu(torch.randn(n, c, h, h, generator=g)).requires_grad_()   # many exact ties at 0
            if kind == "max":
                ref = F.max_pool2d(x, k, s, p, ceil_mode=True)
            else:
                ref = F.avg_pool2d(x, k, s, p, ceil_mode=True, count_include_pad=True)
            gy = torch.randn(ref.shape, generator=g)
            ref.backward(gy)
            ho = ref.shape[2]
            y = backend.put(torch.zeros(n, c + 3, ho, ho))
            am = backend.put(torch.zeros(n, c, ho, ho, dtype=torch.uint8)) if kind == "max" else None
            K.pool_fwd(kind, K.full(backend.put(x.detach())), K.ChanSlice(y, 3, c), am, k, s, p)
            assert torch.equal(y.cpu()[:, 3:], ref.detach()) or rel_err(y.cpu()[:, 3:], ref) < 1e-6, (kind, h)
            gfull = torch.zeros(n, c + 3, ho, ho)
            gfull[:, 3:] = gy
            dx = backend.put(torch.ones(n, c, h, h))
            K.pool_bwd(kind, K.ChanSlice(backend.put(gfull), 3, c), am, K.full(dx), k, s, p, True)
            assert rel_err(dx.cpu() - 1.0, x.grad) < 1e-6, (kind, h, "bwd")


def test_avgpool_behind_projection(backend):
    """The pool-projection branch with the pool moved behind the 1x1 convolution (ssn_avgpool_affine_fwd,
    ssn_channel_sum): relu(scale * avgpool(conv1x1(x)) + shift) equals torch's relu(bn(conv1x1(avgpool(x)) + bias)) in
    the forward pass AND in every gradient (input, weight, bias), negative BN scales included."""
    g = torch.Generator().manual_seed(41)
    for h in ((28, 14, 7) if backend.is_gpu else (8, 6, 7)):
        n, cin, cout = (3, 24, 16) if backend.is_gpu else (2, 5, 4)
        x = torch.relu(torch.randn(n, cin, h, h, generator=g)).requires_grad_()
        w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.3).requires_grad_()
        b = (torch.randn(cout, generator=g) * 0.1).requires_grad_()
        scale = torch.rand(cout, generator=g) + 0.5
        scale[1::3] *= -1.0
        beta = torch.randn(cout, generator=g) * 0.1
        # reference order: pool, conv + bias, affine (a frozen BN folded to scale / beta), ReLU
        ref = torch.relu(F.conv2d(F.avg_pool2d(x, 3, 1, 1, ceil_mode=True, count_include_pad=True), w, b)
                         * scale.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
        gy = torch.randn(ref.shape, generator=g)
        ref.backward(gy)
        shift = (b.detach() * scale + beta)                                 # what bn_fold produces
        # product order: z = conv1x1 without bias (torch here; the conv kernels have their own tests), pool + affine + ReLU
        z = F.conv2d(x.detach(), w.detach())
        y = backend.put(torch.zeros(n, cout + 2, h, h))
        K.avgpool_affine_fwd(K.full(backend.put(z)), K.ChanSlice(y, 2, cout), backend.put(scale), backend.put(shift),
                             True, 3, 1, 1)
        assert rel_err(y.cpu()[:, 2:], ref) < 1e-6, h
        # backward: mask (ReLU + scale), bias gradient = channel sums BEFORE the pool's backward, pool backward -> dz
        gfull = torch.zeros(n, cout + 2, h, h)
        gfull[:, 2:] = gy
        gdev = backend.put(gfull)
        gs = K.ChanSlice(gdev, 2, cout)
        K.relu_bn_bwd(gs, K.ChanSlice(y, 2, cout), backend.put(scale))
        db = backend.put(torch.zeros(cout))
        K.channel_sum(gs, db, backend.put(torch.empty(K.channel_sum_workspace_bytes(n, cout) // 4)))
        assert rel_err(db, b.grad) < 1e-5, (h, "bias")
        dz = backend.put(torch.empty(n, cout, h, h))
        K.pool_bwd("avg", gs, None, K.full(dz), 3, 1, 1, False)
        dz_t = dz.cpu()
        dw = torch.einsum("nohw,nchw->oc", dz_t, x.detach()).view_as(w)
        dx = torch.nn.grad.conv2d_input(x.shape, w.detach(), dz_t)
        assert rel_err(dw, w.grad) < 1e-5 and rel_err(dx, x.grad) < 1e-5, h


def test_global_avgpool_and_dropout(backend):
    g = torch.Generator().manual_seed(5)
    n, c, h = 4, 32, 7
    x = torch.randn(n, c, h, h, generator=g)
    y = backend.put(torch.empty(n, c))
    K.gap_fwd(K.full(backend.put(x)), y)
    assert rel_err(y, x.mean(dim=(2, 3))) < 1e-6
    gy = torch.randn(n, c, generator=g)
    dx = backend.put(torch.empty(n, c, h, h))
    K.gap_bwd(backend.put(gy), K.full(dx))
    assert rel_err(dx, (gy / (h * h)).view(n, c, 1, 1).expand(n, c, h, h)) < 1e-6
    # dropout: mask statistics, scaling, backward consistency, determinism in the seed
    v = torch.randn(64, 1024, generator=g)
    out, mask = backend.put(torch.empty_like(v)), backend.put(torch.empty(v.shape, dtype=torch.uint8))
    K.dropout_fwd(backend.put(v), out, mask, 0.8, 1234)
    m = mask.cpu().bool()
    assert abs(m.float().mean().item() - 0.2) < 0.01
    assert rel_err(out.cpu()[m], v[m] * 5.0) < 1e-6 and (out.cpu()[~m] == 0).all()
    out2, mask2 = backend.put(torch.empty_like(v)), backend.put(torch.empty(v.shape, dtype=torch.uint8))
    K.dropout_fwd(backend.put(v), out2, mask2, 0.8, 1234)
    assert torch.equal(mask.cpu(), mask2.cpu())
    K.dropout_fwd(backend.put(v), out2, mask2, 0.8, 99)
    assert not torch.equal(mask.cpu(), mask2.cpu())
    ctr = backend.put(torch.zeros(1, dtype=torch.int64))      # device-side call counter (graph replays)
    K.dropout_fwd(backend.put(v), out, mask, 0.8, 1234, ctr)
    K.dropout_fwd(backend.put(v), out2, mask2, 0.8, 1234, ctr)
    assert int(ctr.item()) == 2 and not torch.equal(mask.cpu(), mask2.cpu())
    dv = backend.put(torch.empty_like(v))
    K.dropout_bwd(backend.put(v), mask, dv, 0.8)
    assert rel_err(dv, out) < 1e-6


def test_linear(backend):
    g = torch.Generator().manual_seed(6)
    for (r, o, d) in ((32, 21, 1024), (32, 40, 3072), (5, 7, 30)):
        x = torch.randn(r, d, generator=g).requires_grad_()
        w = (torch.randn(o, d, generator=g) * 0.05).requires_grad_()
        b = torch.randn(o, generator=g).requires_grad_()
        ref = F.linear(x, w, b)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        out = backend.put(torch.empty(r, o))
        K.linear_fwd(backend.put(x.detach()), backend.put(w.detach()), backend.put(b.detach()), out)
        assert rel_err(out, ref) < 1e-5
        dx, dw, db = backend.put(torch.ones(r, d)), backend.put(torch.empty(o, d)), backend.put(torch.empty(o))
        K.linear_bwd(backend.put(go), backend.put(x.detach()), backend.put(w.detach()), dx, dw, db, True)
        assert rel_err(dx.cpu() - 1.0, x.grad) < 1e-5 and rel_err(dw, w.grad) < 1e-5 and rel_err(db, b.grad) < 1e-5


def test_row_gather_scatter(backend):
    g = torch.Generator().manual_seed(7)
    src = torch.randn(16, 20, 2, generator=g)
    idx = torch.tensor([0, 8, 3, 15])
    out = backend.put(torch.empty(4, 20, 2))
    K.row_gather(backend.put(src), backend.put(idx), out)
    assert torch.equal(out.cpu(), src[idx])
    back = backend.put(torch.full((16, 20, 2), 3.0))
    K.row_scatter(out, backend.put(idx), back)
    ref = torch.zeros(16, 20, 2)
    ref[idx] = src[idx]
    assert torch.equal(back.cpu(), ref)


def test_ce_loss(backend):
    g = torch.Generator().manual_seed(8)
    for r, c in ((8, 21), (64, 101), (3, 5)):
        x = (torch.randn(r, c, generator=g) * 3).requires_grad_()
        t = torch.randint(0, c, (r,), generator=g)
        ref = F.cross_entropy(x, t)
        ref.backward()
        loss, ws = backend.put(torch.empty(1)), backend.put(torch.empty(2 * r))
        K.ce_loss_fwd(backend.put(x.detach()), backend.put(t), loss, ws)
        assert rel_err(loss, ref.reshape(1)) < 1e-5
        dl = backend.put(torch.empty(r, c))
        K.ce_loss_bwd(backend.put(x.detach()), backend.put(t), ws, backend.put(torch.ones(1)), dl)
        assert rel_err(dl, x.grad) < 1e-5


def test_sgd_and_norm(backend):
    g = torch.Generator().manual_seed(9)
    w0, gr = torch.randn(5000, generator=g), torch.randn(5000, generator=g)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=5e-4)
    w, buf = backend.put(w0.clone()), backend.put(torch.zeros(5000))
    for it in range(3):
        p.grad = gr.clone() * (it + 1)
        opt.step()
        K.sgd_step(w, backend.put(gr * (it + 1)), buf, 0.01, 0.9, 5e-4, 1.0, it == 0)
    assert rel_err(w, p.detach()) < 1e-6
    out, ws = backend.put(torch.zeros(1)), backend.put(torch.empty(1024))
    K.sumsq(backend.put(gr), out, False, ws)
    K.sumsq(backend.put(w0), out, True, ws)
    assert rel_err(out, ((gr ** 2).sum() + (w0 ** 2).sum()).reshape(1)) < 1e-5
    x = backend.put(gr.clone())
    K.scale_(x, backend.put(torch.tensor([0.25])))
    K.scale_(x, None, 2.0)
    assert rel_err(x, gr * 0.5) < 1e-7


def test_sgd_first_step_skipped_by_the_fault_word(backend):
    """SSNSGD when the FIRST step of a parameter is the one the device skips (skip_flag set: a flagged pass, another rank's fault
    through the MAX-reduced word, a parameter whose first gradient arrives late): the momentum buffer must not hold garbage that the
    next step multiplies by the momentum (it was torch.empty_like + a host-side first-step marker)."""
    from action_detection_amd.optim import SSNSGD
    g = torch.Generator().manual_seed(21)
    w0 = [torch.randn(700, generator=g), torch.randn(33, 5, generator=g)]
    grads = [[torch.randn(w.shape, generator=g) for w in w0] for _ in range(3)]
    ref = [torch.nn.Parameter(w.clone()) for w in w0]
    ropt = torch.optim.SGD(ref, lr=0.01, momentum=0.9, weight_decay=5e-4)
    got = [torch.nn.Parameter(backend.put(w.clone())) for w in w0]
    opt = SSNSGD([{"params": got, "lr_mult": 1, "decay_mult": 1, "name": "w"}], lr=0.01, momentum=0.9, weight_decay=5e-4)
    flag = backend.put(torch.ones(2, dtype=torch.int32))
    # poison what a fresh allocation of the buffers' size would hand out (best effort: same-size blocks are recycled by the caching
    # allocator on the GPU); with the fix the buffers are zero-filled whatever the allocation held
    junk = [backend.put(torch.full(w.shape, float("nan"))) for w in w0]
    del junk
    for p_, gr in zip(got, grads[0]):
        p_.grad = backend.put(gr.clone())
    opt.step(skip_flag=flag)                                  # flagged: nothing may move
    for p_, w in zip(got, w0):
        assert torch.equal(p_.detach().cpu(), w)
        assert torch.equal(opt.state[p_]["momentum_buffer"].cpu(), torch.zeros(w.shape))
    flag.zero_()
    for it in (1, 2):                                        # the two steps that are applied == torch's first two steps
        for r, p_, gr in zip(ref, got, grads[it]):
            r.grad = gr.clone()
            p_.grad = backend.put(gr.clone())
        ropt.step()
        opt.step(skip_flag=flag)
    for r, p_ in zip(ref, got):
        assert torch.isfinite(p_.detach().cpu()).all()
        assert rel_err(p_.detach(), r.detach()) < 1e-6
        assert rel_err(opt.state[p_]["momentum_buffer"], ropt.state[r]["momentum_buffer"]) < 1e-6


def test_clip_grad_norm(backend):
    """optim.clip_grad_norm (ssn_train.py:245-248) against torch.nn.utils.clip_grad_norm_, clipping and not clipping."""
    from action_detection_amd.optim import clip_grad_norm
    g = torch.Generator().manual_seed(12)
    shapes = [(64, 3, 7, 7), (64,), (21, 1024), (5, 1000)]
    for max_norm in (0.5, 1e6):
        ref = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        got = [torch.nn.Parameter(backend.put(torch.zeros(s))) for s in shapes]
        for r, p in zip(ref, got):
            r.grad = torch.randn(r.shape, generator=g)
            p.grad = backend.put(r.grad.clone())
        n_ref = float(torch.nn.utils.clip_grad_norm_(ref, max_norm))
        n_got = clip_grad_norm(got, max_norm)
        assert abs(n_got - n_ref) < 1e-5 * n_ref
        for r, p in zip(ref, got):
            assert rel_err(p.grad, r.grad) < 1e-6


def test_fused_relu_bn_backward_epilogues(backend):
    """dgrad / max-pool backward as LAST writer: dx <- (dx_old + contribution) * (y > 0) * scale with the scale's sign
    (negative folded BN scales are real), unchanged where the scale is NaN (channel is not a ReLU output)."""
    g = torch.Generator().manual_seed(10)
    n, cin, h, cout, k, s, p = (4, 32, 14, 48, 3, 1, 1) if backend.is_gpu else (2, 6, 7, 40, 3, 1, 1)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    gy = torch.randn(n, cout, h, h, generator=g)
    y_in = torch.relu(torch.randn(n, cin, h, h, generator=g))
    scale = torch.rand(cin, generator=g) + 0.5
    scale[1::3] *= -1.0                                       # negative folded scales (gamma < 0)
    scale[::3] = float("nan")                                 # pass-through channels
    passthru = torch.isnan(scale).view(1, -1, 1, 1)
    old = torch.randn(n, cin, h, h, generator=g)
    contrib = torch.nn.grad.conv2d_input((n, cin, h, h), w, gy, s, p)
    tot = old + contrib
    ref = torch.where(passthru, tot, torch.where(y_in > 0, tot * scale.view(1, -1, 1, 1), torch.zeros_like(tot)))
    dx = backend.put(old.clone())
    K.conv_dgrad(K.full(backend.put(gy)), K.pack_weights(backend.put(w), True), K.full(dx), k, s, p, True,
                 mask_y=K.full(backend.put(y_in)), mask_scale=backend.put(scale))
    assert rel_err(dx, ref) < 5e-5
    # max-pool backward with the same fusion
    x = torch.relu(torch.randn(n, cin, 2 * h, 2 * h, generator=g)).requires_grad_()
    yp = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
    gp = torch.randn(yp.shape, generator=g)
    yp.backward(gp)
    am = backend.put(torch.zeros(yp.shape, dtype=torch.uint8))
    K.pool_fwd("max", K.full(backend.put(x.detach())), K.full(backend.put(torch.empty(yp.shape))), am, 3, 2, 0)
    ref = torch.where(passthru, x.grad,
                      torch.where(x.detach() > 0, x.grad * scale.view(1, -1, 1, 1), torch.zeros_like(x.grad)))
    dxp = backend.put(torch.empty(x.shape))
    K.pool_bwd("max", K.full(backend.put(gp)), am, K.full(dxp), 3, 2, 0, False,
               mask_y=K.full(backend.put(x.detach())), mask_scale=backend.put(scale))
    assert rel_err(dxp, ref) < 1e-6


def test_multi_tensor_sgd_and_bn_fold(backend):
    """The table-driven multi-tensor launches equal their per-tensor counterparts (incl. > 48 tensors)."""
    g = torch.Generator().manual_seed(11)
    n_t = 53
    ws = [torch.randn(int(torch.randint(1, 9000, (1,), generator=g)), generator=g) for _ in range(n_t)]
    gs = [torch.randn(w.shape, generator=g) for w in ws]
    lrs = [0.01 * (1 + i % 2) for i in range(n_t)]
    wds = [5e-4 * (i % 2) for i in range(n_t)]
    wa, ba = [backend.put(w.clone()) for w in ws], [backend.put(torch.zeros_like(w)) for w in ws]
    wb, bb = [backend.put(w.clone()) for w in ws], [backend.put(torch.zeros_like(w)) for w in ws]
    gd = [backend.put(x) for x in gs]
    for it in range(2):
        K.sgd_step_multi(wa, gd, ba, lrs, wds, 0.9, 1.0, it == 0)
        for w, x, b, lr, wd in zip(wb, gd, bb, lrs, wds):
            K.sgd_step(w, x, b, lr, 0.9, wd, 1.0, it == 0)
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(wa, wb))
    cs = [8, 33, 64]
    prm = [[torch.rand(c, generator=g) + 0.5 for c in cs] for _ in range(5)]   # bias gamma beta mean var
    sa, ha = [backend.put(torch.empty(c)) for c in cs], [backend.put(torch.empty(c)) for c in cs]
    sb, hb = [backend.put(torch.empty(c)) for c in cs], [backend.put(torch.empty(c)) for c in cs]
    dev = [[backend.put(t) for t in row] for row in prm]
    K.bn_fold_multi(dev[0], dev[1], dev[2], dev[3], dev[4], [1e-5] * 3, sa, ha)
    for i in range(3):
        K.bn_fold(dev[0][i], dev[1][i], dev[2][i], dev[3][i], dev[4][i], 1e-5, sb[i], hb[i])
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(sa + ha, sb + hb))


def test_pack_weights_multi(backend):
    """Table-driven packing == per-layer packing, incl. a fused pair (two sources) and the parity layout."""
    g = torch.Generator().manual_seed(12)
    mk = lambda *s: backend.put(torch.randn(*s, generator=g))   # noqa: E731
    wa, wb, w3, w7 = mk(24, 16, 1, 1), mk(40, 16, 1, 1), mk(10, 6, 3, 3), mk(8, 3, 7, 7)
    entries = [([wa, wb], 0), ([wa, wb], 1), ([w3], 0), ([w3], 1), ([w3], 2), ([w7], 0), ([wa], 1)]
    got = K.pack_weights_multi(entries)
    cat = torch.cat([wa, wb], 0).contiguous()
    want = [K.pack_weights(cat, 0), K.pack_weights(cat, 1), K.pack_weights(w3, 0), K.pack_weights(w3, 1),
            K.pack_weights(w3, 2), K.pack_weights(w7, 0), K.pack_weights(wa, 1)]
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b.cpu())


X6_CASES_SMALL = [
    # N, Cin, H, Cout, k, s, p, tile     (Cin deliberately not always a multiple of 16: tail-group masking)
    (2, 16, 6, 96, 1, 1, 0, 2), (2, 24, 10, 40, 3, 1, 1, 3), (1, 32, 8, 130, 1, 1, 0, 0), (2, 16, 10, 33, 3, 2, 1, 4),
    (1, 16, 9, 160, 3, 1, 1, 5), (1, 40, 6, 64, 1, 1, 0, 6), (1, 16, 6, 100, 3, 1, 1, 7), (2, 20, 6, 70, 3, 1, 1, 1),
    (1, 16, 14, 40, 3, 1, 1, 0),
    (2, 16, 6, 48, 1, 1, 0, -1),
    # 8-wave ping-pong workgroups (tile ids 8-15)
    (2, 32, 10, 130, 3, 1, 1, 8), (3, 16, 7, 40, 1, 1, 0, 9), (2, 24, 8, 100, 3, 1, 1, 10), (2, 16, 10, 33, 3, 2, 1, 11),
    (1, 16, 12, 170, 3, 1, 1, 12), (2, 40, 6, 64, 1, 1, 0, 13), (3, 16, 6, 70, 3, 1, 1, 14), (2, 16, 8, 128, 1, 1, 0, 15),
    # one-workgroup-per-CU tiles with 128x64 / 96x64 register tiles per wave
    (2, 16, 12, 130, 3, 1, 1, 16), (2, 24, 8, 70, 1, 1, 0, 17),
]
X6_CASES_GPU = [
    (9, 64, 56, 192, 3, 1, 1, -1), (18, 192, 28, 64, 1, 1, 0, -1), (18, 128, 28, 160, 3, 2, 1, -1),
    (18, 576, 14, 224, 1, 1, 0, -1), (18, 160, 14, 192, 3, 1, 1, -1), (36, 1056, 7, 352, 1, 1, 0, -1),
    (36, 224, 7, 224, 3, 1, 1, -1), (4, 72, 14, 96, 3, 1, 1, -1),
    (5, 96, 28, 96, 3, 1, 1, 0), (5, 96, 28, 96, 3, 1, 1, 1), (5, 96, 28, 96, 3, 1, 1, 2), (5, 96, 28, 96, 3, 1, 1, 3),
    (5, 96, 28, 96, 3, 1, 1, 4), (5, 96, 28, 160, 3, 1, 1, 5), (5, 96, 28, 96, 1, 1, 0, 6), (5, 96, 28, 96, 1, 1, 0, 7),
    (5, 96, 28, 96, 3, 1, 1, 8), (5, 96, 28, 96, 3, 1, 1, 9), (5, 96, 28, 96, 3, 1, 1, 10), (5, 96, 28, 96, 3, 2, 1, 11),
    (5, 96, 28, 160, 3, 1, 1, 12), (5, 96, 28, 96, 1, 1, 0, 13), (5, 96, 28, 96, 1, 1, 0, 14), (5, 96, 28, 130, 3, 1, 1, 15),
    (5, 96, 28, 130, 3, 1, 1, 16), (5, 96, 28, 96, 1, 1, 0, 17), (3, 64, 28, 100, 3, 2, 1, 16), (4, 96, 14, 96, 3, 1, 1, 17), (5, 128, 14, 160, 3, 1, 1, 18), (5, 96, 28, 200, 1, 1, 0, 19),
]


def test_conv_x6_fwd_and_dgrad(backend):
    """Split-operand ("x6": two f16 terms per operand, three products) convolution kernels against fp64 torch: 22 of 24
    significand bits per operand and one dropped product below 2^-22 |ab|, so the SAME tolerances as the f32-MFMA kernels apply
    (2e-5 / 5e-5 relative)."""
    g = torch.Generator().manual_seed(20)
    for (n, cin, h, cout, k, s, p, tile) in (X6_CASES_GPU if backend.is_gpu else X6_CASES_SMALL):
        x = torch.randn(n, cin, h, h, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        ref = F.relu(F.conv2d(x.double(), w.double(), None, s, p) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1))
        ho = ref.shape[2]
        wd = backend.put(w)
        wp, wt = K.pack_weights_multi([([wd], 0), ([wd], 1)], x6=True)
        c0, ctot = 16, cout + 48
        gy = torch.randn(n, cout, ho, ho, generator=g)
        gref = torch.nn.grad.conv2d_input((n, cin, h, h), w.double(), gy.double(), s, p) if s == 1 else None
        # guarded = readable floats in front of the gathered tensor -> 16-byte activation loads where they apply
        for guarded in (False, True):
            def put(t):
                if not guarded:
                    return backend.put(t)
                d = K.guarded_empty(t.shape, backend.put(torch.zeros(1)).device)
                d.copy_(t)
                return d
            yd = backend.put(torch.full((n, ctot, ho, ho), 7.0))
            K.conv_x6_fwd(K.full(put(x)), wp, backend.put(scale), backend.put(shift), K.ChanSlice(yd, c0, cout),
                          k, s, p, True, tile)
            got = yd.cpu()
            assert rel_err(got[:, c0:c0 + cout], ref) < 2e-5, ("fwd", n, cin, h, cout, k, s, tile, guarded)
            assert (got[:, :c0] == 7.0).all() and (got[:, c0 + cout:] == 7.0).all(), "wrote outside its slice"
            if s != 1:
                continue
            dx = backend.put(torch.full((n, cin, h, h), 0.5))
            K.conv_x6_dgrad(K.full(put(gy)), wt, K.full(dx), k, p, True, tile)
            assert rel_err(dx.cpu() - 0.5, gref) < 5e-5, ("dgrad", n, cin, h, cout, k, tile, guarded)
            K.conv_x6_dgrad(K.full(put(gy)), wt, K.full(dx), k, p, False, tile)
            assert rel_err(dx, gref) < 5e-5


def test_conv_x6_is_fp32_accurate(backend):
    """The x6 kernel must be in the accuracy class of the exact-f32 MFMA kernel (measured against fp64), orders of
    magnitude away from a plain f16 / bf16 product."""
    g = torch.Generator().manual_seed(21)
    n, cin, h, cout = (4, 256, 14, 128) if backend.is_gpu else (1, 64, 6, 32)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    ref64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    xd, wd = backend.put(x), backend.put(w)
    y32 = backend.put(torch.empty(n, cout, h, h))
    K.conv_fwd(K.full(xd), K.pack_weights(wd, False), None, None, K.full(y32), 3, 1, 1, False)
    err32 = (y32.cpu().double() - ref64).abs().max().item()
    (wp,) = K.pack_weights_multi([([wd], 0)], x6=True)
    y = backend.put(torch.empty(n, cout, h, h))
    K.conv_x6_fwd(K.full(xd), wp, None, None, K.full(y), 3, 1, 1, False)
    err6 = (y.cpu().double() - ref64).abs().max().item()
    bf = lambda t: t.bfloat16().double()   # noqa: E731
    err_bf16 = (F.conv2d(bf(x), bf(w), None, 1, 1) - ref64).abs().max().item()
    assert err6 < 2 * err32 + 1e-6, (err6, err32)
    assert err6 < err_bf16 / 100, (err6, err_bf16)


def test_conv_split_error_growth_with_k(backend):
    """K-sweep of the f16 2-way split (K = Cin * 9 from 27 to 2304, the longest reduction in BN-Inception): maximum
    error and BIAS (mean signed error) relative to sum|x w| against float64, next to the exact-f32 MFMA kernel and a
    plain fp32 (torch CPU) convolution on the same data -- on zero-mean data and on all-positive data (worst case for a
    systematic error: nothing cancels).

    Operands are scaled per tensor and split with round-to-nearest (x s = hi + lo + e, |e| <= 2^-22 |x s|,
    ssn_common.h); the dropped lo*lo product is <= 2^-22 |x w| and, like e, has no preferred sign.  What remains is
    fp32 accumulation error, common to all three kernels: the bounds below tie the split kernel to the fp32 kernels at
    every K instead of to absolute numbers.  (History: round 1 split into three bf16 terms by truncation -- a relative
    bias of -1.0e-7 on positive data in this very test; the round-to-nearest 3-term version measured 2.2e-7 / 1.9e-6
    max error at K = 2304 signed / positive on the MI355X, profiles/r2_ksweep_bf16x6.txt.)"""
    g = torch.Generator().manual_seed(77)
    cins = (3, 16, 64, 128, 256) if backend.is_gpu else (3, 16, 48)
    n, h, cout = (2, 14, 64) if backend.is_gpu else (1, 5, 32)
    rows = []
    for cin in cins:
        for signed in (True, False):
            x = torch.randn(n, cin, h, h, generator=g)
            w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
            if not signed:
                x, w = x.abs(), w.abs()
            ref = F.conv2d(x.double(), w.double(), None, 1, 1)
            mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)          # sum |x w| per output
            xd, wd = backend.put(x), backend.put(w)
            y6 = backend.put(torch.empty(n, cout, h, h))
            (wp,) = K.pack_weights_multi([([wd], 0)], x6=True)
            K.conv_x6_fwd(K.full(xd), wp, None, None, K.full(y6), 3, 1, 1, False)
            y32 = backend.put(torch.empty(n, cout, h, h))
            K.conv_fwd(K.full(xd), K.pack_weights(wd, False), None, None, K.full(y32), 3, 1, 1, False)
            ycpu = F.conv2d(x, w, None, 1, 1)
            e6 = (y6.cpu().double() - ref) / mag
            e32 = (y32.cpu().double() - ref) / mag
            ecpu = (ycpu.double() - ref) / mag
            rows.append((cin * 9, signed, e6.abs().max().item(), e6.mean().item(), e32.abs().max().item(),
                         e32.mean().item(), ecpu.abs().max().item()))
    for k, signed, m6, b6, m32, b32, mcpu in rows:
        print("K=%5d %s  x6: max %.2e bias %+.2e | f32 MFMA: max %.2e bias %+.2e | torch fp32: max %.2e"
              % (k, "signed  " if signed else "positive", m6, b6, m32, b32, mcpu))
        assert m6 <= 3.0 * max(m32, mcpu) + 2.0 ** -22, (k, signed, m6, m32, mcpu)     # the accuracy class of fp32
        assert abs(b6) <= 3.0 * abs(b32) + 2.0 ** -24, (k, signed, b6, b32)             # no systematic error of its own


def test_conv_split_wide_dynamic_range(backend):
    """The f16 split scales each operand TENSOR by one power of two; elements far below the tensor's maximum keep an
    absolute (not relative) accuracy.  (a) five decades of element magnitudes inside both operands: the error relative
    to sum|x w| stays in the fp32 class.  (b) one huge element pins the scale while every other element sits 2^-20
    below it, where the low f16 term is SUBNORMAL: outputs that do not touch the huge element must still be good to
    ~2^-20 relative (they would be ~2^-12 if the matrix cores flushed subnormal f16 inputs)."""
    g = torch.Generator().manual_seed(91)
    n, cin, h, cout = (2, 128, 14, 64) if backend.is_gpu else (1, 32, 6, 32)
    x = torch.randn(n, cin, h, h, generator=g) * torch.exp(torch.rand(n, cin, h, h, generator=g) * -12.0)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05 * torch.exp(torch.rand(cout, cin, 3, 3, generator=g) * -12.0)

    def run(x, w):
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
        (wp,) = K.pack_weights_multi([([backend.put(w)], 0)], x6=True)
        y = backend.put(torch.empty(n, cout, h, h))
        K.conv_x6_fwd(K.full(backend.put(x)), wp, None, None, K.full(y), 3, 1, 1, False)
        return (y.cpu().double() - ref).abs() / mag

    e = run(x, w)
    assert e.max().item() < 2.0 ** -20, e.max().item()          # (a) measured ~5e-7
    x2 = torch.randn(n, cin, h, h, generator=g) * 2.0 ** -20
    x2[0, 0, 0, 0] = 1.0
    w2 = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    e2 = run(x2, w2)
    e2[0, :, :2, :2] = 0.0                                        # the outputs the huge element reaches
    assert e2.max().item() < 2.0 ** -17, e2.max().item()         # (b) subnormal low terms survive


def test_amax_slots(backend):
    """Every kernel that writes a tracked tensor raises the tensor's amax slot to the largest magnitude it stores."""
    g = torch.Generator().manual_seed(92)
    n, cin, h, cout = 2, 16, 8, 40
    x = torch.randn(n, cin, h, h, generator=g) * 3.0
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    xd = backend.put(x)
    assert K.tensor_amax(xd).item() == x.abs().max().item()
    # forward conv (both kernels), into a channel slice of a wider tracked tensor
    for x6 in (False, True):
        y = K.attach_amax(backend.put(torch.zeros(n, cout + 3, h, h)))
        wp = K.pack_weights_multi([([backend.put(w)], 0)], x6=True)[0] if x6 else K.pack_weights(backend.put(w), False)
        (K.conv_x6_fwd if x6 else K.conv_fwd)(K.full(xd), wp, None, None, K.ChanSlice(y, 3, cout), 3, 1, 1, True)
        assert y._ssn_amax.item() == y.abs().max().item() > 0
    # pools, the in-place ReLU/BN backward, global-pool backward
    y = K.attach_amax(backend.put(torch.zeros(n, cin, 4, 4)))
    K.pool_fwd("max", K.full(xd), K.full(y), None, 3, 2, 0)
    assert y._ssn_amax.item() == y.abs().max().item() > 0
    dy = K.attach_amax(backend.put(torch.randn(n, cin, h, h, generator=g)))
    K.relu_bn_bwd(K.full(dy), K.full(xd), backend.put(torch.full((cin,), 2.5)))
    assert dy._ssn_amax.item() == dy.abs().max().item() > 0
    dx = K.attach_amax(backend.put(torch.zeros(n, cin, h, h)))
    K.gap_bwd(backend.put(torch.randn(n, cin, generator=g)), K.full(dx))
    assert dx._ssn_amax.item() == dx.abs().max().item() > 0
    dxp = K.attach_amax(backend.put(torch.zeros(n, cin, h, h)))
    K.pool_bwd("avg", K.full(dy), None, K.full(dxp), 3, 1, 1, False)
    assert dxp._ssn_amax.item() == dxp.abs().max().item() > 0


def test_conv_x6_fused_pair_and_mask(backend):
    """x6 pack with two sources (fused reduce pair) and the dgrad last-writer ReLU/BN epilogue."""
    g = torch.Generator().manual_seed(22)
    n, cin, h = (4, 64, 14) if backend.is_gpu else (1, 16, 6)
    wa = torch.randn(24, cin, 1, 1, generator=g) * 0.1
    wb = torch.randn(40, cin, 1, 1, generator=g) * 0.1
    wcat = torch.cat([wa, wb])
    x = torch.randn(n, cin, h, h, generator=g)
    wp, wt = K.pack_weights_multi([([backend.put(wa), backend.put(wb)], 0), ([backend.put(wa), backend.put(wb)], 1)],
                                  x6=True)
    y = backend.put(torch.empty(n, 64, h, h))
    K.conv_x6_fwd(K.full(backend.put(x)), wp, None, None, K.full(y), 1, 1, 0, False)
    assert rel_err(y, F.conv2d(x, wcat)) < 2e-5
    gy = torch.randn(n, 64, h, h, generator=g)
    y_in = torch.relu(torch.randn(n, cin, h, h, generator=g))
    scale = torch.rand(cin, generator=g) + 0.5
    scale[1::3] *= -1.0                         # negative folded scales keep their sign
    scale[::3] = float("nan")                   # not a ReLU output: pass through
    passthru = torch.isnan(scale).view(1, -1, 1, 1)
    old = torch.randn(n, cin, h, h, generator=g)
    tot = old + torch.nn.grad.conv2d_input((n, cin, h, h), wcat, gy, 1, 0)
    ref = torch.where(passthru, tot, torch.where(y_in > 0, tot * scale.view(1, -1, 1, 1), torch.zeros_like(tot)))
    dx = backend.put(old.clone())
    K.conv_x6_dgrad(K.full(backend.put(gy)), wt, K.full(dx), 1, 0, True, mask_y=K.full(backend.put(y_in)),
                    mask_scale=backend.put(scale))
    assert rel_err(dx, ref) < 5e-5


def test_conv_wgrad_x6(backend):
    """Split-operand (2 x f16) weight gradient (+ fp32 bias gradient) vs torch autograd in fp64; tolerance as the f32 kernel (5e-5)."""
    g = torch.Generator().manual_seed(30)
    cases = ([(6, 24, 14, 80, 3, -1), (4, 64, 28, 96, 1, -1), (3, 40, 56, 64, 3, 2), (5, 576, 14, 224, 1, -1),
              (2, 20, 14, 33, 3, 0), (2, 16, 28, 100, 3, 1), (2, 64, 14, 130, 1, 3), (2, 16, 14, 96, 3, 4),
              (2, 32, 14, 64, 1, 5), (2, 16, 14, 128, 3, 6), (3, 32, 14, 96, 3, 7), (2, 24, 28, 200, 3, 8),
              (1, 64, 14, 100, 1, 7), (3, 16, 14, 192, 1, 8), (5, 192, 7, 320, 3, -1), (4, 1056, 7, 128, 1, 2),
              (3, 40, 7, 96, 3, 7), (2, 24, 5, 64, 1, 0), (3, 64, 14, 130, 3, 9), (3, 48, 14, 160, 3, 10), (2, 96, 14, 200, 1, 11),
              (4, 32, 7, 192, 3, 11)] if backend.is_gpu else
             [(2, 6, 8, 40, 3, 0), (1, 16, 6, 33, 1, 1), (1, 8, 10, 130, 3, 2), (2, 4, 6, 70, 3, 3), (1, 20, 8, 96, 1, 4),
              (1, 4, 14, 64, 3, 5), (2, 8, 6, 128, 1, 6), (3, 5, 4, 20, 3, -1), (1, 30, 6, 100, 3, 7), (1, 8, 6, 200, 1, 8), (3, 12, 7, 40, 3, 4), (2, 20, 5, 33, 1, 0), (1, 30, 4, 130, 3, 9), (1, 20, 4, 200, 1, 11)])
    for (n, cin, h, cout, k, cfg) in cases:
        p = (k - 1) // 2
        x = torch.randn(n, cin, h, h, generator=g)
        w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).double().requires_grad_()
        b = torch.zeros(cout, dtype=torch.double, requires_grad=True)
        y = F.conv2d(x.double(), w, b, 1, p)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.double())
        assert K.wgrad_x6_supported(k, 1, p, h, h)
        dev = backend.put(torch.zeros(1)).device
        xd = K.guarded_empty(x.shape, dev)
        xd.copy_(x)
        ws = backend.put(torch.empty(K.wgrad_x6_workspace_bytes(n, cin, cout, h, h, k, cfg) // 4))
        dw, db = backend.put(torch.empty(cout, cin, k, k)), backend.put(torch.empty(cout))
        K.conv_wgrad_x6(K.full(backend.put(gy)), K.full(xd), dw, db, k, p, ws, cfg)
        assert rel_err(dw, w.grad) < 5e-5, ("wgrad x6", n, cin, h, cout, k, cfg)
        assert rel_err(db, b.grad) < 5e-5
        # channel slices of wider tensors (the guard is then the channels below the slice)
        xw = backend.put(torch.randn(n, cin + 8, h, h, generator=g))
        xw[:, 8:].copy_(backend.put(x))
        K.conv_wgrad_x6(K.full(backend.put(gy)), K.ChanSlice(xw, 8, cin), dw, None, k, p, ws, cfg)
        assert rel_err(dw, w.grad) < 5e-5, ("wgrad x6 slice", n, cin, h, cout, k, cfg)


def test_conv_x6_rect_fwd(backend):
    """Rectangular taps of the Inception-v3 layers (5x5, 1x7, 7x1, 1x3, 3x1) on the x6 kernel vs fp64 torch, with and
    without the 16-byte-load path (planes that are / are not a multiple of 4 pixels), channel tails, every tile."""
    g = torch.Generator().manual_seed(41)
    cases = ([(3, 48, 16, 64, 5, 5, 2, 2, -1), (2, 128, 17, 128, 1, 7, 0, 3, 2), (2, 128, 17, 192, 7, 1, 3, 0, 5),
              (2, 96, 8, 100, 1, 3, 0, 1, 6), (2, 40, 8, 96, 3, 1, 1, 0, 1), (2, 32, 12, 64, 1, 7, 0, 3, -1),
              (1, 64, 35, 96, 5, 5, 2, 2, 2)] if backend.is_gpu else
             [(1, 8, 6, 40, 5, 5, 2, 2, -1), (1, 20, 5, 33, 1, 7, 0, 3, 2), (1, 16, 6, 64, 7, 1, 3, 0, 6),
              (1, 8, 4, 32, 1, 3, 0, 1, 1), (1, 8, 4, 70, 3, 1, 1, 0, 5)])
    for (n, cin, h, cout, kh, kw, ph, pw, tile) in cases:
        x = torch.randn(n, cin, h, h, generator=g)
        w = torch.randn(cout, cin, kh, kw, generator=g) * 0.1
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        ref = F.relu(F.conv2d(x.double(), w.double(), None, 1, (ph, pw)) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1))
        dev = backend.put(torch.zeros(1)).device
        xd = K.guarded_empty(x.shape, dev)
        xd.copy_(x)
        wp = K.pack_weights_rect(backend.put(w))
        y = backend.put(torch.empty(n, cout, h, h))
        K.conv_x6_fwd_rect(K.full(xd), wp, backend.put(scale), backend.put(shift), K.full(y), kh, kw, ph, pw, True, tile)
        assert rel_err(y, ref) < 2e-6, ("x6 rect", n, cin, h, cout, kh, kw, tile)


def test_conv_x6_dgrad_s2(backend):
    """Stride-2 dgrad as four parity-class stride-1 problems on the x6 kernel vs torch autograd in fp64: plain,
    accumulating, with the fused ReLU/BN backward, into a channel slice; 16-byte-load and narrow paths."""
    g = torch.Generator().manual_seed(43)
    cases = ([(3, 128, 28, 160, -1), (2, 96, 28, 96, 2), (2, 128, 14, 192, 3), (2, 256, 14, 256, 5), (1, 40, 12, 70, 6),
              (2, 24, 6, 33, 1)] if backend.is_gpu else [(1, 8, 8, 40, -1), (2, 20, 4, 33, 3), (1, 16, 6, 20, 2)])
    for (n, cin, h, cout, tile) in cases:
        x = torch.randn(n, cin, h, h, generator=g).double().requires_grad_()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1)
        y = F.conv2d(x, w.double(), None, 2, 1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.double())
        dev = backend.put(torch.zeros(1)).device
        gd = K.guarded_empty(gy.shape, dev)
        gd.copy_(gy)
        wt = K.pack_dgrad_s2(backend.put(w))
        dx = backend.put(torch.full((n, cin, h, h), 7.0))
        K.conv_x6_dgrad_s2(K.full(gd), wt, K.full(dx), False, tile)
        assert rel_err(dx, x.grad) < 2e-6, ("dgrad s2", n, cin, h, cout, tile)
        # accumulate + last-writer mask, into a slice of a wider tensor
        prev = torch.randn(n, cin + 5, h, h, generator=g)
        act = torch.randn(n, cin + 5, h, h, generator=g)
        msc = torch.rand(cin, generator=g) + 0.5
        msc[1::3] = -msc[1::3]                                   # negative folded scales
        msc[::3] = float("nan")                                  # channels that are not ReLU outputs
        wide = backend.put(prev.clone())
        K.conv_x6_dgrad_s2(K.full(gd), wt, K.ChanSlice(wide, 5, cin), True, tile,
                           mask_y=K.ChanSlice(backend.put(act), 5, cin), mask_scale=backend.put(msc))
        tot = prev[:, 5:].double() + x.grad
        m = msc.view(1, -1, 1, 1).double()
        ref = torch.where(torch.isnan(m), tot, torch.where(act[:, 5:].double() > 0, tot * m, torch.zeros_like(tot)))
        assert rel_err(wide[:, 5:], ref) < 2e-6, ("dgrad s2 acc+mask", n, cin, h, cout, tile)
        assert torch.equal(wide[:, :5].cpu(), prev[:, :5])


def test_conv_split_rect_backward(backend):
    """Backward of the rectangular-tap layers (Inception-v3 training): data gradient as a forward correlation with the
    transposed, tap-reversed operand (plain / accumulating + fused ReLU-BN mask into a channel slice) and the weight + bias
    gradient with runtime taps, vs torch autograd in fp64; planes that are / are not a multiple of 4 pixels."""
    g = torch.Generator().manual_seed(47)
    cases = ([(3, 48, 16, 64, 5, 5, -1, -1), (2, 128, 17, 160, 1, 7, 2, 2), (2, 160, 17, 192, 7, 1, 5, 3), (2, 96, 8, 100, 1, 3, 6, 0),
              (2, 40, 8, 96, 3, 1, 1, 6), (1, 48, 35, 64, 5, 5, 2, 5), (4, 384, 8, 384, 3, 1, -1, -1)] if backend.is_gpu else
             [(1, 8, 6, 40, 5, 5, -1, -1), (1, 20, 5, 33, 1, 7, 2, 2), (1, 16, 6, 64, 7, 1, 6, 3), (2, 8, 4, 32, 1, 3, 1, 0),
              (1, 8, 3, 70, 3, 1, 5, 6)])
    for (n, cin, h, cout, kh, kw, tile, wcfg) in cases:
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        x = torch.randn(n, cin, h, h, generator=g).double().requires_grad_()
        w = (torch.randn(cout, cin, kh, kw, generator=g) * 0.1).double().requires_grad_()
        b = torch.zeros(cout, dtype=torch.double, requires_grad=True)
        y = F.conv2d(x, w, b, 1, (ph, pw))
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.double())
        dev = backend.put(torch.zeros(1)).device
        guard = K.wgrad_x6_rect_guard_floats(ph, pw, h)
        gd = K.guarded_empty(gy.shape, dev)
        gd.copy_(gy)
        wt = K.pack_dgrad_rect(backend.put(w.detach().float()))
        dx = backend.put(torch.full((n, cin, h, h), 7.0))
        K.conv_x6_dgrad_rect(K.full(gd), wt, K.full(dx), kh, kw, ph, pw, False, tile)
        assert rel_err(dx, x.grad) < 2e-6, ("dgrad rect", n, cin, h, cout, kh, kw, tile)
        prev = torch.randn(n, cin + 5, h, h, generator=g)
        act = torch.randn(n, cin + 5, h, h, generator=g)
        msc = torch.rand(cin, generator=g) + 0.5
        msc[1::3] = -msc[1::3]
        msc[::3] = float("nan")
        wide = backend.put(prev.clone())
        K.conv_x6_dgrad_rect(K.full(gd), wt, K.ChanSlice(wide, 5, cin), kh, kw, ph, pw, True, tile,
                             mask_y=K.ChanSlice(backend.put(act), 5, cin), mask_scale=backend.put(msc))
        tot = prev[:, 5:].double() + x.grad
        m = msc.view(1, -1, 1, 1).double()
        ref = torch.where(torch.isnan(m), tot, torch.where(act[:, 5:].double() > 0, tot * m, torch.zeros_like(tot)))
        assert rel_err(wide[:, 5:], ref) < 2e-6, ("dgrad rect acc+mask", n, cin, h, cout, kh, kw, tile)
        assert torch.equal(wide[:, :5].cpu(), prev[:, :5])
        # weight + bias gradient
        xd = K.guarded_empty(x.shape, dev, guard)
        xd.copy_(x.detach().float())
        ws = backend.put(torch.empty(K.wgrad_x6_rect_workspace_bytes(n, cin, cout, h, h, kh, kw, wcfg) // 4))
        dw, db = backend.put(torch.empty(cout, cin, kh, kw)), backend.put(torch.empty(cout))
        K.conv_wgrad_x6_rect(K.full(gd), K.full(xd), dw, db, kh, kw, ph, pw, ws, wcfg)
        assert rel_err(dw, w.grad) < 5e-5, ("wgrad rect", n, cin, h, cout, kh, kw, wcfg)
        assert rel_err(db, b.grad) < 5e-5


def test_pack_rect_multi(backend):
    """All rectangular-tap operands of a pass in one call == the single-layer packers, bit for bit (forward and dgrad form);
    more entries than one launch table holds."""
    g = torch.Generator().manual_seed(49)
    shapes = [(40, 8, 5, 5), (33, 20, 1, 7), (64, 16, 7, 1), (32, 8, 1, 3), (70, 8, 3, 1)] * (9 if backend.is_gpu else 1)
    ws = [backend.put(torch.randn(sh, generator=g) * (0.01 + 0.1 * (i % 4))) for i, sh in enumerate(shapes)]
    for dgrad in (False, True):
        multi = K.pack_rect_multi(ws, dgrad=dgrad)
        assert len(multi) == len(ws)
        for w, m in zip(ws, multi):
            one = K.pack_dgrad_rect(w) if dgrad else K.pack_weights_rect(w)
            assert torch.equal(one.cpu().view(torch.int32), m.cpu().view(torch.int32))


def test_conv_split_valid_and_stride2_pad0_dgrad(backend):
    """The unpadded 3x3 layers of Inception-v3: stride 1 (dx is larger than dy) on the square dgrad kernel, stride 2 as four
    parity-class launches with the two-tap classes on the EVEN rows / columns; odd and even input sizes; accumulate + mask."""
    g = torch.Generator().manual_seed(48)
    cases = ([(2, 32, 21, 32, 1, -1), (2, 80, 17, 192, 1, 2), (2, 288, 35, 384, 2, -1), (3, 96, 17, 96, 2, 3), (2, 192, 17, 320, 2, 5),
              (2, 40, 8, 48, 2, 1), (1, 24, 12, 70, 2, 6)] if backend.is_gpu else
             [(1, 8, 7, 40, 1, -1), (1, 8, 9, 40, 2, -1), (2, 20, 5, 33, 2, 3), (1, 16, 8, 20, 2, 2), (1, 8, 6, 16, 2, 6)])
    for (n, cin, h, cout, s, tile) in cases:
        x = torch.randn(n, cin, h, h + 2, generator=g).double().requires_grad_()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1)
        y = F.conv2d(x, w.double(), None, s, 0)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.double())
        dev = backend.put(torch.zeros(1)).device
        gd = K.guarded_empty(gy.shape, dev)
        gd.copy_(gy)
        if s == 1:
            (wt,) = K.pack_weights_multi([([backend.put(w)], 1)], x6=True)
            run = lambda dst, acc, **kw: K.conv_x6_dgrad(K.full(gd), wt, dst, 3, 0, acc, tile, **kw)   # noqa: E731
        else:
            wt = K.pack_dgrad_s2(backend.put(w))
            run = lambda dst, acc, **kw: K.conv_x6_dgrad_s2(K.full(gd), wt, dst, acc, tile, pad=0, **kw)   # noqa: E731
        dx = backend.put(torch.full(tuple(x.shape), 7.0))
        run(K.full(dx), False)
        assert rel_err(dx, x.grad) < 2e-6, ("dgrad pad 0", n, cin, h, cout, s, tile)
        prev = torch.randn(n, cin + 5, h, h + 2, generator=g)
        act = torch.randn(n, cin + 5, h, h + 2, generator=g)
        msc = torch.rand(cin, generator=g) + 0.5
        msc[1::3] = -msc[1::3]
        msc[::3] = float("nan")
        wide = backend.put(prev.clone())
        run(K.ChanSlice(wide, 5, cin), True, mask_y=K.ChanSlice(backend.put(act), 5, cin), mask_scale=backend.put(msc))
        tot = prev[:, 5:].double() + x.grad
        m = msc.view(1, -1, 1, 1).double()
        ref = torch.where(torch.isnan(m), tot, torch.where(act[:, 5:].double() > 0, tot * m, torch.zeros_like(tot)))
        assert rel_err(wide[:, 5:], ref) < 2e-6, ("dgrad pad 0 acc+mask", n, cin, h, cout, s, tile)
        assert torch.equal(wide[:, :5].cpu(), prev[:, :5])
        if s == 1:
            # weight gradient of the unpadded layer: dy laid into planes of x's size, then the same-grid kernel with taps 0..2
            wd = w.double().requires_grad_()
            bd = torch.zeros(cout, dtype=torch.double, requires_grad=True)
            F.conv2d(x.detach(), wd, bd, 1, 0).backward(gy.double())
            xd = K.guarded_empty(tuple(x.shape), dev)
            xd.copy_(x.detach().float())
            gp = backend.put(torch.full((n, cout, h, h + 2), 3.0))
            K.embed_planes(K.full(gd), gp)
            assert torch.equal(gp[:, :, :h - 2, :h].cpu(), gy) and float(gp[:, :, h - 2:].abs().max()) == 0 and float(gp[:, :, :, h:].abs().max()) == 0
            ws = backend.put(torch.empty(K.wgrad_x6_rect_workspace_bytes(n, cin, cout, h, h + 2, 3, 3) // 4))
            dw, db = backend.put(torch.empty(cout, cin, 3, 3)), backend.put(torch.empty(cout))
            K.conv_wgrad_x6_rect(K.full(gp), K.full(xd), dw, db, 3, 3, 0, 0, ws)
            assert rel_err(dw, wd.grad) < 5e-5 and rel_err(db, bd.grad) < 5e-5, ("wgrad valid", n, cin, h, cout)


def test_bn_train(backend):
    """Training-mode BatchNorm2d + ReLU (bn_mode 'partial' / 'full'): statistics, running-stat update, output and the full
    backward against torch autograd in float64, on channel slices, with a large per-channel offset (|mean| >> sigma)."""
    g = torch.Generator().manual_seed(93)
    for (n, c, h, off) in ([(4, 20, 14, 50.0), (33, 7, 5, 0.0), (2, 64, 28, 3.0)] if backend.is_gpu else [(3, 5, 6, 50.0), (33, 3, 2, 0.0)]):
        z0 = torch.randn(n, c, h, h, generator=g) * (torch.rand(1, c, 1, 1, generator=g) + 0.5) + off * torch.randn(1, c, 1, 1, generator=g)
        bias = torch.randn(c, generator=g)
        gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
        gy = torch.randn(n, c, h, h, generator=g)
        bn = torch.nn.BatchNorm2d(c, eps=1e-5).double()
        bn.weight.data.copy_(gamma), bn.bias.data.copy_(beta)
        bn.running_mean.normal_(generator=g), bn.running_var.uniform_(0.5, 1.5, generator=g)
        rm0, rv0 = bn.running_mean.clone().float(), bn.running_var.clone().float()
        zin = (z0.double() + bias.double().view(1, c, 1, 1)).requires_grad_()
        yref = torch.relu(bn(zin))
        yref.backward(gy.double())
        # HIP path on slices of wider tensors
        zw = backend.put(torch.zeros(n, c + 3, h, h)); zw[:, 3:] = backend.put(z0)
        yw = backend.put(torch.zeros(n, c + 2, h, h))
        dyw = backend.put(torch.zeros(n, c + 1, h, h)); dyw[:, 1:] = backend.put(gy)
        dzw = backend.put(torch.zeros(n, c + 4, h, h))
        zs, ys, dys, dzs = K.ChanSlice(zw, 3, c), K.ChanSlice(yw, 2, c), K.ChanSlice(dyw, 1, c), K.ChanSlice(dzw, 4, c)
        mean, invstd = backend.put(torch.empty(c)), backend.put(torch.empty(c))
        rm, rv = backend.put(rm0.clone()), backend.put(rv0.clone())
        ws = backend.put(torch.empty(K.bn_train_workspace_floats(n, c)))
        K.bn_train_stats(zs, backend.put(bias), mean, invstd, rm, rv, 1e-5, 0.1, ws)
        K.bn_train_apply(zs, ys, mean, invstd, backend.put(gamma), backend.put(beta), True)
        assert rel_err(yw[:, 2:], yref) < 2e-6, (n, c, h, "y")
        assert rel_err(rm, bn.running_mean) < 1e-6 and rel_err(rv, bn.running_var) < 1e-6
        dgamma, dbeta = backend.put(torch.empty(c)), backend.put(torch.empty(c))
        K.bn_train_bwd(dys, ys, zs, mean, invstd, backend.put(gamma), dgamma, dbeta, dzs, ws, True)
        assert rel_err(dgamma, bn.weight.grad) < 1e-5 and rel_err(dbeta, bn.bias.grad) < 1e-5
        assert rel_err(dzw[:, 4:], zin.grad) < 1e-5, (n, c, h, "dz")
        assert float(dzw[:, :4].abs().max()) == 0.0 and float(yw[:, :2].abs().max()) == 0.0


def test_conv_split_three_sources_and_raw_rows(backend):
    """The fused launch on a block input: reduce pair + pool projection = one convolution with three weight sources whose
    last output channels are raw (no affine, no ReLU); forward, and the dgrad operand packed from the same three sources."""
    g = torch.Generator().manual_seed(94)
    n, cin, h = (3, 192, 28) if backend.is_gpu else (1, 24, 6)
    ca, cb, cp = (64, 64, 32) if backend.is_gpu else (32, 32, 32)
    x = torch.randn(n, cin, h, h, generator=g)
    ws = [torch.randn(c, cin, 1, 1, generator=g) * 0.1 for c in (ca, cb, cp)]
    cout = ca + cb + cp
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    ref = F.conv2d(x.double(), torch.cat(ws).double())
    want = ref.clone()
    want[:, :ca + cb] = torch.relu(ref[:, :ca + cb] * scale[:ca + cb].double().view(1, -1, 1, 1)
                                   + shift[:ca + cb].double().view(1, -1, 1, 1))
    wd = [backend.put(w) for w in ws]
    wp, wt = K.pack_weights_multi([(wd, 0), (wd, 1)], x6=True)
    for tile in (-1, 3, 2):
        y = backend.put(torch.zeros(n, cout, h, h))
        K.conv_x6_fwd(K.full(backend.put(x)), wp, backend.put(scale), backend.put(shift), K.full(y), 1, 1, 0, True, tile,
                      raw_from=ca + cb)
        assert rel_err(y, want) < 2e-5, tile
        assert float(y[:, ca + cb:].min()) < 0          # the raw rows keep their negative values
    gy = torch.randn(n, cout, h, h, generator=g)
    dx = backend.put(torch.zeros(n, cin, h, h))
    K.conv_x6_dgrad(K.full(backend.put(gy)), wt, K.full(dx), 1, 0, False)
    dref = torch.nn.grad.conv2d_input(x.shape, torch.cat(ws).double(), gy.double())
    assert rel_err(dx, dref) < 5e-5


def test_conv_split_block_input_launch(backend):
    """The ONE launch on an Inception block input: 1x1 branch + reduce pair + pool projection as a convolution with four
    weight sources whose first rows live at the head of the block-output tensor and whose other rows behind the block's own
    channels (row gap in forward, channel gap in dgrad, row gap on G in wgrad); the last rows raw."""
    g = torch.Generator().manual_seed(95)
    n, cin, h = (3, 192, 28) if backend.is_gpu else (1, 32, 6)
    c1, ca, cb, cp, cblk = (64, 64, 64, 32, 256) if backend.is_gpu else (32, 32, 32, 32, 160)
    m, cr = c1 + ca + cb + cp, ca + cb + cp
    gapc = cblk - c1
    x = torch.randn(n, cin, h, h, generator=g)
    ws = [torch.randn(c, cin, 1, 1, generator=g) * 0.1 for c in (c1, ca, cb, cp)]
    wcat = torch.cat(ws).double()
    scale, shift = torch.rand(m, generator=g) + 0.5, torch.randn(m, generator=g) * 0.2
    ref = F.conv2d(x.double(), wcat)
    aff = c1 + ca + cb
    want = ref.clone()
    want[:, :aff] = torch.relu(ref[:, :aff] * scale[:aff].double().view(1, -1, 1, 1) + shift[:aff].double().view(1, -1, 1, 1))
    wd = [backend.put(w) for w in ws]
    wp, wt = K.pack_weights_multi([(wd, 0), (wd, 1)], x6=True)
    xd = K.guarded_empty(x.shape, backend.put(torch.zeros(1)).device)
    xd.copy_(backend.put(x))
    for tile in (-1, 3, 2):
        t = backend.put(torch.full((n, cblk + cr, h, h), 7.0))
        tscale = torch.full((cblk + cr,), float("nan"))       # per-CHANNEL scales of the wide tensor; shifts go by row
        tscale[:c1], tscale[cblk:] = scale[:c1], scale[c1:]
        K.conv_x6_fwd(K.full(xd), wp, backend.put(tscale), backend.put(shift), K.ChanSlice(t, 0, m), 1, 1, 0, True, tile,
                      raw_from=aff, row_split=c1, row_gap=gapc)
        assert rel_err(t[:, :c1], want[:, :c1]) < 2e-5 and rel_err(t[:, cblk:], want[:, c1:]) < 2e-5, tile
        assert (t[:, c1:cblk] == 7.0).all(), "wrote into the block's other branches"
    # dgrad: the gradient tensor has the same layout
    gy = torch.randn(n, m, h, h, generator=g)
    gt = K.guarded_empty((n, cblk + cr, h, h), xd.device)
    gt.fill_(float("nan"))                      # the channels between the two ranges must never be read
    gt[:, :c1] = backend.put(gy[:, :c1])
    gt[:, cblk:] = backend.put(gy[:, c1:])
    K.attach_amax(gt, K.tensor_amax(backend.put(gy)))
    dx = backend.put(torch.zeros(n, cin, h, h))
    K.conv_x6_dgrad(K.ChanSlice(gt, 0, m), wt, K.full(dx), 1, 0, False, k_split=c1, k_gap=gapc)
    assert rel_err(dx, torch.nn.grad.conv2d_input(x.shape, wcat, gy.double())) < 5e-5
    # wgrad
    dw, db = backend.put(torch.empty(m, cin, 1, 1)), backend.put(torch.empty(m))
    wsz = backend.put(torch.empty(K.wgrad_x6_workspace_bytes(n, cin, m, h, h, 1, -1) // 4))
    K.conv_wgrad_x6(K.ChanSlice(gt, 0, m), K.full(xd), dw, db, 1, 0, wsz, -1, g_row_split=c1, g_row_gap=gapc)
    assert rel_err(dw, torch.nn.grad.conv2d_weight(x.double(), wcat.shape, gy.double())) < 5e-5
    assert rel_err(db, gy.double().sum(dim=(0, 2, 3))) < 5e-5


def test_stem_space_to_depth(backend):
    """The 7x7 / stride-2 / pad-3 stem on the split kernels through its space-to-depth form: forward (4x4 taps on 4C channels,
    2 padding pixels in front / 1 behind), weight + bias gradient (wgrad with 4x4 taps, mapped back to the 7x7 layout)."""
    g = torch.Generator().manual_seed(96)
    for (n, c, h, cout) in ([(4, 3, 224, 64), (2, 10, 64, 64)] if backend.is_gpu else [(1, 3, 16, 32), (2, 2, 12, 32)]):
        x = torch.randn(n, c, h, h, generator=g) * 50
        w = (torch.randn(cout, c, 7, 7, generator=g) * 0.05).double().requires_grad_()
        b = torch.zeros(cout, dtype=torch.double, requires_grad=True)
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
        z = F.conv2d(x.double(), w, b, 2, 3)
        ref = torch.relu(z * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
        gy = torch.randn(z.shape, generator=g)
        z.backward(gy.double())
        xs = K.space_to_depth2(backend.put(x))
        assert xs._ssn_amax.item() == x.abs().max().item()
        xs_ref = x.reshape(n, c, h // 2, 2, h // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, 4 * c, h // 2, h // 2)
        assert torch.equal(xs.cpu(), xs_ref)
        w2 = K.s2d_weights(backend.put(w.detach().float()))
        assert rel_err(F.conv2d(F.pad(xs_ref.double(), (2, 1, 2, 1)), w2.cpu().double()), z.detach() - b.detach().view(1, -1, 1, 1)) < 1e-12
        y = backend.put(torch.zeros(n, cout, h // 2, h // 2))
        K.conv_x6_fwd_rect(K.full(xs), K.pack_weights_rect(w2), backend.put(scale), backend.put(shift), K.full(y), 4, 4, 2, 2, True)
        assert rel_err(y, ref) < 2e-5, (n, c, h)
        dw2 = backend.put(torch.empty(cout, 4 * c, 4, 4))
        db = backend.put(torch.empty(cout))
        ws = backend.put(torch.empty(K.wgrad_x6_workspace_bytes(n, 4 * c, cout, h // 2, h // 2, 4, -1) // 4))
        K.conv_wgrad_x6(K.full(backend.put(gy)), K.full(xs), dw2, db, 4, 2, ws, -1)
        dw = backend.put(torch.empty(cout, c, 7, 7))
        K.s2d_weights_bwd(dw2, dw)
        assert rel_err(dw, w.grad) < 5e-5 and rel_err(db, b.grad) < 5e-5, (n, c, h)


def test_fused_heads_match_the_separate_kernels(backend):
    """functional.HeadsFn (STPP + three Linear heads + prop_type row selection, one launch each way) against the chain of
    StppFn / LinearFn / RowGatherFn it replaces in SSN.train_forward: outputs and every gradient (features, 3 weights, 3 biases),
    for both STPP configurations of the reference, with and without the regression head, and against float64 torch."""
    from action_detection_amd import functional as FN
    from action_detection_amd.ops.ssn_ops import StructuredTemporalPyramidPooling
    g = torch.Generator().manual_seed(31)
    for cfg, with_reg in (((1, 1, 1), True), ((1, (1, 2), 1), True), ((1, 1, 1), False)):
        p, s, d, c = 16, 9, 64, 5
        stpp = StructuredTemporalPyramidPooling(d, True, configs=cfg)
        m = stpp.feat_multiplier
        table = stpp.table_for((2, 7, 9))
        ft = torch.randn(p * s, d, generator=g)
        sc = torch.rand(p, 2, generator=g)
        ws = [torch.randn(c + 1, d, generator=g) * 0.1, torch.randn(c, m * d, generator=g) * 0.1,
              torch.randn(2 * c, m * d, generator=g) * 0.1 if with_reg else None]
        bs = [torch.randn(c + 1, generator=g), torch.randn(c, generator=g), torch.randn(2 * c, generator=g) if with_reg else None]
        ptype = torch.tensor(([0, 1, 1, 1, 1, 1, 1, 2] * 2))
        sets = ((ptype == 0) | (ptype == 2), (ptype == 0) | (ptype == 1), ptype == 0)
        idx = [torch.nonzero(q).reshape(-1) for q in sets]
        pos = [(torch.cumsum(q.to(torch.int32), 0, dtype=torch.int32) - 1).masked_fill(~q, -1) for q in sets]
        gouts = [torch.randn(i.numel(), w.shape[0], generator=g) if w is not None else None for i, w in zip(idx, ws)]

        def leaf(t):
            return None if t is None else backend.put(t).clone().requires_grad_()

        def run(fused):
            f, w_, b_ = leaf(ft), [leaf(t) for t in ws], [leaf(t) for t in bs]
            di = [backend.put(t) for t in idx]
            if fused:
                dp = [backend.put(t) for t in pos]
                outs = FN.HeadsFn.apply(f, backend.put(sc), table, 9, tuple(di) if with_reg else (di[0], di[1], None),
                                        tuple(dp) if with_reg else (dp[0], dp[1], None), w_[0], b_[0], w_[1], b_[1], w_[2], b_[2])
            else:
                a, st = FN.StppFn.apply(f, backend.put(sc), table, 9)
                full = [FN.LinearFn.apply(a, w_[0], b_[0]), FN.LinearFn.apply(st, w_[1], b_[1]),
                        FN.LinearFn.apply(st, w_[2], b_[2]) if with_reg else None]
                outs = [None if o is None else FN.RowGatherFn.apply(o, i) for o, i in zip(full, di)]
            loss = sum((o * backend.put(go)).sum() for o, go in zip(outs, gouts) if o is not None)
            loss.backward()
            return [o for o in outs if o is not None], [t.grad for t in [f] + w_ + b_ if t is not None]

        o1, g1 = run(True)
        o0, g0 = run(False)
        for a, b in zip(o1 + g1, o0 + g0):
            assert rel_err(a, b) < 1e-6, (cfg, with_reg, rel_err(a, b))
        # and against float64 torch
        f64 = ft.double().requires_grad_()
        w64 = [None if t is None else t.double().requires_grad_() for t in ws]
        b64 = [None if t is None else t.double().requires_grad_() for t in bs]
        src = f64.view(p, s, d)
        parts = []
        for lo, hi, norm, col in stpp.part_table((2, 7, 9)):
            v = src[:, lo:hi].mean(1) / norm
            if col >= 0:
                v = v * sc[:, col:col + 1].double()
            parts.append(v)
        act64, st64 = src[:, 2:7].mean(1), torch.cat(parts, 1)
        outs64 = [(act64 @ w64[0].t() + b64[0])[idx[0]], (st64 @ w64[1].t() + b64[1])[idx[1]],
                  (st64 @ w64[2].t() + b64[2])[idx[2]] if with_reg else None]
        sum((o * go.double()).sum() for o, go in zip(outs64, gouts) if o is not None).backward()
        ref = [o for o in outs64 if o is not None] + [t.grad for t in [f64] + w64 + b64 if t is not None]
        for a, b in zip(o1 + g1, ref):
            assert rel_err(a, b) < 2e-6, (cfg, with_reg, "float64", rel_err(a, b))
