"""End-to-end parity of the HIP path with the CPU oracle at the real 224x224 resolution (-m gpu).

BASELINE.json configs: [0] RGB 1 video x 9 segments forward (plumbing), [1] RGB 32 proposals (here the
oracle-sized 16-proposal slice; the full 288-frame batch is checked through size-independent
properties), [2] Flow.  Tolerance: logits / losses 1e-4 relative (north star); gradients 1e-3 relative
per tensor (they pass through ~140 fp32 reductions in a different summation order).
"""
import pytest
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss
from action_detection_amd.ssn_models import SSN
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch
from test_kernels import rel_err

pytestmark = pytest.mark.gpu


def build(modality, cfg, dropout=0.0, num_class=20, negative_gamma_frac=0.0):
    torch.manual_seed(0)
    m = SSN(num_class, 2, 5, 2, modality, dropout=dropout, stpp_cfg=cfg)
    init_backbone_synthetic(m.base_model, negative_gamma_frac=negative_gamma_frac)
    init_heads_synthetic(m)
    o = O.OracleSSN(num_class, 2, 5, 2, modality, dropout=dropout, stpp_cfg=cfg)
    o.load_state_dict(m.state_dict())
    return m.to("cuda:0").train(), o.train()


def losses(out, v):
    return (ActivityLoss()(out[0], out[1]), CompletenessLoss()(out[2], out[3], 1, 7),
            ClassWiseRegressionLoss()(out[4], out[5], out[6]))


@pytest.mark.parametrize("modality,cfg,v,neg", [("RGB", (1, 1, 1), 4, 0.0), ("Flow", (1, (1, 2), 1), 4, 0.0),
                                                ("Flow", (1, 1, 1), 2, 0.25)])
def test_fwd_bwd_matches_oracle(hip_library, modality, cfg, v, neg):
    """BASELINE configs 2 (RGB) and 3 (Flow) at FULL size (4 videos = 32 proposals = 288 frames of 224^2), and a Flow slice
    with a quarter of the frozen-BN scales negative.  Logits / losses 1e-4 against the fp32 CPU oracle; gradients against the
    oracle, a float64 referee, and -- the tight check -- a MASK-FORCED float64 referee (below).

    Why the gradient check is a distribution and not a per-tensor bound: the loss is only piecewise smooth in the
    weights -- a ReLU (or max-pool) unit whose pre-activation is within rounding of zero takes a different branch in
    two fp32 implementations, and the gradient of every layer below it moves by a finite amount.  The reference's own
    torch-CPU fp32 path is off the float64 gradient by 2.4e-3 on one tensor for that reason
    (profiles/r2_grad_flip_diag.txt; identical numbers with the exact-f32 MFMA kernels, so it is not a property of the
    operand split).  Which units flip is an accident of rounding; how MANY tensors are affected and how close the bulk
    is to float64 is what a correct implementation controls.  Asserted: every tensor within 5e-3 of the fp32 oracle and of float64
    (fixed a priori), and -- the bar that does not depend on which units flip -- every tensor within 5e-5 of the float64 referee that
    is forced to take the HIP forward's ReLU / max-pool decisions."""
    m, o = build(modality, cfg, negative_gamma_frac=neg)
    m.base_model.debug_keep_saved = True
    batch = make_batch(v, modality, 20, seed=5)
    out = m(*[t.cuda() for t in batch])
    ref = o(*batch)
    for i, (a, b) in enumerate(zip(out, ref)):
        if i % 2 == 1 or i == 6:
            assert torch.equal(a.cpu(), b), i
        else:
            assert rel_err(a, b) < 1e-4, (i, rel_err(a, b))
    a, c, r = losses(out, v)
    total = a + 0.1 * c + 0.1 * r
    rt, ra, rc, rr = O.ssn_total_loss(ref, v)
    for got, want in ((a, ra), (c, rc), (r, rr), (total, rt)):
        assert abs(got.item() - want.item()) <= 1e-4 * abs(want.item()) + 1e-7
    total.backward()
    rt.backward()
    o64 = O.OracleSSN(20, 2, 5, 2, modality, dropout=0, stpp_cfg=cfg).double()
    o64.load_state_dict({k: t.double() for k, t in o.state_dict().items()})
    o64.train()
    b64 = [t.double() if t.is_floating_point() else t for t in batch]
    t64, _, _, _ = O.ssn_total_loss(o64(*b64), v)
    t64.backward()
    ref64 = dict((n, p.grad) for n, p in o64.named_parameters() if p.grad is not None)
    e_hip, e_cpu = [], []
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), o.named_parameters()):
        assert n1 == n2
        if p2.grad is None:
            assert p1.grad is None, n1
            continue
        assert rel_err(p1.grad, p2.grad) < 5e-3, (n1, rel_err(p1.grad, p2.grad))       # hard cap, every tensor
        e_hip.append(rel_err(p1.grad, ref64[n1]))
        e_cpu.append(rel_err(p2.grad, ref64[n1]))
    e_hip, e_cpu = torch.tensor(e_hip), torch.tensor(e_cpu)
    print("gradients vs float64 (%d tensors): HIP median %.2e max %.2e, %d above 1e-3 | torch fp32 CPU median %.2e max "
          "%.2e, %d above 1e-3" % (len(e_hip), e_hip.median(), e_hip.max(), int((e_hip > 1e-3).sum()), e_cpu.median(),
                                   e_cpu.max(), int((e_cpu > 1e-3).sum())))
    assert e_hip.max() < 5e-3
    # (No bar is derived from the torch-fp32 column: a single unit near the top of the network that lands on the other side of zero
    # moves EVERY tensor's error by ~1e-4, in either implementation -- round 4 had to move a "2 x the oracle's median" bar to 3 x after
    # one configuration measured 2.15 x, which is a measurement, not a bar.  The printed line stays as a diagnostic; THE gradient bar
    # is the forced referee below, every tensor to 5e-5, next to the a-priori 5e-3 cap per tensor above.)

    # ---- the tight check: float64 referee with the HIP forward's discrete decisions forced ----
    # The statistical statement above is all that can be said against an INDEPENDENT forward (units within rounding of zero
    # land on different sides).  Forcing the oracle to take the ReLU / max-pool decisions the HIP forward took (the sign its
    # backward kernels read, the argmax its pools stored) removes that term: what is left is a smooth function of the weights,
    # and every gradient tensor must agree with float64 to rounding -- 5e-5 relative per tensor, at the full batch.
    relu, pools = m.base_model.export_decisions()
    o64.base_model.forced = ({k: t.cpu() for k, t in relu.items()}, [t.cpu() for t in pools.values()])
    o64.base_model.audit = {}
    o64.zero_grad(set_to_none=True)
    out_m = o64(*b64)
    for i in (0, 2, 4):
        assert rel_err(out[i], out_m[i]) < 2e-6, ("forced forward", i, rel_err(out[i], out_m[i]))
    # [r6] the exporter itself: the forced decisions may differ from the referee's OWN (sign of its float64 pre-activation, true window
    # maximum) only on units within rounding of the threshold -- a shifted / transposed / stale mask would differ on about half of the
    # units and pass the 5e-5 bar only by accident.  Bars fixed a priori: fewer than 1e-4 of the units, none further from the threshold
    # than 1e-3 of its layer's largest magnitude.
    fr, far, fq, gap = O.audit_summary(o64.base_model.audit)
    print("mask audit: %.2e of the ReLU units differ (furthest %.2e of the layer maximum), %.2e of the pool windows (gap %.2e)" % (fr, far, fq, gap))
    assert fr < 1e-4 and far < 1e-3 and fq < 1e-4 and gap < 1e-3, (fr, far, fq, gap)
    o64.base_model.audit = None
    O.ssn_total_loss(out_m, v)[0].backward()
    worst = ("", 0.0)
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), o64.named_parameters()):
        if p2.grad is None:
            continue
        e = rel_err(p1.grad, p2.grad)
        if e > worst[1]:
            worst = (n1, e)
    print("gradients vs mask-forced float64 referee: worst %s %.2e" % worst)
    assert worst[1] < 5e-5, worst


def test_negative_bn_gammas(hip_library):
    """A quarter of the frozen-BN scales negative (as in trained checkpoints): the fused ReLU/BN backward in the dgrad /
    pool epilogues must apply them with their sign (round 1 used 'scale < 0' as its 'not a ReLU output' marker and
    silently produced wrong backbone gradients for such channels)."""
    m, o = build("RGB", (1, 1, 1), negative_gamma_frac=0.25)
    assert sum(int((b.weight < 0).sum()) for b in o.modules() if isinstance(b, torch.nn.BatchNorm2d)) > 1000
    batch = make_batch(2, "RGB", 20, seed=9)
    out = m(*[t.cuda() for t in batch])
    ref = o(*batch)
    for i in (0, 2, 4):
        assert rel_err(out[i], ref[i]) < 1e-4, (i, rel_err(out[i], ref[i]))
    a, c, r = losses(out, 2)
    (a + 0.1 * c + 0.1 * r).backward()
    O.ssn_total_loss(ref, 2)[0].backward()
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), o.named_parameters()):
        if p2.grad is not None:
            assert rel_err(p1.grad, p2.grad) < 5e-3, (n1, rel_err(p1.grad, p2.grad))


def test_single_video_forward_only(hip_library):
    """configs[0]: 1 video x 8 proposals x 9 segments; the reference's regression loss cannot run at V=1."""
    m, o = build("RGB", (1, 1, 1))
    batch = make_batch(1, "RGB", 20, seed=6)
    out = m(*[t.cuda() for t in batch])
    ref = o(*batch)
    for i in (0, 2, 4):
        assert rel_err(out[i], ref[i]) < 1e-4
    assert out[4].shape == (1, 20, 2) and out[0].shape == (2, 21) and out[2].shape == (7, 20)


def test_full_batch_properties(hip_library):
    """configs[1] at full size (288 frames): properties that do not need the CPU oracle.

    * batch independence: the first 2 videos of a 4-video forward equal a 2-video forward (exactly: same
      tile decomposition is not guaranteed, so 1e-5 relative);
    * determinism: two runs are bit-identical (no float atomics anywhere on the path);
    * dropout=0.8 in train mode zeroes ~80% of the backbone features and is rescaled by 5.
    """
    m, _ = build("RGB", (1, 1, 1))
    b4 = [t.cuda() for t in make_batch(4, "RGB", 20, seed=7)]
    with torch.no_grad():
        o4 = m(*b4)
        o4b = m(*b4)
        o2 = m(*[t[:2].contiguous() for t in b4])
    assert all(torch.equal(x, y) for x, y in zip(o4, o4b))
    assert rel_err(o4[0][:4], o2[0]) < 1e-5 and rel_err(o4[2][:14], o2[2]) < 1e-5
    out = m(*b4)
    a, c, r = losses(out, 4)
    (a + 0.1 * c + 0.1 * r).backward()
    g1 = [p.grad.clone() for p in m.parameters() if p.grad is not None]
    m.zero_grad(set_to_none=True)
    out = m(*b4)
    a, c, r = losses(out, 4)
    (a + 0.1 * c + 0.1 * r).backward()
    g2 = [p.grad for p in m.parameters() if p.grad is not None]
    assert all(torch.equal(x, y) for x, y in zip(g1, g2)), "backward is not deterministic"
    md, _ = build("RGB", (1, 1, 1), dropout=0.8)
    feat = md.base_model.features(b4[0].reshape(-1, 3, 224, 224)[:36])
    dropped = md.base_model.fc(feat)
    frac = (dropped == 0).float().mean().item()
    assert 0.75 < frac < 0.85 + (feat == 0).float().mean().item()
    keep = dropped != 0
    assert rel_err(dropped[keep], feat[keep] * 5.0) < 1e-6


def test_no_cpu_fallback(hip_library):
    m, _ = build("RGB", (1, 1, 1))
    with pytest.raises(RuntimeError):
        m.cpu()(*make_batch(2, "RGB", 20, seed=8, input_size=32))


def test_chunked_execution_matches_single_pass(hip_library):
    """A batch above the 2 GiB a kernel operand can address runs as sub-batches (forward + backward per chunk, flat conv
    gradients summed): same features and parameter gradients as one pass.  (The limit is lowered instead of using 700 frames.)"""
    from action_detection_amd.bninception import BNInception
    torch.manual_seed(0)
    m = BNInception()
    init_backbone_synthetic(m)
    m.eval().cuda()
    x = (torch.randn(7, 3, 224, 224) * 50).cuda()
    w = torch.randn(7, 1024, generator=torch.Generator().manual_seed(2)).cuda()

    def run():
        m.zero_grad(set_to_none=True)
        f = m.features(x)
        (f * w).sum().backward()
        return f.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    f1, g1 = run()
    assert len(m._chunk_bounds(x)) == 2
    m.max_operand_bytes = 3 * 4 * 64 * 112 * 112 + 4096           # three images of the largest activation per chunk
    assert m._chunk_bounds(x) == [0, 2, 5, 7]
    f3, g3 = run()
    assert torch.equal(f1, f3)                                     # per-image results do not depend on the batch they run in
    for n in g1:
        assert rel_err(g3[n], g1[n]) < 2e-5, (n, rel_err(g3[n], g1[n]))   # (sums over 7 images in a different order)


def test_per_layer_wgrad_launches_with_two_lanes_are_bit_identical(hip_library):
    """with the grouped weight-gradient launches AND the deferred reductions switched off, every layer's split-K slabs
    used to share ONE workspace region -- safe on one stream, a race once the 3x3 branch's weight gradient runs on the side lane beside
    double_3x3_1/2's on the main one.  With lanes on each layer owns its region: the gradients must equal the one-stream run's bit for
    bit (same kernels, same split plans, same reduction order), three times in a row."""
    from action_detection_amd.bninception import BNInception
    torch.manual_seed(0)
    m = BNInception()
    init_backbone_synthetic(m)
    m.eval().cuda()
    m.group_wgrad = False
    m.defer_wgrad_reduce = False
    x = (torch.randn(6, 3, 224, 224) * 50).cuda()
    w = torch.randn(6, 1024, generator=torch.Generator().manual_seed(2)).cuda()

    def run(lanes):
        m.branch_lanes = lanes
        m.zero_grad(set_to_none=True)
        f = m.features(x)
        (f * w).sum().backward()
        torch.cuda.synchronize()
        return f.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    run(False)                      # (calibrates the delayed scales of both passes)
    f0, g0 = run(False)
    for _ in range(3):
        f1, g1 = run(True)
        assert torch.equal(f0, f1)
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n
