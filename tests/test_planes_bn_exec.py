"""Training-mode BatchNorm layers on the planes executor (planes_exec.py + csrc/planes_bn.hip).

bn_mode 'partial' leaves the FIRST BatchNorm2d of the backbone in training mode, 'full' all of them
(/root/reference/ssn_models.py:95-105,156-174).  The five-layer backbone of tests/tiny_backbone.py runs both settings on the product's
executor through the host emulator, three different batches in a row (delayed scales, range guard, running statistics that must
advance exactly once per call even when the guard repeats a pass), against the float64 torch module with the same BatchNorm2d
modules in training mode, forced onto the product's ReLU / max-pool decisions.  The full-size check against the reference's own
numbers is tests/test_golden.py::test_product_bn_modes_match_reference_*.
"""
import pytest
import torch

import action_detection_amd  # noqa: F401
from tiny_backbone import TinyBackbone, TinyRef, init_tiny

LAYERS = ("conv1_3x3", "inception_t_1x1", "branch_3x3")


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


def _pair(dev, train_ids, momentum=0.1):
    net = init_tiny(TinyBackbone()).to(dev).train()
    net.debug_keep_saved = True
    ref = TinyRef()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    for lid in train_ids:
        for m in (getattr(net, lid + "_bn"), getattr(ref, lid + "_bn")):
            m.train()
            m.momentum = momentum
            m.weight.requires_grad = True
            m.bias.requires_grad = True
    return net, ref


def _run(dev, train_ids, momentum=0.1):
    net, ref = _pair(dev, train_ids, momentum)
    assert net._train_bn_ids() == list(train_ids)
    g = torch.Generator().manual_seed(5)
    for call, (n, mag) in enumerate([(3, 40.0), (4, 40.0), (3, 900.0)]):      # the last batch leaves the calibrated range
        x = torch.randn(n, 3, 16, 16, generator=g) * mag
        w = torch.randn(n, 32, generator=g)
        net.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        before = {k: v.clone() for k, v in ref.state_dict().items()}
        f = net.features(x.to(dev))
        saved = net._last_saved[0]
        assert len(saved) == 7 and isinstance(saved[5], dict) and set(saved[5]["bnstat"]) == set(train_ids), "planes executor"
        relu, pool = net.export_decisions()
        ref.forced = ({k: t.cpu() for k, t in relu.items()}, {k: t.cpu() for k, t in pool.items()})
        fr = ref(x)
        assert rel_err(f, fr) < 1e-5, ("features", call, rel_err(f, fr))
        (f * w.to(dev)).sum().backward()
        (fr * w.double()).sum().backward()
        for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
            assert n1 == n2
            if p2.grad is None:
                assert p1.grad is None, n1
                continue
            lid = n1.rsplit(".", 1)[0]
            if n1.endswith(".bias") and lid in train_ids:
                # the bias of a convolution in front of a training-mode BatchNorm: an exactly zero gradient, rounding noise on both sides
                gw = dict(net.named_parameters())[lid + ".weight"].grad
                assert float(p1.grad.double().norm()) < 1e-4 * float(gw.double().norm()) + 1e-12, n1
                continue
            assert rel_err(p1.grad, p2.grad) < 5e-5, (n1, call, rel_err(p1.grad, p2.grad))
        # running statistics: exactly one momentum update per call (the guard's repeated passes restart from the old values)
        for lid in LAYERS:
            bn, bnr = getattr(net, lid + "_bn"), getattr(ref, lid + "_bn")
            if lid in train_ids:
                assert rel_err(bn.running_mean, bnr.running_mean) < 1e-5 and rel_err(bn.running_var, bnr.running_var) < 1e-5, lid
                assert int(bn.num_batches_tracked) == int(bnr.num_batches_tracked) == call + 1
                assert not torch.equal(bnr.running_mean, before[lid + "_bn.running_mean"])
            else:
                assert torch.equal(bn.running_mean.cpu().double(), before[lid + "_bn.running_mean"])
        assert not net.scale_fault()
    stats = net.guard_stats()
    assert stats["fwd"] + stats["bwd"] >= 1, "the third batch must have tripped the range guard"
    return net


def test_partial_bn_on_planes(emu):
    _run(torch.device("cpu"), ("conv1_3x3",))


def test_full_bn_on_planes(emu):
    _run(torch.device("cpu"), LAYERS)


def test_cumulative_average_bn_on_planes(emu):
    """BatchNorm2d(momentum=None): torch's cumulative moving average (factor 1 / num_batches_tracked, counted first) -- the running
    statistics must follow torch's module over three calls (incl. the one the range guard repeats)."""
    _run(torch.device("cpu"), ("conv1_3x3", "branch_3x3"), momentum=None)


def test_switch_sends_training_bn_to_the_fp32_layout(emu):
    net, _ = _pair(torch.device("cpu"), ("conv1_3x3",))
    net.planes_train_bn = False
    f = net.features(torch.randn(2, 3, 16, 16))
    assert len(net._last_saved[0]) == 6 and f.shape == (2, 32)


@pytest.mark.gpu
def test_partial_and_full_bn_on_planes_gpu(hip_library):
    _run(torch.device("cuda:0"), ("conv1_3x3",))
    _run(torch.device("cuda:0"), LAYERS)


def test_switching_the_bn_policy_on_one_net_keeps_the_gradients_right(emu):
    """[r6] One backbone object, its BatchNorm policy switched between calls (SSN.train() / .eval() around validation, a test that toggles
    layers): the per-tensor vectors of folded-BN scales are kept across passes (NaN = "not a frozen ReLU / BN output: the fused backward
    leaves this gradient alone"), and an entry a frozen plan wrote must read NaN again once its layer is in training mode.  Round 6's
    first version of that cache was keyed by size only: Inception-v3's fused vs per-layer plans differed by 0.4 in their gradients
    (tests/test_inceptionv3.py::test_inceptionv3_backbone_backward_gpu caught it on the GPU; this is its CPU-tier guard)."""
    dev = torch.device("cpu")
    net, ref = _pair(dev, ())
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 16, 16, generator=g) * 40.0
    w = torch.randn(3, 32, generator=g)

    def check(train_ids):
        for lid in LAYERS:
            for m in (getattr(net, lid + "_bn"), getattr(ref, lid + "_bn")):
                m.train(lid in train_ids)
                m.weight.requires_grad = m.bias.requires_grad = lid in train_ids
        ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
        net.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        f = net.features(x)
        relu, pool = net.export_decisions()
        ref.forced = ({k: t.cpu() for k, t in relu.items()}, {k: t.cpu() for k, t in pool.items()})
        fr = ref(x)
        assert rel_err(f, fr) < 1e-5
        (f * w).sum().backward()
        (fr * w.double()).sum().backward()
        for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
            if p2.grad is None or (n1.endswith(".bias") and n1.rsplit(".", 1)[0] in train_ids):
                continue
            assert rel_err(p1.grad, p2.grad) < 5e-5, (train_ids, n1, rel_err(p1.grad, p2.grad))

    check(())                          # frozen: every scale vector folded
    check(("branch_3x3",))             # that layer's entries must be NaN now
    check(("conv1_3x3",))              # ... and folded again, another layer's NaN
    check(())
