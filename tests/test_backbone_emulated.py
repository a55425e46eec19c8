"""Backbone executor through the host emulator (CPU tier, opt-in: ~4 min): forward + backward of BNInception on two
32 x 32 images with a quarter of the BN gammas negative, (1) against the oracle's torch-CPU backbone and (2) with the
average pools behind / in front of their 1x1 projection (the two execution orders of the pool-projection branch must
agree in the features and in every parameter gradient, the projections' bias gradients included)."""
import os

import pytest
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd.bninception import BNInception
from action_detection_amd.synthetic import init_backbone_synthetic



def _run(m, x, w):
    m.zero_grad(set_to_none=True)
    f = m.features(x)
    (f * w).sum().backward()
    return f.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.slow_emu
@pytest.mark.skipif(os.environ.get("SSN_SLOW") != "1", reason="~4 min through the host emulator; set SSN_SLOW=1")
def test_executor_vs_oracle_and_pool_orders(emu):
    torch.manual_seed(0)
    m = BNInception()
    init_backbone_synthetic(m, negative_gamma_frac=0.25)
    m.eval()
    o = O.OracleBNInception()
    o.load_state_dict(m.state_dict())
    o.eval()
    x = torch.randn(2, 3, 32, 32) * 60
    w = torch.randn(2, 1024, generator=torch.Generator().manual_seed(1))
    m.pool_after_projection = True
    f1, g1 = _run(m, x, w)
    m.pool_after_projection = False
    f0, g0 = _run(m, x, w)
    fo = o.features(x)
    (fo * w).sum().backward()
    for f in (f0, f1):
        assert ((f - fo).abs().max() / fo.abs().max()).item() < 1e-5
    assert ((f1 - f0).abs().max() / f0.abs().max()).item() < 2e-6
    worst = 0.0
    for n in g0:
        e = ((g1[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-20)).item()
        worst = max(worst, e)
        assert e < 1e-4, (n, e)
    for n, p in o.named_parameters():
        if p.grad is not None and n.endswith("pool_proj.bias"):
            assert ((g1[n] - p.grad).abs().max() / p.grad.abs().max()).item() < 1e-2, n
    print("pool order: worst gradient difference", worst)


@pytest.mark.slow_emu
@pytest.mark.skipif(os.environ.get("SSN_SLOW") != "1", reason="~4 min through the host emulator; set SSN_SLOW=1")
def test_executor_one_image_vs_oracle(emu):
    """The planes executor end to end -- BN fold, every conv / pool / weight-gradient launch of the
    plan, calibration of the delayed scales -- on ONE 32 x 32 image against the oracle's torch-CPU backbone."""
    torch.manual_seed(0)
    m = BNInception()
    init_backbone_synthetic(m, negative_gamma_frac=0.25)
    m.eval()
    o = O.OracleBNInception()
    o.load_state_dict(m.state_dict())
    o.eval()
    x = torch.randn(1, 3, 32, 32) * 60
    w = torch.randn(1, 1024, generator=torch.Generator().manual_seed(1))
    f, g = _run(m, x, w)
    fo = o.features(x)
    (fo * w).sum().backward()
    assert ((f - fo).abs().max() / fo.abs().max()).item() < 1e-5
    errs = []
    for n, p in o.named_parameters():
        if p.grad is not None and n in g:
            errs.append(((g[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-20)).item())
    errs.sort()
    assert len(errs) > 100 and errs[len(errs) // 2] < 1e-4, (len(errs), errs[len(errs) // 2])


def test_fold_bn_scale_vectors(emu):
    """Always-on (seconds): the folded-BN scale vectors of the planes executor -- one NaN-initialised buffer sliced per tensor --
    against a per-layer fold in torch: every conv + BN + ReLU output channel carries gamma / sqrt(var + eps), everything else NaN."""
    from action_detection_amd import planes_exec
    torch.manual_seed(0)
    m = BNInception()
    init_backbone_synthetic(m, negative_gamma_frac=0.25)
    m.eval()
    plan, shapes = m._plan(torch.zeros(1, 3, 224, 224))
    tscale, shift_of, _ = planes_exec._fold_bn(m, plan, shapes, torch.device("cpu"))
    n_checked = 0
    for op in plan:
        if op["kind"] != "conv" or "raw_from" in op or "row_split" in op:
            continue
        dst, c0 = op.get("final", (op["dst"], op["dst_c0"]))
        off = 0
        for lid, c in zip(op["lids"], op["couts"]):
            bn = getattr(m, lid + "_bn")
            ref = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            got = tscale[dst][c0 + off:c0 + off + c]
            assert torch.allclose(got, ref, rtol=1e-6, atol=0), lid
            off += c
            n_checked += 1
    assert n_checked > 30
    assert torch.isnan(tscale["data"]).all() if "data" in tscale else True
    assert all(v.data_ptr() % 32 == 0 for v in tscale.values())
