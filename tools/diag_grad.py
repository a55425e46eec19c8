"""GPU diagnostic: which executor option moves the HIP gradients away from the float64 referee?
(V = 2 RGB 224^2, the case of tests/test_model_gpu.py::test_fwd_bwd_matches_oracle.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import action_detection_amd  # noqa: F401,E402
import ssn_oracle as O  # noqa: E402
from test_model_gpu import build, losses  # noqa: E402
from action_detection_amd.synthetic import make_batch  # noqa: E402


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main():
    v, cfg = 2, (1, 1, 1)
    m, o = build("RGB", cfg)
    batch = make_batch(v, "RGB", 20, seed=5)
    t0 = time.time()
    o64 = O.OracleSSN(20, 2, 5, 2, "RGB", dropout=0, stpp_cfg=cfg).double()
    o64.load_state_dict({k: t.double() for k, t in o.state_dict().items()})
    o64.train()
    b64 = [t.double() if t.is_floating_point() else t for t in batch]
    t64, _, _, _ = O.ssn_total_loss(o64(*b64), v)
    t64.backward()
    ref = dict((n, p.grad) for n, p in o64.named_parameters() if p.grad is not None)
    print("fp64 referee: %.1f s" % (time.time() - t0), flush=True)
    rt, _, _, _ = O.ssn_total_loss(o(*batch), v)
    rt.backward()
    e32 = {n: rel(p.grad, ref[n]) for n, p in o.named_parameters() if p.grad is not None}
    print("torch fp32 CPU worst:", sorted(e32.items(), key=lambda kv: -kv[1])[:3], flush=True)
    bm = m.base_model
    variants = [("default", {}), ("pool order manifest", dict(pool_after_projection=False)),
                ("wgrad f32", dict(wgrad_x6=False)), ("all f32", dict(conv_precision="f32")),
                ("single stream", dict(overlap_wgrad=False, branch_streams=False)),
                ("default again", {})]
    base = dict(pool_after_projection=True, wgrad_x6=True, conv_precision="split", overlap_wgrad=True, branch_streams=True)
    for name, kw in variants:
        for k, val in dict(base, **kw).items():
            setattr(bm, k, val)
        m.zero_grad(set_to_none=True)
        out = m(*[t.cuda() for t in batch])
        a, c, r = losses(out, v)
        (a + 0.1 * c + 0.1 * r).backward()
        torch.cuda.synchronize()
        errs = {n: rel(p.grad, ref[n]) for n, p in m.named_parameters() if p.grad is not None}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
        print("== %-20s n>1e-3: %d   n>3e-4: %d" % (name, sum(e > 1e-3 for e in errs.values()), sum(e > 3e-4 for e in errs.values())))
        for n, e in worst:
            print("     %-50s hip %.2e   cpu32 %.2e" % (n, e, e32[n]), flush=True)


if __name__ == "__main__":
    main()
