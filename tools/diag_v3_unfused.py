"""Diagnostic (round 6): Inception-v3 gradients, fused plan vs one launch per layer -- which tensors differ."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import action_detection_amd  # noqa
from action_detection_amd.inceptionv3 import InceptionV3
from action_detection_amd.synthetic import init_backbone_synthetic
action_detection_amd.build()
torch.manual_seed(0)
prod = InceptionV3(num_classes=10, input_size=299)
init_backbone_synthetic(prod)
prod.eval().to("cuda:0")
g = torch.Generator().manual_seed(3)
x = (torch.randint(0, 256, (4, 3, 299, 299), generator=g).float() - 110.0).cuda()
w = torch.randn(4, 2048, generator=g).cuda()
def run():
    prod.zero_grad(set_to_none=True)
    f = prod.features(x)
    (f * w).sum().backward()
    torch.cuda.synchronize()
    return f.detach().clone(), {n: p.grad.clone() for n, p in prod.named_parameters() if p.grad is not None}
def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300)).item()
for lanes in ([int(a) for a in sys.argv[1:]] or [1]):
    prod.branch_lanes = bool(lanes)
    prod.fuse_block_inputs = True
    f1, g1 = run(); f1b, g1b = run()
    print("lanes", lanes, "fused repeat: feat", rel(f1b, f1), "worst grad", max(rel(g1b[n], g1[n]) for n in g1))
    prod.fuse_block_inputs = False
    f2, g2 = run(); f2b, g2b = run()
    print("   unfused repeat: feat", rel(f2b, f2), "worst grad", max(rel(g2b[n], g2[n]) for n in g2))
    print("   fused vs unfused: feat", rel(f2, f1))
    bad = sorted(((rel(g2[n], g1[n]), n) for n in g1), reverse=True)[:12]
    for e, n in bad:
        print("      %-40s %.3e" % (n, e))
