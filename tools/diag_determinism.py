"""GPU diagnostic: is the training step bit-deterministic?  N forward + backward passes of config 2 on the same tensors."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import action_detection_amd  # noqa: F401,E402
from test_model_gpu import build, losses  # noqa: E402
from action_detection_amd.synthetic import make_batch  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m, _ = build("RGB", (1, 1, 1))
batch = [t.cuda() for t in make_batch(4, "RGB", 20, seed=7)]
ref = None
bad = 0
for i in range(reps):
    m.zero_grad(set_to_none=True)
    out = m(*batch)
    a, c, r = losses(out, 4)
    (a + 0.1 * c + 0.1 * r).backward()
    torch.cuda.synchronize()
    cur = [o.detach().clone() for o in out[0::2]] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
    if ref is None:
        ref = cur
    else:
        diff = [j for j, (x, y) in enumerate(zip(ref, cur)) if not torch.equal(x, y)]
        if diff:
            bad += 1
            print("run %d differs in %d tensors (first: %d, max abs diff %.3e)" % (i, len(diff), diff[0], (ref[diff[0]] - cur[diff[0]]).abs().max().item()))
print("determinism: %d of %d repeat runs differ from the first" % (bad, reps - 1))
