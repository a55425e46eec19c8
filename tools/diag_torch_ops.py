"""Which torch-native ops (each one a kernel launch + a dependent-launch boundary on the GPU) does a training step issue, and
from where?  Runs SSN training steps (real BN-Inception on the planes path, 1 video) under a TorchDispatchMode and prints every
aten op of the third step that launches something, with its call site.

    python tools/diag_torch_ops.py          (on the MI355X)
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import action_detection_amd as pkg  # noqa: E402,F401
from action_detection_amd import _lib  # noqa: E402
from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss  # noqa: E402
from action_detection_amd.optim import SSNSGD  # noqa: E402
from action_detection_amd.ssn_models import SSN  # noqa: E402
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch  # noqa: E402

pkg.build()
dev = torch.device("cuda:0")
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.empty", "aten.as_strided", "aten.slice", "aten.select", "aten.t.",
        "aten.alias", "aten.expand", "aten.reshape", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.permute",
        "aten._local_scalar_dense", "aten.empty_like", "aten.new_empty", "aten.lift_fresh", "aten.is_same_size", "aten.stride",
        "aten.sym_", "aten.narrow", "aten.unbind", "aten.split", "aten._to_copy.default_cpu", "aten.item", "aten.is_pinned",
        "aten.set_.", "aten.resize_", "aten.empty_strided", "aten.view_as", "aten.is_nonzero", "aten.equal")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()
        self.on = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if self.on and not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "action-detection_amd" in fr.filename or fr.filename.endswith("diag_torch_ops.py"):
                    site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            numel = next((a.numel() for a in args if isinstance(a, torch.Tensor)), 0)
            self.hits[(name, site, numel)] += 1
        return func(*args, **(kwargs or {}))


torch.manual_seed(0)
v, num_class = 1, 20
model = SSN(num_class, 2, 5, 2, "RGB", dropout=0.8, stpp_cfg=(1, 1, 1))
init_backbone_synthetic(model.base_model)
init_heads_synthetic(model, std=0.001)
model.to(dev).train()
opt = SSNSGD(model.get_optim_policies(), lr=0.001, momentum=0.9, weight_decay=5e-4)
crit = (ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss())
batch = [t.to(dev) for t in make_batch(v, "RGB", num_class, seed=0)]


def step():
    out = model(*batch)
    loss = crit[0](out[0], out[1]) + 0.1 * crit[1](out[2], out[3], 1, 7) + 0.1 * crit[2](out[4], out[5], out[6])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


log = Log()
with log:
    step()          # first step: calibration passes, allocations
    step()
    log.on = True
    step()
for (name, site, numel), c in sorted(log.hits.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d x %-34s numel %-9d %s" % (c, name, numel, site))
print("total", sum(log.hits.values()))
