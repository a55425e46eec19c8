"""TEST INFRASTRUCTURE: a five-layer backbone on the product's planes executor, small enough for the host emulator.

``TinyBackbone`` is a ``bninception.BNInception`` whose manifest is five ops -- 3x3 convolution on the frames, 3x3 / stride-2 ceil
max pool, a 1x1 and a 3x3 branch writing the two halves of a concat-free output, global average pool -- so every code path of
``planes_exec.run_forward / run_backward`` (delayed scales, calibration, the range guard, fused ReLU / frozen-BN masks, max-pool
backward, gradient accumulation into a shared input, ready ranges for a reducer) runs in seconds on the CPU tier.  ``TinyRef`` is
the same network in float64 torch (the referee); with ``forced`` it takes the product forward's ReLU / max-pool decisions
(``BNInception.export_decisions``), like ``oracle.ssn_oracle.OracleBNInception.forced``.
"""
import torch
import torch.nn.functional as F
from torch import nn

from action_detection_amd.bninception import BNInception

C1, CA, CB = 16, 16, 16


def tiny_manifest(in_channels, size):
    hp = -(-(size - 3) // 2) + 1
    if (hp - 1) * 2 >= size:
        hp -= 1
    ops = [("conv", "conv1_3x3", "data", "t1", 0, in_channels, C1, 3, 1, 1),
           ("pool", "pool1_3x3_s2", "max", "t1", "t2", 0, 3, 2, 0, True),
           ("conv", "inception_t_1x1", "t2", "out", 0, C1, CA, 1, 1, 0),
           ("conv", "branch_3x3", "t2", "out", CA, C1, CB, 3, 1, 1),
           ("gap", "global_pool", "out", "global_pool")]
    tensors = {"data": (in_channels, size, size), "t1": (C1, size, size), "t2": (C1, hp, hp), "out": (CA + CB, hp, hp),
               "global_pool": (CA + CB, 1, 1)}
    return ops, tensors


class TinyBackbone(BNInception):
    def __init__(self, in_channels=3, input_size=16):
        nn.Module.__init__(self)
        self.in_channels = in_channels
        self.input_size_hint = input_size
        self._conv_ids = []
        for op in tiny_manifest(in_channels, input_size)[0]:
            if op[0] == "conv":
                _, lid, _, _, _, cin, cout, k, s, p = op
                setattr(self, lid, nn.Conv2d(cin, cout, k, s, p, bias=True))
                setattr(self, lid + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
                self._conv_ids.append(lid)
        self.fc = nn.Identity()
        self._init_executor()
        self.layout = "planes"

    def _manifest(self, x):
        return tiny_manifest(self.in_channels, x.shape[2])

    def train(self, mode=True):      # frozen BatchNorm, as SSN.train() leaves the backbone (ssn_models.py:156-174)
        super().train(mode)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
                m.weight.requires_grad = False
                m.bias.requires_grad = False
        return self


class TinyRef(nn.Module):
    """float64 torch restatement of TinyBackbone (parameters copied with load_state_dict)."""

    def __init__(self, in_channels=3):
        super().__init__()
        for lid, cin, cout, k, p in (("conv1_3x3", in_channels, C1, 3, 1), ("inception_t_1x1", C1, CA, 1, 0),
                                     ("branch_3x3", C1, CB, 3, 1)):
            setattr(self, lid, nn.Conv2d(cin, cout, k, 1, p, bias=True))
            setattr(self, lid + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
        self.forced = None
        self.double().eval()
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.requires_grad = False
                m.bias.requires_grad = False

    def _cbr(self, lid, x):
        z = getattr(self, lid + "_bn")(getattr(self, lid)(x))
        if self.forced is not None:
            return z * self.forced[0][lid].to(z.dtype)
        return F.relu(z)

    def forward(self, x):
        x = self._cbr("conv1_3x3", x.double())
        if self.forced is None:
            x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
        else:
            local = self.forced[1]["pool1_3x3_s2"]
            n, c, h, w = x.shape
            ho, wo = local.shape[2], local.shape[3]
            hh = torch.arange(ho).view(1, 1, ho, 1) * 2 + local // 3
            ww = torch.arange(wo).view(1, 1, 1, wo) * 2 + local % 3
            x = x.flatten(2).gather(2, (hh * w + ww).flatten(2)).view(n, c, ho, wo)
        y = torch.cat([self._cbr("inception_t_1x1", x), self._cbr("branch_3x3", x)], 1)
        return y.mean(dim=(2, 3))


def init_tiny(net, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            elif isinstance(m, nn.BatchNorm2d):
                c = m.num_features
                m.weight.copy_(torch.rand(c, generator=g) + 0.5)
                m.bias.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return net
