"""Inception-v3 backbone executor (forward only): the stand-in for ``model_zoo.InceptionV3``.

The reference builds it with ``getattr(model_zoo, 'InceptionV3')()`` and replaces its ``top_cls_fc``
(/root/reference/ssn_models.py:133-139, 69-74); its tester scores 299x299 crops with it on ActivityNet
(/root/reference/ssn_test.py:57-83, BASELINE.json configs[4]).  Same parameter surface convention as
``bninception.BNInception`` (``<layer>.weight/.bias``, ``<layer>_bn.*``, ``top_cls_fc``); ``features`` walks the
manifest of ``inceptionv3_spec`` and launches the gfx950 kernels:

  * first conv (3 input channels): exact-f32 MFMA kernel (``ssn_conv_bn_relu_fwd``);
  * 1x1 / 3x3 layers (stride 1 or 2, pad 0 or 1): bf16-split kernel (``ssn_conv_x6_fwd``);
  * 5x5, 1x7, 7x1, 1x3, 3x1 layers: the same kernel with rectangular taps (``ssn_conv_x6_fwd_rect``);
  * pools and the global average pool: ``ssn_pool_fwd``, ``ssn_global_avgpool_fwd``;
  branches write straight into their channel slice of the block output (no concat).

Only inference is built (the reference trains SSN on Inception-v3 too, but the MI355X backward covers
BN-Inception, BASELINE.json configs 1-4): calling it with gradients enabled raises.
"""
import torch
from torch import nn

from . import kernels as K
from .inceptionv3_spec import FEATURE_DIM, build_manifest
from .kernels import ChanSlice, full


class InceptionV3(nn.Module):
    """Drop-in for ``model_zoo.InceptionV3`` (ctor signature of the upstream zoo: num_classes)."""

    def __init__(self, num_classes=1000, in_channels=3, input_size=299):
        super().__init__()
        self.in_channels = in_channels
        ops, _ = build_manifest(in_channels, input_size)
        self._conv_ids = []
        for op in ops:
            if op[0] == "conv":
                _, lid, _, _, _, cin, cout, kh, kw, s, ph, pw = op
                setattr(self, lid, nn.Conv2d(cin, cout, (kh, kw), s, (ph, pw), bias=True))
                setattr(self, lid + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
                self._conv_ids.append(lid)
        self.top_cls_fc = nn.Linear(FEATURE_DIM, num_classes)
        self._packed = {}     # layer id -> (weight data_ptr, version, packed operand)

    def _packed_weight(self, lid, conv, kind):
        w = conv.weight
        key = (w.data_ptr(), w._version, kind)
        hit = self._packed.get(lid)
        if hit is not None and hit[0] == key:
            return hit[1]
        wd = w.detach()
        kh, kw = wd.shape[2], wd.shape[3]
        if kind == "f32":
            wp = K.pack_weights(wd, False)
        elif kind == "x6":
            (wp,) = K.pack_weights_multi([([wd], 0)], x6=True)
        else:
            wp = K.pack_weights_rect(wd)
        self._packed[lid] = (key, wp)
        return wp

    def features(self, x):
        if x.dim() != 4:
            raise ValueError("expected NCHW input")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError(
                "the Inception-v3 backbone is built for inference (dense testing, torch.no_grad()); the MI355X "
                "backward covers BN-Inception")
        for lid in self._conv_ids:
            if getattr(self, lid + "_bn").training:
                raise NotImplementedError("training-mode BatchNorm is not built; call .eval() (SSN keeps BN frozen)")
        first = getattr(self, self._conv_ids[0])
        if x.shape[1] != first.in_channels:
            raise ValueError("input has %d channels, first conv expects %d" % (x.shape[1], first.in_channels))
        if x.shape[2] != x.shape[3]:
            raise ValueError("square inputs only")
        ops, shapes = build_manifest(first.in_channels, x.shape[2])
        x = x.contiguous().float()
        n, dev = x.shape[0], x.device
        acts = {"data": x}

        # one amax slot per activation tensor (kernels.py: "amax slots"), handed out from one zeroed pool
        slot_pool = torch.zeros(len(shapes) + 1, device=dev, dtype=torch.float32)
        slot_of = {}

        def get(name):
            if name not in acts:
                c, h, w = shapes[name]
                i = slot_of.setdefault(name, len(slot_of))
                acts[name] = K.attach_amax(K.guarded_empty((n, c, h, w), dev), slot_pool[i:i + 1])
            return acts[name]

        # folded frozen-BN affine of all layers in one launch
        conv_ops = [op for op in ops if op[0] == "conv"]
        total = sum(op[6] for op in conv_ops)
        scale_flat = torch.empty(total, device=dev, dtype=torch.float32)
        shift_flat = torch.empty(total, device=dev, dtype=torch.float32)
        fold = ([], [], [], [], [], [], [], [])
        aff, off = {}, 0
        for op in conv_ops:
            lid, cout = op[1], op[6]
            conv, bn = getattr(self, lid), getattr(self, lid + "_bn")
            aff[lid] = (scale_flat[off:off + cout], shift_flat[off:off + cout])
            for lst, v in zip(fold, (conv.bias.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                     bn.running_var, bn.eps, aff[lid][0], aff[lid][1])):
                lst.append(v)
            off += cout
        K.bn_fold_multi(*fold)

        last_use = {}
        for i, op in enumerate(ops):
            last_use[op[3] if op[0] == "pool" else op[2]] = i
        feat = None
        for i, op in enumerate(ops):
            if op[0] == "conv":
                _, lid, src, dst, c0, cin, cout, kh, kw, s, ph, pw = op
                conv = getattr(self, lid)
                scale, shift = aff[lid]
                xs = ChanSlice(acts[src], 0, cin)
                ys = ChanSlice(get(dst), c0, cout)
                if cin < 16:
                    assert kh == kw and ph == pw
                    K.conv_fwd(xs, self._packed_weight(lid, conv, "f32"), scale, shift, ys, kh, s, ph, True)
                elif kh == kw and kh in (1, 3):
                    assert ph == pw
                    K.conv_x6_fwd(xs, self._packed_weight(lid, conv, "x6"), scale, shift, ys, kh, s, ph, True)
                else:
                    assert s == 1
                    K.conv_x6_fwd_rect(xs, self._packed_weight(lid, conv, "rect"), scale, shift, ys, kh, kw, ph, pw, True)
            elif op[0] == "pool":
                _, lid, kind, src, dst, c0, k, s, p = op
                c = shapes[src][0]
                K.pool_fwd(kind, full(acts[src]), ChanSlice(get(dst), c0, c), None, k, s, p)
            else:
                _, lid, src, dst = op
                feat = torch.empty((n, shapes[src][0]), device=dev, dtype=torch.float32)
                K.gap_fwd(full(acts[src]), feat)
            # drop activations as soon as their last consumer has been launched (same stream: safe to reuse)
            for name in [nm for nm, last in last_use.items() if last == i and nm != "data"]:
                acts.pop(name, None)
        return feat

    def forward(self, x):
        return self.top_cls_fc(self.features(x))
