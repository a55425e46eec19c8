"""Inception-v3 backbone executor: the stand-in for ``model_zoo.InceptionV3``.

The reference builds it with ``getattr(model_zoo, 'InceptionV3')()`` and replaces its ``top_cls_fc``
(/root/reference/ssn_models.py:133-139, 69-74); its tester scores 299x299 crops with it on ActivityNet
(/root/reference/ssn_test.py:57-83, BASELINE.json configs[4]) and ``ssn_train.py`` trains on it like on BN-Inception
(``loss.backward()``, /root/reference/ssn_train.py:236).  Same parameter surface convention as
``bninception.BNInception`` (``<layer>.weight/.bias``, ``<layer>_bn.*``, ``top_cls_fc``) and the SAME executor: this
class only supplies the manifest of ``inceptionv3_spec`` as a launch plan.  What the plan adds to BN-Inception's:

  * 5x5, 1x7, 7x1, 1x3, 3x1 layers: the split-precision kernel with rectangular taps -- forward
    (``ssn_conv_x6_fwd_rect``), data gradient as the forward correlation with the transposed, tap-reversed operand
    (``ssn_conv_x6_dgrad_rect``: the same kernel instantiations), weight gradient with runtime taps
    (``ssn_conv_wgrad_x6_rect``);
  * unpadded 3x3 layers: stride 1 on the square kernels (dx larger than dy), stride 2 as four parity-class launches with
    the two-tap classes on the even rows / columns (``ssn_conv_x6_dgrad_s2(pad=0)``); their weight gradients and the
    3-channel first layer on the exact-f32 kernels;
  * average-pool branches evaluated behind their 1x1 projection (``BNInception._move_avg_pools``: the pool then runs on
    32-192 channels instead of 192-2048).
"""
from torch import nn

from .bninception import BNInception
from .inceptionv3_spec import FEATURE_DIM, build_manifest


class InceptionV3(BNInception):
    """Drop-in for ``model_zoo.InceptionV3`` (ctor signature of the upstream zoo: num_classes)."""

    def __init__(self, num_classes=1000, in_channels=3, input_size=299):
        nn.Module.__init__(self)
        self.in_channels = in_channels
        self.input_size_hint = input_size
        ops, _ = build_manifest(in_channels, input_size)
        self._conv_ids = []
        for op in ops:
            if op[0] == "conv":
                _, lid, _, _, _, cin, cout, kh, kw, s, ph, pw = op
                setattr(self, lid, nn.Conv2d(cin, cout, (kh, kw), s, (ph, pw), bias=True))
                setattr(self, lid + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
                self._conv_ids.append(lid)
        self.top_cls_fc = nn.Linear(FEATURE_DIM, num_classes)
        self._init_executor()
        self.fuse_block_inputs = True     # reduce pairs / block-input merges of BNInception's plan (False: one launch per layer)

    def forward(self, x):
        return self.top_cls_fc(self.features(x))

    def _manifest(self, x):
        cin = getattr(self, self._conv_ids[0]).in_channels
        if x.shape[1] != cin:
            raise ValueError("input has %d channels, first conv expects %d" % (x.shape[1], cin))
        if x.shape[2] != x.shape[3]:
            raise ValueError("square inputs only")
        return build_manifest(cin, x.shape[2])

    def _build_plan(self, x):
        ops, shapes = self._manifest(x)
        shapes = dict(shapes)
        plan = []
        train_bn = set(self._train_bn_ids())
        blocks_frozen = not (train_bn - {self._conv_ids[0]})      # bn_mode 'partial': only the stem's BatchNorm is in training mode
        fuse = self.fuse_block_inputs and self.conv_precision == "split" and blocks_frozen
        # The two 1x1 "reduce" convolutions of a block (5x5 + double 3x3, 7x7 + double 7x7, 3x3 + 7x7x3 / double 3x3) read the
        # same input and write private tensors: planned as ONE convolution with concatenated output channels into a shared
        # "<block>_reduce" tensor, as in BNInception._plan -- the precondition for its block-input merges below.
        reduces = {}
        for op in ops:
            if op[0] == "conv" and (op[7], op[8], op[9]) == (1, 1, 1) and op[4] == 0 and shapes[op[3]][0] == op[6] \
                    and op[2] != "data" and op[1].endswith("_reduce"):
                reduces.setdefault(op[2], []).append(op)
        mates = {v[0][1]: v[1] for v in reduces.values() if len(v) == 2} if fuse else {}
        skip = {m[1] for m in mates.values()}
        alias = {}       # manifest tensor -> (plan tensor, channel offset)
        for op in ops:
            if op[0] == "conv":
                _, lid, src, dst, c0, cin, cout, kh, kw, s, ph, pw = op
                if lid in skip:
                    continue
                psrc, sc0 = alias.get(src, (src, 0))
                if lid in mates:
                    mop = mates[lid]
                    red = lid[:lid.index("_", 6) + 1] + "reduce"       # "mixed_5b_reduce"
                    shapes[red] = (cout + mop[6],) + tuple(shapes[dst][1:])
                    alias[dst], alias[mop[3]] = (red, 0), (red, cout)
                    del shapes[dst], shapes[mop[3]]
                    plan.append(dict(kind="conv", lids=[lid, mop[1]], src=psrc, src_c0=sc0, cin=cin, dst=red, dst_c0=0,
                                     cout=cout + mop[6], couts=[cout, mop[6]], k=1, s=1, p=0))
                    continue
                square = kh == kw and ph == pw
                plan.append(dict(kind="conv", lids=[lid], src=psrc, src_c0=sc0, cin=cin, dst=dst, dst_c0=c0, cout=cout,
                                 couts=[cout], k=kh if square else 0, s=s, p=ph if square else 0, kh=kh, kw=kw, ph=ph, pw=pw))
            elif op[0] == "pool":
                _, lid, kind, src, dst, c0, k, s, p = op
                plan.append(dict(kind="pool", lid=lid, pool=kind, src=src, dst=dst, dst_c0=c0, c=shapes[src][0], k=k, s=s, p=p))
            else:
                _, lid, src, dst = op
                plan.append(dict(kind="gap", lid=lid, src=src, dst=dst, c=shapes[src][0]))
        if self.pool_after_projection and blocks_frozen:
            merge = fuse and self.merge_projection
            plan = self._move_avg_pools(plan, shapes, merge=merge)
            if merge:        # 1x1 branch + reduce pair + projection: one launch on the block input
                plan = self._merge_block_heads(plan, shapes)
        if train_bn:
            plan = self._split_train_bn(plan, shapes, train_bn)
        return plan, shapes
