"""CPU oracle for the SSN hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU fp32 / numpy, the algorithm of the reference's
data-parallel hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product package never does (the product fails loudly if its HIP library is missing).

Pinning status
--------------
* STPP, STPPReorgainzed, OHEM hinge / completeness loss, class-wise regression loss,
  the SSN head wiring, ``prepare_test_fc`` folding and ``get_optim_policies`` are pinned
  against the reference's own classes imported from /root/reference (see
  ``oracle/make_golden.py``; fixtures in ``tests/golden/``).
* The BN-Inception backbone is NOT in the reference tree (un-vendored ``model_zoo``
  submodule tracking ``branch=master``, /root/reference/.gitmodules:1-4; no pinned SHA, no
  network).  Its arithmetic is restated from the published BN-Inception topology
  (Ioffe & Szegedy 2015; Caffe layer naming used by yjxiong/tensorflow-model-zoo.torch):
  **backbone parity is unpinned against upstream**; it is well-defined between this oracle
  and the HIP path because both consume the same weights.

Each function cites the reference file:line it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------
# Backbone (restated; see "Pinning status")
# --------------------------------------------------------------------------------------

# (name, 1x1, 3x3reduce, 3x3, dbl_reduce, dbl_a, dbl_b, pool kind, pool proj, stride)
INCEPTION_ROWS = (
    ("3a", 64, 64, 64, 64, 96, 96, "avg", 32, 1),
    ("3b", 64, 64, 96, 64, 96, 96, "avg", 64, 1),
    ("3c", None, 128, 160, 64, 96, 96, "max", None, 2),
    ("4a", 224, 64, 96, 96, 128, 128, "avg", 128, 1),
    ("4b", 192, 96, 128, 96, 128, 128, "avg", 128, 1),
    ("4c", 160, 128, 160, 128, 160, 160, "avg", 128, 1),
    ("4d", 96, 128, 192, 160, 192, 192, "avg", 128, 1),
    ("4e", None, 128, 192, 192, 256, 256, "max", None, 2),
    ("5a", 352, 192, 320, 160, 224, 224, "avg", 128, 1),
    ("5b", 352, 192, 320, 192, 224, 224, "max", 128, 1),
)


def _audit_relu(audit, name, z, mask):
    """Mask audit of a forced evaluation (tests/test_model_gpu.py, __graft_entry__.smoke): where does the FORCED ReLU decision differ from
    the sign of this evaluation's own pre-activation, and how far from zero is the pre-activation there (relative to the layer's
    largest)?  A correct exporter differs only on units within rounding of zero; a wrong one (shifted / transposed / stale masks) on
    about half of them."""
    if audit is None:
        return
    own = z > 0
    bad = own != mask.to(torch.bool)
    nbad = int(bad.sum())
    far = float((z.abs() * bad).max() / (z.abs().max() + 1e-300)) if nbad else 0.0
    audit.setdefault("relu", {})[name] = (nbad, z.numel(), far)


def _audit_pool(audit, idx, x, picked, k, s, p, ceil_mode):
    """... and for a forced max pool: how far below its window's true maximum is the element the forced argmax picked?"""
    if audit is None:
        return
    true = F.max_pool2d(x, k, s, p, ceil_mode=ceil_mode)
    gap = float(((true - picked).abs().max()) / (x.abs().max() + 1e-300))
    audit.setdefault("pool", {})[idx] = (int((true != picked).sum()), picked.numel(), gap)



def audit_summary(audit):
    """(fraction of ReLU units whose forced decision differs, largest |pre-activation| among them relative to its layer's largest,
    fraction of pool windows whose forced pick is not the maximum, largest (maximum - pick) relative to the pool input's largest)"""
    r, q = audit.get("relu", {}), audit.get("pool", {})
    nr, tr = sum(v[0] for v in r.values()), sum(v[1] for v in r.values())
    nq, tq = sum(v[0] for v in q.values()), sum(v[1] for v in q.values())
    return (nr / max(tr, 1), max([v[2] for v in r.values()] + [0.0]), nq / max(tq, 1), max([v[2] for v in q.values()] + [0.0]))


class OracleBNInception(nn.Module):
    """torch-CPU BN-Inception with upstream layer ids as attribute names.

    Every conv has bias=True and is followed by BatchNorm2d(eps=1e-5) + ReLU; all pools
    use ceil_mode=True, average pools count_include_pad=True.
    Exposes ``fc`` (Linear 1024->num_classes) and ``last_layer_name`` handling exactly as
    the reference expects at /root/reference/ssn_models.py:121-127,69-74.
    """

    def __init__(self, num_classes=1000, in_channels=3):
        super().__init__()
        self._order = []

        def add_conv(name, cin, cout, k, stride=1, pad=0):
            setattr(self, name, nn.Conv2d(cin, cout, k, stride, pad, bias=True))
            setattr(self, name + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
            self._order.append(name)

        add_conv("conv1_7x7_s2", in_channels, 64, 7, 2, 3)
        add_conv("conv2_3x3_reduce", 64, 64, 1)
        add_conv("conv2_3x3", 64, 192, 3, 1, 1)
        cin = 192
        for (nm, c1, r3, c3, rd, da, db, pk, pp, st) in INCEPTION_ROWS:
            p = "inception_%s_" % nm
            if c1:
                add_conv(p + "1x1", cin, c1, 1)
            add_conv(p + "3x3_reduce", cin, r3, 1)
            add_conv(p + "3x3", r3, c3, 3, st, 1)
            add_conv(p + "double_3x3_reduce", cin, rd, 1)
            add_conv(p + "double_3x3_1", rd, da, 3, 1, 1)
            add_conv(p + "double_3x3_2", da, db, 3, st, 1)
            if pp:
                add_conv(p + "pool_proj", cin, pp, 1)
            cin = (c1 or 0) + c3 + db + (pp if pp else cin)
        self.fc = nn.Linear(cin, num_classes)

    # Mask-forced evaluation (gradient referee of tests/test_model_gpu.py): with `forced = (relu, pools)` -- {layer id: bool mask},
    # [window-local argmax per max pool, in forward order] taken from ANOTHER implementation's forward -- every ReLU multiplies by
    # the given mask and every max pool picks the given element.  The network is then smooth in its weights, so two correct
    # implementations must agree on the gradients to rounding (no "which side of zero did this unit land on" term).
    forced = None
    audit = None      # a dict: the forced evaluation records where the forced decisions differ from its own (_audit_relu / _audit_pool)

    def _cbr(self, name, x):
        conv = getattr(self, name)
        bn = getattr(self, name + "_bn")
        z = bn(conv(x))
        if self.forced is not None:
            _audit_relu(self.audit, name, z.detach(), self.forced[0][name])
            return z * self.forced[0][name].to(z.dtype)
        return F.relu(z)

    def _maxpool(self, x, k, s, p):
        if self.forced is None:
            return F.max_pool2d(x, k, s, p, ceil_mode=True)
        local = self.forced[1][self._pool_i]
        self._pool_i += 1
        n, c, h, w = x.shape
        ho, wo = local.shape[2], local.shape[3]
        hh = torch.arange(ho).view(1, 1, ho, 1) * s - p + local // k
        ww = torch.arange(wo).view(1, 1, 1, wo) * s - p + local % k
        picked = x.flatten(2).gather(2, (hh * w + ww).flatten(2)).view(n, c, ho, wo)
        _audit_pool(self.audit, self._pool_i - 1, x.detach(), picked.detach(), k, s, p, True)
        return picked

    def features(self, x):
        self._pool_i = 0
        x = self._cbr("conv1_7x7_s2", x)
        x = self._maxpool(x, 3, 2, 0)
        x = self._cbr("conv2_3x3_reduce", x)
        x = self._cbr("conv2_3x3", x)
        x = self._maxpool(x, 3, 2, 0)
        for (nm, c1, r3, c3, rd, da, db, pk, pp, st) in INCEPTION_ROWS:
            p = "inception_%s_" % nm
            outs = []
            if c1:
                outs.append(self._cbr(p + "1x1", x))
            outs.append(self._cbr(p + "3x3", self._cbr(p + "3x3_reduce", x)))
            d = self._cbr(p + "double_3x3_reduce", x)
            d = self._cbr(p + "double_3x3_1", d)
            outs.append(self._cbr(p + "double_3x3_2", d))
            if pp:
                if pk == "avg":
                    q = F.avg_pool2d(x, 3, 1, 1, ceil_mode=True, count_include_pad=True)
                else:
                    q = self._maxpool(x, 3, 1, 1)
                outs.append(self._cbr(p + "pool_proj", q))
            else:
                outs.append(self._maxpool(x, 3, 2, 0))
            x = torch.cat(outs, 1)
        x = F.avg_pool2d(x, x.shape[-1], 1, 0, ceil_mode=True, count_include_pad=True)
        return x.flatten(1)

    def forward(self, x):
        return self.fc(self.features(x))


# --------------------------------------------------------------------------------------
# STPP (training) -- /root/reference/ops/ssn_ops.py:13-79
# --------------------------------------------------------------------------------------

class OracleInceptionV3(nn.Module):
    """torch-CPU Inception-v3 (Szegedy et al. 2016: stem, 3 x block A at 35x35, grid reduction, 4 x block C at 17x17 with
    factorised 7x7, grid reduction, 2 x block E at 8x8 with split 3x3), the backbone the reference's tester loads with
    ``getattr(model_zoo, 'InceptionV3')()`` (/root/reference/ssn_models.py:133-139; last layer ``top_cls_fc``, 2048
    inputs, 299x299).  Not in the reference tree: **unpinned against upstream**, like the BN-Inception above; written
    block by block here, independently of the product's flat manifest (tests compare the two parameter for parameter).
    Every conv has a bias and is followed by BatchNorm2d(eps=1e-5) + ReLU; average pools count the padding."""

    def __init__(self, num_classes=1000, in_channels=3):
        super().__init__()

        def add(name, cin, cout, k, stride=1, pad=0):
            setattr(self, name, nn.Conv2d(cin, cout, k, stride, pad, bias=True))
            setattr(self, name + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
            return cout

        add("conv_1a_3x3", in_channels, 32, 3, 2)
        add("conv_2a_3x3", 32, 32, 3)
        add("conv_2b_3x3", 32, 64, 3, 1, 1)
        add("conv_3b_1x1", 64, 80, 1)
        add("conv_4a_3x3", 80, 192, 3)
        cin = 192
        self.blocks_a = (("mixed_5b", 32), ("mixed_5c", 64), ("mixed_5d", 64))
        for nm, pf in self.blocks_a:
            p = nm + "_"
            add(p + "1x1", cin, 64, 1)
            add(p + "5x5_reduce", cin, 48, 1)
            add(p + "5x5", 48, 64, 5, 1, 2)
            add(p + "double_3x3_reduce", cin, 64, 1)
            add(p + "double_3x3_1", 64, 96, 3, 1, 1)
            add(p + "double_3x3_2", 96, 96, 3, 1, 1)
            add(p + "pool_proj", cin, pf, 1)
            cin = 64 + 64 + 96 + pf
        add("mixed_6a_3x3", cin, 384, 3, 2)
        add("mixed_6a_double_3x3_reduce", cin, 64, 1)
        add("mixed_6a_double_3x3_1", 64, 96, 3, 1, 1)
        add("mixed_6a_double_3x3_2", 96, 96, 3, 2)
        cin = 384 + 96 + cin
        self.blocks_c = (("mixed_6b", 128), ("mixed_6c", 160), ("mixed_6d", 160), ("mixed_6e", 192))
        for nm, c7 in self.blocks_c:
            p = nm + "_"
            add(p + "1x1", cin, 192, 1)
            add(p + "7x7_reduce", cin, c7, 1)
            add(p + "1x7", c7, c7, (1, 7), 1, (0, 3))
            add(p + "7x1", c7, 192, (7, 1), 1, (3, 0))
            add(p + "double_7x7_reduce", cin, c7, 1)
            add(p + "double_7x1_1", c7, c7, (7, 1), 1, (3, 0))
            add(p + "double_1x7_1", c7, c7, (1, 7), 1, (0, 3))
            add(p + "double_7x1_2", c7, c7, (7, 1), 1, (3, 0))
            add(p + "double_1x7_2", c7, 192, (1, 7), 1, (0, 3))
            add(p + "pool_proj", cin, 192, 1)
            cin = 768
        add("mixed_7a_3x3_reduce", cin, 192, 1)
        add("mixed_7a_3x3", 192, 320, 3, 2)
        add("mixed_7a_7x7x3_reduce", cin, 192, 1)
        add("mixed_7a_7x7x3_1x7", 192, 192, (1, 7), 1, (0, 3))
        add("mixed_7a_7x7x3_7x1", 192, 192, (7, 1), 1, (3, 0))
        add("mixed_7a_7x7x3_3x3", 192, 192, 3, 2)
        cin = 320 + 192 + cin
        for nm in ("mixed_7b", "mixed_7c"):
            p = nm + "_"
            add(p + "1x1", cin, 320, 1)
            add(p + "3x3_reduce", cin, 384, 1)
            add(p + "3x3_1x3", 384, 384, (1, 3), 1, (0, 1))
            add(p + "3x3_3x1", 384, 384, (3, 1), 1, (1, 0))
            add(p + "double_3x3_reduce", cin, 448, 1)
            add(p + "double_3x3_1", 448, 384, 3, 1, 1)
            add(p + "double_3x3_1x3", 384, 384, (1, 3), 1, (0, 1))
            add(p + "double_3x3_3x1", 384, 384, (3, 1), 1, (1, 0))
            add(p + "pool_proj", cin, 192, 1)
            cin = 2048
        self.top_cls_fc = nn.Linear(2048, num_classes)

    # Mask-forced evaluation (see OracleBNInception.forced): ({layer id: bool mask}, [window-local argmax per max pool, in forward
    # order]) taken from another implementation's forward; every ReLU multiplies by the mask, every max pool gathers the given element.
    forced = None
    audit = None

    def _cbr(self, name, x):
        z = getattr(self, name + "_bn")(getattr(self, name)(x))
        if self.forced is not None:
            _audit_relu(self.audit, name, z.detach(), self.forced[0][name])
            return z * self.forced[0][name].to(z.dtype)
        return F.relu(z)

    def _maxpool(self, x):
        if self.forced is None:
            return F.max_pool2d(x, 3, 2)
        local = self.forced[1][self._pool_i]
        self._pool_i += 1
        n, c, h, w = x.shape
        ho, wo = local.shape[2], local.shape[3]
        hh = torch.arange(ho).view(1, 1, ho, 1) * 2 + local // 3
        ww = torch.arange(wo).view(1, 1, 1, wo) * 2 + local % 3
        picked = x.flatten(2).gather(2, (hh * w + ww).flatten(2)).view(n, c, ho, wo)
        _audit_pool(self.audit, self._pool_i - 1, x.detach(), picked.detach(), 3, 2, 0, False)
        return picked

    def features(self, x):
        c = self._cbr
        self._pool_i = 0
        x = c("conv_2b_3x3", c("conv_2a_3x3", c("conv_1a_3x3", x)))
        x = self._maxpool(x)
        x = c("conv_4a_3x3", c("conv_3b_1x1", x))
        x = self._maxpool(x)
        for nm, _ in self.blocks_a:
            p = nm + "_"
            x = torch.cat([c(p + "1x1", x), c(p + "5x5", c(p + "5x5_reduce", x)),
                           c(p + "double_3x3_2", c(p + "double_3x3_1", c(p + "double_3x3_reduce", x))),
                           c(p + "pool_proj", F.avg_pool2d(x, 3, 1, 1, count_include_pad=True))], 1)
        p = "mixed_6a_"
        x = torch.cat([c(p + "3x3", x), c(p + "double_3x3_2", c(p + "double_3x3_1", c(p + "double_3x3_reduce", x))),
                       self._maxpool(x)], 1)
        for nm, _ in self.blocks_c:
            p = nm + "_"
            b7 = c(p + "7x1", c(p + "1x7", c(p + "7x7_reduce", x)))
            bd = c(p + "double_7x7_reduce", x)
            for s in ("double_7x1_1", "double_1x7_1", "double_7x1_2", "double_1x7_2"):
                bd = c(p + s, bd)
            x = torch.cat([c(p + "1x1", x), b7, bd,
                           c(p + "pool_proj", F.avg_pool2d(x, 3, 1, 1, count_include_pad=True))], 1)
        p = "mixed_7a_"
        b7 = c(p + "7x7x3_reduce", x)
        for s in ("7x7x3_1x7", "7x7x3_7x1", "7x7x3_3x3"):
            b7 = c(p + s, b7)
        x = torch.cat([c(p + "3x3", c(p + "3x3_reduce", x)), b7, self._maxpool(x)], 1)
        for nm in ("mixed_7b", "mixed_7c"):
            p = nm + "_"
            b3 = c(p + "3x3_reduce", x)
            bd = c(p + "double_3x3_1", c(p + "double_3x3_reduce", x))
            x = torch.cat([c(p + "1x1", x), c(p + "3x3_1x3", b3), c(p + "3x3_3x1", b3),
                           c(p + "double_3x3_1x3", bd), c(p + "double_3x3_3x1", bd),
                           c(p + "pool_proj", F.avg_pool2d(x, 3, 1, 1, count_include_pad=True))], 1)
        return F.adaptive_avg_pool2d(x, 1).flatten(1)

    def forward(self, x):
        return self.top_cls_fc(self.features(x))


def parse_stage_config(cfg):
    """/root/reference/ops/ssn_ops.py:13-19."""
    if isinstance(cfg, int):
        return (cfg,), cfg
    if isinstance(cfg, (tuple, list)):
        return tuple(cfg), sum(cfg)
    raise ValueError("Incorrect STPP config {}".format(cfg))


def stage_ticks(stage_len, n_part):
    """Integer part boundaries of one pyramid level.

    Follows /root/reference/ops/ssn_ops.py:53-55: ``torch.arange(0, len + 1e-5, len / n_part)``
    (float32 result of a double-precision ``start + i*step``) followed by ``int()``.
    Restated with numpy float64 -> float32 -> truncation.
    """
    step = stage_len / n_part
    n = int(math.ceil((stage_len + 1e-5) / step))
    vals = (np.arange(n, dtype=np.float64) * step).astype(np.float32)
    return [int(v) for v in vals]


def stpp_part_table(seg_split, configs):
    """List of (seg_lo, seg_hi, norm, scale_col) for every STPP output part, in output order.

    scale_col: 0 -> multiply by scaling[:,0] (starting stage), 1 -> scaling[:,1] (ending),
    -1 -> unscaled (course stage).  /root/reference/ops/ssn_ops.py:49-64.
    """
    x1, x2, n_seg = seg_split
    bounds = ((0, x1, 0), (x1, x2, -1), (x2, n_seg, 1))
    table = []
    for (lo, hi, col), cfg in zip(bounds, configs):
        parts, mult = parse_stage_config(cfg)
        length = hi - lo
        for n_part in parts:
            t = stage_ticks(length, n_part)
            for i in range(n_part):
                table.append((lo + t[i], lo + t[i + 1], mult, col))
    return table


def stpp_forward(ft, scaling, seg_split, configs=(1, (1, 2), 1), standalone_classifier=True):
    """/root/reference/ops/ssn_ops.py:39-70.  ft [P*S, D], scaling [..., 2] -> (act_ft, stpp_ft)."""
    n_seg = seg_split[2]
    d = ft.shape[1]
    src = ft.reshape(-1, n_seg, d)
    scaling = scaling.reshape(-1, 2)
    feats = []
    for lo, hi, norm, col in stpp_part_table(seg_split, configs):
        part = src[:, lo:hi, :].mean(dim=1) / norm
        if col >= 0:
            part = part * scaling[:, col].reshape(-1, 1)
        feats.append(part)
    stpp_ft = torch.cat(feats, dim=1)
    if not standalone_classifier:
        return stpp_ft, stpp_ft
    course = src[:, seg_split[0]:seg_split[1], :].mean(dim=1)
    return course, stpp_ft


# --------------------------------------------------------------------------------------
# STPPReorgainzed (dense testing) -- /root/reference/ops/ssn_ops.py:82-170
# --------------------------------------------------------------------------------------

def stpp_reorganized(scores, proposal_ticks, scaling, act_len, comp_len, reg_len,
                     stpp_cfg=(1, 1, 1), standalone_classifier=True, with_regression=True):
    """scores [T, D] fp32, proposal_ticks [P,4] int, scaling [P,2] -> (act, comp, reg)."""
    scores = np.asarray(scores, dtype=np.float32)
    ticks_all = np.asarray(proposal_ticks).astype(np.int64)
    scaling = np.asarray(scaling)
    cfg = [parse_stage_config(c)[0] for c in stpp_cfg]
    mult = sum(sum(c) for c in cfg)
    a_stop = act_len if standalone_classifier else act_len * mult
    c_stop = a_stop + comp_len * mult
    r_stop = c_stop + reg_len * mult
    assert scores.shape[1] == r_stop if with_regression else scores.shape[1] >= c_stop
    T = scores.shape[0]
    P = ticks_all.shape[0]

    def pspool(raw, ticks, sc, score_len):
        out = np.zeros(score_len, dtype=np.float32)
        offset = 0
        for si, stage in enumerate(cfg):
            s = sc[0] if si == 0 else (sc[1] if si == len(cfg) - 1 else 1.0)
            left = int(ticks[si])
            right = int(max(ticks[si] + 1, ticks[si + 1]))
            if right <= 0 or left >= T:
                offset += sum(stage)
                continue
            for n_part in stage:
                pt = np.arange(left, right + 1e-5, (right - left) / n_part)
                for i in range(n_part):
                    pl, pr = int(pt[i]), int(pt[i + 1])
                    if pr - pl >= 1:
                        blk = raw[pl:pr, offset * score_len:(offset + 1) * score_len]
                        out += (torch.from_numpy(np.ascontiguousarray(blk)).mean(dim=0).numpy()
                                * np.float32(s)).astype(np.float32)
                    offset += 1
        return out

    act = np.zeros((P, act_len), np.float32)
    comp = np.zeros((P, comp_len), np.float32)
    reg = np.zeros((P, reg_len), np.float32) if with_regression else None
    raw_a, raw_c, raw_r = scores[:, :a_stop], scores[:, a_stop:c_stop], scores[:, c_stop:r_stop]
    for i in range(P):
        tk = ticks_all[i]
        if standalone_classifier:
            lo, hi = int(tk[1]), int(max(tk[1] + 1, tk[2]))
            act[i] = torch.from_numpy(np.ascontiguousarray(raw_a[lo:hi])).mean(dim=0).numpy()
        else:
            act[i] = pspool(raw_a, tk, scaling[i], act_len)
        comp[i] = pspool(raw_c, tk, scaling[i], comp_len)
        if with_regression:
            reg[i] = pspool(raw_r, tk, scaling[i], reg_len)
    return act, comp, reg


def dense_test_video(net, frames_gen, frame_cnt, prop_ticks, prop_scaling, num_class, num_crop=10, stats=None,
                     stpp_cfg=(1, 1, 1)):
    """The per-video body of the reference tester, /root/reference/ssn_test.py:66-92, on ``net`` = an OracleSSN in
    test mode with prepare_test_fc() done: score every frame batch (:81-83), average the crops AFTER the folded FC
    (:84-85), re-organised STPP (:87), regression de-normalisation with ``stats`` (:88-90)."""
    length = (3 if net.modality == "RGB" else 2) * net.new_length
    output_dim = net.test_fc.out_features
    output = torch.zeros((frame_cnt, output_dim))
    cnt = 0
    with torch.no_grad():
        for frames in frames_gen:
            inp = frames.view(-1, length, frames.size(-2), frames.size(-1))
            rst, _ = net(inp, None, None, None, None)
            sc = rst.view(num_crop, -1, output_dim).mean(dim=0)
            output[cnt:cnt + sc.size(0), :] = sc
            cnt += sc.size(0)
    act, comp, reg = stpp_reorganized(output.numpy(), prop_ticks, prop_scaling, num_class + 1, num_class,
                                      num_class * 2, stpp_cfg=stpp_cfg, with_regression=net.with_regression)
    if reg is not None:
        reg = reg.reshape(-1, num_class, 2)
        if stats is not None:
            reg[:, :, 0] = reg[:, :, 0] * stats[1][0] + stats[0][0]
            reg[:, :, 1] = reg[:, :, 1] * stats[1][1] + stats[0][1]
    return act, comp, reg, output.numpy()


# --------------------------------------------------------------------------------------
# Input transforms after decoding / scaling -- /root/reference/transforms.py
# --------------------------------------------------------------------------------------

def oversample_transform(frames, crop_w, crop_h, mean, std, roll, is_flow):
    """frames: list of uint8 HxWxC (RGB) or HxW ('L', flow) arrays = the img_group after GroupScale.  Follows
    GroupOverSample (transforms.py:103-132), Stack(roll) (:256-268), ToTorchFormatTensor(div=False) (:271-288) and
    GroupNormalize (:67-80) with numpy slicing in place of the PIL calls (crop, FLIP_LEFT_RIGHT, ImageOps.invert)."""
    image_h, image_w = frames[0].shape[:2]
    w_step, h_step = (image_w - crop_w) // 4, (image_h - crop_h) // 4          # fill_fix_offset(False, ...), :184-193
    offsets = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
    group = []
    for o_w, o_h in offsets:
        normal, flip = [], []
        for i, img in enumerate(frames):
            crop = img[o_h:o_h + crop_h, o_w:o_w + crop_w]
            normal.append(crop)
            fc = crop[:, ::-1]
            flip.append(255 - fc if (is_flow and i % 2 == 0) else fc)
        group += normal + flip
    if is_flow:
        stacked = np.concatenate([np.expand_dims(x, 2) for x in group], axis=2)
    elif roll:
        stacked = np.concatenate([np.array(x)[:, :, ::-1] for x in group], axis=2)
    else:
        stacked = np.concatenate(group, axis=2)
    t = torch.from_numpy(np.ascontiguousarray(stacked)).permute(2, 0, 1).contiguous().float()
    rep_mean = list(mean) * (t.size(0) // len(mean))
    rep_std = list(std) * (t.size(0) // len(std))
    for ch, m, sd in zip(t, rep_mean, rep_std):
        ch.sub_(m).div_(sd)
    return t


# --------------------------------------------------------------------------------------
# Losses -- /root/reference/ops/ssn_ops.py:173-258, /root/reference/ssn_train.py:133,210-214
# --------------------------------------------------------------------------------------

def ohem_hinge(pred, labels, is_positive, ohem_ratio, group_size):
    """/root/reference/ops/ssn_ops.py:179-213.  Returns (loss scalar, grad wrt pred for d(loss)=1)."""
    pred_np = pred.detach().numpy()
    n = pred_np.shape[0]
    assert n == len(labels)
    losses = np.zeros(n, np.float32)
    slopes = np.zeros(n, np.float32)
    for i in range(n):
        v = np.float32(1) - np.float32(is_positive) * pred_np[i, int(labels[i]) - 1]
        losses[i] = max(np.float32(0), v)
        slopes[i] = -is_positive if losses[i] != 0 else 0
    grp = losses.reshape(-1, group_size)
    keep = int(group_size * ohem_ratio)
    order = np.argsort(-grp, axis=1, kind="stable")[:, :keep]
    total = np.float32(0)
    grad = np.zeros_like(pred_np)
    for g in range(grp.shape[0]):
        total = np.float32(total + np.float32(grp[g, order[g]].sum(dtype=np.float32)))
        for idx in order[g]:
            loc = idx + g * group_size
            grad[loc, int(labels[loc]) - 1] = slopes[loc]
    return total, grad


def completeness_loss(pred, labels, sample_split, sample_group_size, ohem_ratio=0.17):
    """/root/reference/ops/ssn_ops.py:223-239.  Returns (loss, dloss/dpred)."""
    c = pred.shape[1]
    p3 = pred.reshape(-1, sample_group_size, c)
    l2 = np.asarray(labels).reshape(-1, sample_group_size)
    pos = p3[:, :sample_split, :].reshape(-1, c)
    neg = p3[:, sample_split:, :].reshape(-1, c)
    pos_ls, pos_g = ohem_hinge(pos, l2[:, :sample_split].reshape(-1), 1, 1.0, sample_split)
    neg_ls, neg_g = ohem_hinge(neg, l2[:, sample_split:].reshape(-1), -1, ohem_ratio,
                               sample_group_size - sample_split)
    pos_cnt = pos.shape[0]
    neg_cnt = int(neg.shape[0] * ohem_ratio)
    den = float(pos_cnt + neg_cnt)
    loss = np.float32(pos_ls / den + neg_ls / den)
    g3 = np.zeros((p3.shape[0], sample_group_size, c), np.float32)
    g3[:, :sample_split, :] = pos_g.reshape(-1, sample_split, c) / den
    g3[:, sample_split:, :] = neg_g.reshape(-1, sample_group_size - sample_split, c) / den
    return loss, g3.reshape(-1, c)


def classwise_regression_loss(pred, labels, targets):
    """/root/reference/ops/ssn_ops.py:251-258: pick pred[i, label_i-1, :], SmoothL1(mean) * 2."""
    idx = labels.long() - 1
    rows = torch.arange(pred.shape[0])
    picked = pred[rows, idx, :]
    return F.smooth_l1_loss(picked.reshape(-1), targets.reshape(-1)) * 2


def activity_loss(logits, target):
    """/root/reference/ssn_train.py:133,210 (torch.nn.CrossEntropyLoss, mean)."""
    return F.cross_entropy(logits, target.long())


# --------------------------------------------------------------------------------------
# SSN wiring -- /root/reference/ssn_models.py
# --------------------------------------------------------------------------------------

class OracleSSN(nn.Module):
    """Restatement of the reference SSN for BNInception / InceptionV3 (ssn_models.py:10-300)."""

    def __init__(self, num_class, starting_segment=2, course_segment=5, ending_segment=2,
                 modality="RGB", new_length=None, dropout=0.8, no_regression=False,
                 test_mode=False, stpp_cfg=(1, (1, 2), 1), bn_mode="frozen", base_model="BNInception"):
        super().__init__()
        self.modality = modality
        self.starting_segment, self.course_segment, self.ending_segment = (
            starting_segment, course_segment, ending_segment)
        self.num_segments = starting_segment + course_segment + ending_segment
        self.new_length = (1 if modality == "RGB" else 5) if new_length is None else new_length
        self.dropout = dropout
        self.with_regression = not no_regression
        self.test_mode = test_mode
        self.stpp_cfg = stpp_cfg
        self.num_class = num_class
        # ssn_models.py:121-131 + :318-343 (flow: first conv gets 2*new_length input channels)
        # (RGBDiff, :345-376 with keep_rgb = False: 3 * new_length channels of frame differences)
        cin = 3 if modality == "RGB" else (3 if modality == "RGBDiff" else 2) * self.new_length
        if base_model == "InceptionV3":     # ssn_models.py:133-139
            self.base_model = OracleInceptionV3(in_channels=3)
            self.last_layer_name, first = "top_cls_fc", "conv_1a_3x3"
        else:
            self.base_model = OracleBNInception(in_channels=3)
            self.last_layer_name, first = "fc", "conv1_7x7_s2"
        feat = getattr(self.base_model, self.last_layer_name).in_features
        # ssn_models.py:69-74
        setattr(self.base_model, self.last_layer_name, nn.Identity() if dropout == 0 else nn.Dropout(p=dropout))
        if modality in ("Flow", "RGBDiff"):
            old = getattr(self.base_model, first)
            new = nn.Conv2d(cin, old.out_channels, old.kernel_size, old.stride, old.padding, bias=True)
            new.weight.data = old.weight.data.mean(dim=1, keepdim=True).expand(-1, cin, -1, -1).contiguous()
            new.bias.data = old.bias.data
            setattr(self.base_model, first, new)
        mult = sum(parse_stage_config(c)[1] for c in stpp_cfg)
        self.feat_multiplier = mult
        # ssn_models.py:76-91
        self.activity_fc = nn.Linear(feat, num_class + 1)
        self.completeness_fc = nn.Linear(feat * mult, num_class)
        self.regressor_fc = nn.Linear(feat * mult, 2 * num_class) if self.with_regression else None
        for fc in (self.activity_fc, self.completeness_fc, self.regressor_fc):
            if fc is not None:
                nn.init.normal_(fc.weight, 0, 0.001)
                nn.init.constant_(fc.bias, 0)
        self.test_fc = None
        # ssn_models.py:95-105
        self.freeze_count = {"partial": 2, "frozen": 1, "full": None}[bn_mode]

    def train(self, mode=True):
        """ssn_models.py:156-174."""
        super().train(mode)
        if self.freeze_count is None:
            return self
        count = 0
        for m in self.base_model.modules():
            if isinstance(m, nn.BatchNorm2d):
                count += 1
                if count >= self.freeze_count:
                    m.eval()
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
        return self

    def forward(self, input, aug_scaling=None, target=None, reg_target=None, prop_type=None):
        sample_len = (3 if self.modality == "RGB" else 2) * self.new_length
        if self.modality == "RGBDiff":
            # ssn_models.py:302-316 (_get_diff, keep_rgb = False): differences of the new_length + 1 stacked frames
            sample_len = 3 * self.new_length
            v = input.reshape((-1, self.num_segments, self.new_length + 1, 3) + tuple(input.shape[-2:]))
            input = v[:, :, 1:] - v[:, :, :-1]
        x = input.reshape((-1, sample_len) + tuple(input.shape[-2:]))
        base_out = getattr(self.base_model, self.last_layer_name)(self.base_model.features(x))
        if self.test_mode:
            # ssn_models.py:291-300
            return self.test_fc(base_out), base_out
        # ssn_models.py:259-289
        seg_split = [self.starting_segment, self.starting_segment + self.course_segment, self.num_segments]
        act_ft, comp_ft = stpp_forward(base_out, aug_scaling, seg_split, self.stpp_cfg, True)
        raw_act = self.activity_fc(act_ft)
        raw_comp = self.completeness_fc(comp_ft)
        t = prop_type.reshape(-1)
        act_idx = torch.nonzero((t == 0) | (t == 2)).reshape(-1)
        comp_idx = torch.nonzero((t == 0) | (t == 1)).reshape(-1)
        target = target.reshape(-1)
        if not self.with_regression:
            return raw_act[act_idx], target[act_idx], raw_comp[comp_idx], target[comp_idx]
        reg_idx = torch.nonzero(t == 0).reshape(-1)
        raw_reg = self.regressor_fc(comp_ft).reshape(-1, self.num_class, 2)
        reg_target = reg_target.reshape(-1, 2)
        return (raw_act[act_idx], target[act_idx], raw_comp[comp_idx], target[comp_idx],
                raw_reg[reg_idx], target[reg_idx], reg_target[reg_idx])

    def prepare_test_fc(self):
        """ssn_models.py:176-201: fold the three heads into one Linear over per-snippet features."""
        m = self.feat_multiplier
        d = self.activity_fc.in_features

        def fold(fc):
            w = fc.weight.data.reshape(fc.out_features, m, d).permute(1, 0, 2).reshape(-1, d)
            b = fc.bias.data.reshape(1, -1).repeat(m, 1).reshape(-1) / m
            return w, b

        ws, bs = [self.activity_fc.weight.data], [self.activity_fc.bias.data]
        for fc in (self.completeness_fc, self.regressor_fc):
            if fc is not None:
                w, b = fold(fc)
                ws.append(w)
                bs.append(b)
        w, b = torch.cat(ws), torch.cat(bs)
        self.test_fc = nn.Linear(d, w.shape[0])
        self.test_fc.weight.data = w
        self.test_fc.bias.data = b


def detections_for_video(rel_prop, act_scores, comp_scores, reg_scores, num_class, nms_threshold, top_k=0,
                         no_regression=False, video_cls_score=None, cls_top_k=1, softmax_bf=False):
    """The per-video part of /root/reference/eval_detection_results.py: gen_detection_results (:91-144: the two
    branches without external class scores, and -- video_cls_score given -- the `--cls_scores` branch :130-144),
    temporal_nms (ops/utils.py:56-82) and perform_regression (:167-178).
    -> {cls: float64 array [n, 5] = (start, end, score, loc, dur)}

    The reference calls ``np.argsort`` with numpy's default (unstable) sort, so WHICH of several exactly equal scores
    it keeps / ranks first is an accident of the numpy build.  This restatement -- and the product -- pin the choice a
    stable sort makes (``kind="stable"``): of scores tied at the k-th place the higher flat indices survive the top-k,
    and inside a class tied scores are visited higher proposal index first.  Pinned against the reference's own
    functions by tests/golden/ref_detection.npz (cases without exact ties, where the sort kind cannot matter)."""
    def softmax(scores):                                           # ops/utils.py:35-37
        es = np.exp(scores - scores.max(axis=-1)[..., None])
        return es / es.sum(axis=-1)[..., None]

    def temporal_nms(bboxes, thresh):                              # ops/utils.py:56-82
        t1, t2, scores = bboxes[:, 0], bboxes[:, 1], bboxes[:, 2]
        durations = t2 - t1
        order = scores.argsort(kind="stable")[::-1]
        keep = []
        while order.size > 0:
            i = order[0]
            keep.append(i)
            tt1 = np.maximum(t1[i], t1[order[1:]])
            tt2 = np.minimum(t2[i], t2[order[1:]])
            intersection = tt2 - tt1
            iou = intersection / (durations[i] + durations[order[1:]] - intersection).astype(float)
            inds = np.where(iou <= thresh)[0]
            order = order[inds + 1]
        return bboxes[keep, :]

    rel_prop = np.asarray(rel_prop)
    rel_prop = np.squeeze(rel_prop, 0) if rel_prop.ndim == 3 else rel_prop
    act_scores, comp_scores = np.asarray(act_scores), np.asarray(comp_scores)
    if reg_scores is None:
        reg_scores = np.zeros((len(rel_prop), num_class, 2), dtype=np.float32)
    reg_scores = np.asarray(reg_scores).reshape((-1, num_class, 2))
    dets = {}
    if video_cls_score is not None:       # :130-144 (top_k plays no role here; --softmax_before_filter = softmax_bf)
        combined = (softmax(act_scores)[:, 1:] if softmax_bf else act_scores[:, 1:]) * np.exp(comp_scores)
        for video_cls in np.argsort(np.asarray(video_cls_score), kind="stable")[-cls_top_k:]:
            dets[int(video_cls)] = np.concatenate((rel_prop, combined[:, video_cls][:, None], reg_scores[:, video_cls, 0][:, None],
                                                   reg_scores[:, video_cls, 1][:, None]), axis=1)
    elif top_k <= 0:
        combined = softmax(act_scores)[:, 1:] * np.exp(comp_scores)
        for i in range(num_class):
            dets[i] = np.concatenate((rel_prop, combined[:, i][:, None], reg_scores[:, i, 0][:, None],
                                      reg_scores[:, i, 1][:, None]), axis=1)
    else:
        combined = softmax(act_scores[:, 1:]) * np.exp(comp_scores)
        for k in np.argsort(combined.ravel(), kind="stable")[-top_k:]:
            cls, prop_idx = k % num_class, k // num_class
            row = [rel_prop[prop_idx, 0], rel_prop[prop_idx, 1], combined[prop_idx, cls],
                   reg_scores[prop_idx, cls, 0], reg_scores[prop_idx, cls, 1]]
            dets[cls] = np.array([row]) if cls not in dets else np.vstack([dets[cls], row])
    dets = {c: temporal_nms(v, nms_threshold) for c, v in dets.items()}
    if not no_regression:
        for c, d in dets.items():
            center, duration = (d[:, 0] + d[:, 1]) / 2, d[:, 1] - d[:, 0]
            new_center = center + duration * d[:, 3]
            new_duration = duration * np.exp(d[:, 4])
            dets[c] = np.concatenate((np.clip(new_center - new_duration / 2, 0, 1)[:, None],
                                      np.clip(new_center + new_duration / 2, 0, 1)[:, None], d[:, 2:]), axis=1)
    return {c: d for c, d in dets.items() if len(d)}, combined


class OracleBinaryClassifier(nn.Module):
    """Restatement of the reference's actionness classifier, /root/reference/binary_model.py:7-254, for the
    BNInception backbone: backbone -> (Dropout) -> mean over the course segments (:229-231) -> classifier_fc (:232);
    test mode: test_fc on single frames (:237-240, :245-254)."""

    def __init__(self, num_class, course_segment, modality="RGB", new_length=None, dropout=0.8, test_mode=False):
        super().__init__()
        self.modality, self.course_segment, self.test_mode = modality, course_segment, test_mode
        self.new_length = (1 if modality == "RGB" else 5) if new_length is None else new_length
        # (RGBDiff, :345-376 with keep_rgb = False: 3 * new_length channels of frame differences)
        cin = 3 if modality == "RGB" else (3 if modality == "RGBDiff" else 2) * self.new_length
        self.base_model = OracleBNInception(in_channels=3)
        feat = self.base_model.fc.in_features
        self.base_model.fc = nn.Identity() if dropout == 0 else nn.Dropout(p=dropout)      # :121-125
        if modality == "Flow":                                                             # :54-79
            old = self.base_model.conv1_7x7_s2
            new = nn.Conv2d(cin, 64, 7, 2, 3, bias=True)
            new.weight.data = old.weight.data.mean(dim=1, keepdim=True).expand(-1, cin, -1, -1).contiguous()
            new.bias.data = old.bias.data
            self.base_model.conv1_7x7_s2 = new
        self.classifier_fc = nn.Linear(feat, num_class)                                     # :127-130
        nn.init.normal_(self.classifier_fc.weight, 0, 0.001)
        nn.init.constant_(self.classifier_fc.bias, 0)
        self.test_fc = None

    def train(self, mode=True):
        """:203-216 (bn_mode 'frozen': every BatchNorm2d in eval mode)."""
        super().train(mode)
        for m in self.base_model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
                m.weight.requires_grad = False
                m.bias.requires_grad = False
        return self

    def prepare_test_fc(self):
        self.test_fc = nn.Linear(self.classifier_fc.in_features, self.classifier_fc.out_features)
        self.test_fc.weight.data = self.classifier_fc.weight.data
        self.test_fc.bias.data = self.classifier_fc.bias.data

    def forward(self, inputdata, target=None):
        sample_len = (3 if self.modality == "RGB" else 2) * self.new_length
        x = inputdata.reshape((-1, sample_len) + tuple(inputdata.shape[-2:]))
        base_out = self.base_model.fc(self.base_model.features(x))
        if self.test_mode:
            return self.test_fc(base_out), base_out
        src = base_out.view(-1, self.course_segment, base_out.size(1))
        return self.classifier_fc(src.mean(dim=1)), target.view(-1)


def ssn_total_loss(outputs, num_videos, comp_weight=0.1, reg_weight=0.1, ohem_ratio=0.17,
                   fg_per_video=1, group_size=7):
    """Loss mix of /root/reference/ssn_train.py:210-214 with differentiable torch pieces.

    The completeness term uses the numpy OHEM restatement for the value and injects its
    analytic gradient through a surrogate (sum(pred * g)), so .backward() reproduces the
    reference's autograd.Function backward (ops/ssn_ops.py:203-213).
    """
    act, act_t, comp, comp_t, reg, reg_l, reg_t = outputs
    act_loss = activity_loss(act, act_t)
    c_val, c_grad = completeness_loss(comp.detach(), comp_t.numpy(), fg_per_video, group_size, ohem_ratio)
    surrogate = (comp * torch.from_numpy(c_grad)).sum()
    comp_loss = surrogate - surrogate.detach() + torch.tensor(float(c_val))
    reg_loss = classwise_regression_loss(reg, reg_l, reg_t)
    total = act_loss + comp_weight * comp_loss + reg_weight * reg_loss
    return total, act_loss, comp_loss, reg_loss
