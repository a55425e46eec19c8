"""Mirror of the reference's ``ssn_models.SSN`` (/root/reference/ssn_models.py:10-395) on the
MI355X-native kernels.

Same constructor, ``forward`` 7-tuple / test 2-tuple, ``prepare_test_fc``,
``get_optim_policies``, BN-freezing ``train()`` and ``state_dict`` keys as the reference, so it
drops into the loops of ssn_train.py:205-253 and ssn_test.py:78-92.  Differences that are
deliberate and documented in DESIGN.md:
  * BNInception and InceptionV3 are built (training and testing, one executor); resnet/vgg raise;
  * the backbone is this repo's ``bninception.BNInception`` executor instead of ``model_zoo``;
  * ``bn_mode`` 'partial' / 'full' run the training-mode BatchNorm kernels of csrc/bn_train.hip (batch statistics per
    rank, like the per-replica statistics of the reference's DataParallel);
  * one host sync per forward (prop_type -> row indices) instead of the reference's three
    ``nonzero()`` syncs.
"""
import os

import numpy as np
import torch
from torch import nn

from . import functional as FN
from .bninception import BNInception
from .inceptionv3 import InceptionV3
from .ops.ssn_ops import Identity, StructuredTemporalPyramidPooling

_dropout_calls = [0]


class HipLinear(nn.Linear):
    """nn.Linear whose forward/backward are the ssn_linear_* kernels (state_dict-compatible)."""

    def forward(self, input):
        lead = input.shape[:-1]
        out = FN.LinearFn.apply(input.reshape(-1, input.shape[-1]), self.weight, self.bias)
        return out.reshape(lead + (self.out_features,))


class HipDropout(nn.Dropout):
    """nn.Dropout on the ssn_dropout_* kernels.

    Philox stream keyed by (seed drawn once from torch's CPU RNG) + a per-call counter that lives in
    device memory and is advanced by the kernel launch itself, so a hipGraph replay of a captured step
    still draws a fresh mask every time.
    """

    def forward(self, input):
        if not self.training or self.p == 0:
            return input
        st = getattr(self, "_rng_state", None)
        if st is None or st[1].device != input.device:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            st = (seed, torch.zeros(1, dtype=torch.int64, device=input.device))
            self._rng_state = st
        return FN.DropoutFn.apply(input, float(self.p), st[0], st[1])


class SSN(torch.nn.Module):
    def __init__(self, num_class,
                 starting_segment, course_segment, ending_segment, modality,
                 base_model='BNInception', new_length=None,
                 dropout=0.8,
                 crop_num=1, no_regression=False, test_mode=False,
                 stpp_cfg=(1, (1, 2), 1), bn_mode='frozen', verbose=False):
        super(SSN, self).__init__()
        self.modality = modality
        self.num_segments = starting_segment + course_segment + ending_segment
        self.starting_segment = starting_segment
        self.course_segment = course_segment
        self.ending_segment = ending_segment
        self.reshape = True
        self.dropout = dropout
        self.crop_num = crop_num
        self.with_regression = not no_regression
        self.test_mode = test_mode
        self.bn_mode = bn_mode

        if new_length is None:
            self.new_length = 1 if modality == "RGB" else 5
        else:
            self.new_length = new_length

        if verbose:
            print("Initializing SSN with base model: {} ({} / {}+{}+{} segments / dropout {} / regression {} / "
                  "bn_mode {} / stpp {})".format(base_model, modality, starting_segment, course_segment,
                                                 ending_segment, dropout, self.with_regression, bn_mode, stpp_cfg))

        self._prepare_base_model(base_model)
        self._prepare_ssn(num_class, stpp_cfg)

        if self.modality == 'Flow':
            self.base_model = self._construct_flow_model(self.base_model)
        elif self.modality == 'RGBDiff':
            self.base_model = self._construct_diff_model(self.base_model)

        self.prepare_bn()

    # ---- /root/reference/ssn_models.py:69-93
    def _prepare_ssn(self, num_class, stpp_cfg):
        feature_dim = getattr(self.base_model, self.base_model.last_layer_name).in_features
        if self.dropout == 0:
            setattr(self.base_model, self.base_model.last_layer_name, Identity())
        else:
            setattr(self.base_model, self.base_model.last_layer_name, HipDropout(p=self.dropout))

        self.stpp = StructuredTemporalPyramidPooling(feature_dim, True, configs=stpp_cfg)
        self.activity_fc = HipLinear(self.stpp.activity_feat_dim(), num_class + 1)
        self.completeness_fc = HipLinear(self.stpp.completeness_feat_dim(), num_class)

        nn.init.normal_(self.activity_fc.weight.data, 0, 0.001)
        nn.init.constant_(self.activity_fc.bias.data, 0)
        nn.init.normal_(self.completeness_fc.weight.data, 0, 0.001)
        nn.init.constant_(self.completeness_fc.bias.data, 0)

        self.test_fc = None
        if self.with_regression:
            self.regressor_fc = HipLinear(self.stpp.completeness_feat_dim(), 2 * num_class)
            nn.init.normal_(self.regressor_fc.weight.data, 0, 0.001)
            nn.init.constant_(self.regressor_fc.bias.data, 0)
        else:
            self.regressor_fc = None

        return feature_dim

    # ---- /root/reference/ssn_models.py:95-105
    def prepare_bn(self):
        if self.bn_mode == 'partial':
            self.freeze_count = 2
        elif self.bn_mode == 'frozen':
            self.freeze_count = 1
        elif self.bn_mode == 'full':
            self.freeze_count = None
        else:
            raise ValueError("unknown bn mode")

    # ---- /root/reference/ssn_models.py:107-154
    _BACKBONES = {   # name -> (constructor, last layer, input size): the ones built on the MI355X kernels
        'BNInception': (BNInception, 'fc', 224),            # training and testing
        'InceptionV3': (InceptionV3, 'top_cls_fc', 299),    # training and testing (BASELINE.json configs[4] tests on it)
    }

    def _prepare_base_model(self, base_model):
        if base_model in SSN._BACKBONES:      # (BinaryClassifier borrows this method)
            ctor, last, size = SSN._BACKBONES[base_model]
            self.base_model = ctor()
            self.base_model.last_layer_name = last
            self.input_size = size
            # Caffe-heritage statistics: BGR means on the 0..255 scale, unit std; flow is centred at 128
            self.input_std = [1]
            if self.modality == 'Flow':
                self.input_mean = [128]
            elif self.modality == 'RGBDiff':
                self.input_mean = [104, 117, 128] * (1 + self.new_length)
            else:
                self.input_mean = [104, 117, 128]
        elif 'resnet' in base_model or 'vgg' in base_model or 'inception' in base_model:
            raise NotImplementedError(
                "base model {} is not built: the MI355X hot path covers BNInception and InceptionV3 "
                "(training and testing)".format(base_model))
        else:
            raise ValueError('Unknown base model: {}'.format(base_model))

    # ---- /root/reference/ssn_models.py:156-174
    def train(self, mode=True):
        super(SSN, self).train(mode)
        count = 0
        if self.freeze_count is None:
            return self
        for m in self.base_model.modules():
            if isinstance(m, nn.BatchNorm2d):
                count += 1
                if count >= self.freeze_count:
                    m.eval()
                    # shutdown update in frozen mode
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
        return self

    # ---- range guard of the backbone's delayed scales (no counterpart in the reference, whose cuDNN path stores fp32): eager
    # calls repair themselves (planes_exec.py); a training loop that replays a captured step polls these
    def scale_fault_flag(self, device=None):
        """Device int32 word for ``SSNSGD.step(skip_flag=)``: non-zero = a pass of the backbone left the range of its scales."""
        dev = next(self.base_model.parameters()).device if device is None else device
        return self.base_model.planes_flag(dev)[0:1]

    def scale_fault(self):
        return self.base_model.scale_fault()

    def recalibrate_scales(self):
        self.base_model.recalibrate()

    # ---- /root/reference/ssn_models.py:176-201
    def prepare_test_fc(self):
        m = self.stpp.feat_multiplier
        d = self.activity_fc.in_features
        n_out = (self.activity_fc.out_features + self.completeness_fc.out_features * m
                 + (self.regressor_fc.out_features * m if self.with_regression else 0))
        self.test_fc = HipLinear(d, n_out)
        dev = self.activity_fc.weight.device

        def reorganise(fc):
            w = fc.weight.data.view(fc.out_features, m, d).transpose(0, 1).contiguous().view(-1, d)
            b = fc.bias.data.view(1, -1).expand(m, fc.out_features).contiguous().view(-1) / m
            return w, b

        weights, biases = [self.activity_fc.weight.data], [self.activity_fc.bias.data]
        weights_b = reorganise(self.completeness_fc)
        weights.append(weights_b[0])
        biases.append(weights_b[1])
        if self.with_regression:
            wr, br = reorganise(self.regressor_fc)
            weights.append(wr)
            biases.append(br)
        self.test_fc.weight.data = torch.cat(weights).to(dev)
        self.test_fc.bias.data = torch.cat(biases).to(dev)

    # ---- /root/reference/ssn_models.py:203-251
    def get_optim_policies(self):
        """Five parameter groups with the reference's multipliers: the first convolution (weight / bias), every other
        convolution and linear layer (weight / bias), BatchNorm1d parameters; BatchNorm2d is frozen and gets none."""
        groups = {"first_conv_weight": [], "first_conv_bias": [], "normal_weight": [], "normal_bias": [],
                  "BN scale/shift": []}
        seen_conv = False
        for m in self.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv1d, torch.nn.Linear)):
                first = isinstance(m, (torch.nn.Conv2d, torch.nn.Conv1d)) and not seen_conv
                seen_conv = seen_conv or isinstance(m, (torch.nn.Conv2d, torch.nn.Conv1d))
                ps = list(m.parameters())
                groups["first_conv_weight" if first else "normal_weight"].append(ps[0])
                if len(ps) == 2:
                    groups["first_conv_bias" if first else "normal_bias"].append(ps[1])
            elif isinstance(m, torch.nn.BatchNorm1d):
                groups["BN scale/shift"].extend(m.parameters())
            elif isinstance(m, torch.nn.BatchNorm2d):
                continue
            elif len(m._modules) == 0 and len(list(m.parameters())) > 0:
                raise ValueError("New atomic module type: {}. Need to give it a learning policy".format(type(m)))
        mult = {"first_conv_weight": (1, 1), "first_conv_bias": (2, 0), "normal_weight": (1, 1), "normal_bias": (2, 0),
                "BN scale/shift": (1, 0)}
        return [{'params': ps, 'lr_mult': mult[name][0], 'decay_mult': mult[name][1], 'name': name}
                for name, ps in groups.items()]

    # ---- /root/reference/ssn_models.py:253-300
    def forward(self, input, aug_scaling=None, target=None, reg_target=None, prop_type=None):
        if not self.test_mode:
            return self.train_forward(input, aug_scaling, target, reg_target, prop_type)
        else:
            return self.test_forward(input)

    def _backbone(self, input):
        sample_len = (3 if self.modality == "RGB" else 2) * self.new_length
        if self.modality == 'RGBDiff':
            sample_len = 3 * self.new_length
            input = self._get_diff(input)
        x = input.reshape((-1, sample_len) + tuple(input.shape[-2:]))
        feat = self.base_model.features(x)
        return getattr(self.base_model, self.base_model.last_layer_name)(feat)

    # STPP + the three heads + the row selection as one launch each way (functional.HeadsFn); False: the separate kernels
    fused_heads = os.environ.get("SSN_FUSED_HEADS", "1") != "0"

    def _fused_train_heads(self, base_out, aug_scaling, target, reg_target, prop_type):
        seg_split = (self.starting_segment, self.starting_segment + self.course_segment, self.num_segments)
        table = self.stpp.table_for(seg_split)
        dev = base_out.device
        idx = self._row_indexers(prop_type, dev)
        pos = self._indexer_cache[4]
        reg = self.regressor_fc if self.with_regression else None
        outs = FN.HeadsFn.apply(base_out, aug_scaling, table, seg_split[2], idx if reg is not None else idx[:2] + (None,),
                                pos if reg is not None else pos[:2] + (None,),
                                self.activity_fc.weight, self.activity_fc.bias, self.completeness_fc.weight, self.completeness_fc.bias,
                                None if reg is None else reg.weight, None if reg is None else reg.bias)

        labels = self._select_labels(target, reg_target if reg is not None else None, idx, dev)
        if reg is not None:
            return (outs[0], labels[0], outs[1], labels[1], outs[2].reshape(-1, self.completeness_fc.out_features, 2), labels[2], labels[3])
        return (outs[0], labels[0], outs[1], labels[1])

    @staticmethod
    def _select_labels(target, reg_target, idx, dev):
        """target[act rows], target[comp rows], target[reg rows], reg_target[reg rows] (ssn_models.py:275-289): integer / target
        bookkeeping, one launch (ssn_label_select) instead of four index_select."""
        from . import kernels as K
        target = target.reshape(-1).to(dev).long().contiguous()
        has = reg_target is not None
        outs = [torch.empty(idx[k].numel(), device=dev, dtype=torch.int64) for k in range(3 if has else 2)] + ([] if has else [None])
        out_reg = None
        if has:
            reg_target = reg_target.reshape(-1, 2).to(dev).float().contiguous()
            out_reg = torch.empty((idx[2].numel(), 2), device=dev, dtype=torch.float32)
        K.label_select(target, reg_target, (idx[0], idx[1], idx[2] if has else None), outs, out_reg)
        return outs[0], outs[1], outs[2], out_reg

    def train_forward(self, input, aug_scaling, target, reg_target, prop_type):
        base_out = self._backbone(input)
        if (self.fused_heads and self.stpp.sc and (1 + self.stpp.feat_multiplier) * self.stpp.feat_dim * 4 <= 65536
                and self.num_segments <= 16
                and type(self.activity_fc) is HipLinear and type(self.completeness_fc) is HipLinear):
            return self._fused_train_heads(base_out, aug_scaling, target, reg_target, prop_type)
        activity_ft, completeness_ft = self.stpp(base_out, aug_scaling,
                                                 [self.starting_segment,
                                                  self.starting_segment + self.course_segment,
                                                  self.num_segments])
        raw_act_fc = self.activity_fc(activity_ft)
        raw_comp_fc = self.completeness_fc(completeness_ft)

        # the reference's three nonzero() calls (ssn_models.py:275-282) -> one host read of prop_type
        dev = raw_act_fc.device
        act_indexer, comp_indexer, reg_indexer = self._row_indexers(prop_type, dev)
        idx = (act_indexer, comp_indexer, reg_indexer)
        labels = self._select_labels(target, reg_target if self.with_regression else None, idx, dev)
        if self.with_regression:
            raw_regress_fc = self.regressor_fc(completeness_ft).reshape(-1, self.completeness_fc.out_features, 2)
            return (FN.RowGatherFn.apply(raw_act_fc, act_indexer), labels[0],
                    FN.RowGatherFn.apply(raw_comp_fc, comp_indexer), labels[1],
                    FN.RowGatherFn.apply(raw_regress_fc, reg_indexer), labels[2], labels[3])
        else:
            return (FN.RowGatherFn.apply(raw_act_fc, act_indexer), labels[0],
                    FN.RowGatherFn.apply(raw_comp_fc, comp_indexer), labels[1])

    def _row_indexers(self, prop_type, dev):
        """Rows of the activity / completeness / regression heads by proposal type (ssn_models.py:275-282).  The
        reference recomputes ``nonzero()`` on every forward (three device syncs); here ``prop_type`` is read to the
        host once per forward (no sync at all when it is a CPU tensor, as the DataLoader delivers it) and the device
        index tensors are reused only while its CONTENTS are unchanged.  During hipGraph capture nothing can be read
        back: the pattern of the last eager forward on the same tensor is used, and the graph is only valid for it."""
        cache = getattr(self, "_indexer_cache", None)
        capturing = prop_type.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing:
            # (`_version` counts in-place writes: a prop_type buffer modified since the eager forward that read it would
            # silently bake stale row indices -- and gather sizes -- into the graph)
            key = (prop_type.data_ptr(), tuple(prop_type.shape), str(dev))
            if cache is None or cache[0] != key or cache[3] != prop_type._version:
                raise RuntimeError("hipGraph capture of SSN.forward needs one eager forward with the same, unmodified "
                                   "prop_type tensor first (its proposal-type pattern is baked into the captured graph; "
                                   "re-capture whenever the pattern changes)")
            return cache[2]
        type_host = prop_type.detach().reshape(-1).cpu()
        if cache is not None and cache[0][2] == str(dev) and torch.equal(cache[1], type_host):
            self._indexer_cache = ((prop_type.data_ptr(), tuple(prop_type.shape), str(dev)), cache[1], cache[2], prop_type._version,
                                   cache[4])
            return cache[2]
        sets = ((type_host == 0) | (type_host == 2), (type_host == 0) | (type_host == 1), type_host == 0)
        idx = tuple(torch.nonzero(m_).reshape(-1).to(dev) for m_ in sets)
        # the inverse maps the fused head kernels use: row of proposal p in each gathered output, -1 = not selected
        pos = tuple((torch.cumsum(m_.to(torch.int32), 0, dtype=torch.int32) - 1).masked_fill(~m_, -1).to(dev) for m_ in sets)
        self._indexer_cache = ((prop_type.data_ptr(), tuple(prop_type.shape), str(dev)), type_host.clone(), idx, prop_type._version,
                               pos)
        return idx

    def test_forward(self, input):
        base_out = self._backbone(input)
        return self.test_fc(base_out), base_out

    # ---- /root/reference/ssn_models.py:302-316
    def _get_diff(self, input, keep_rgb=False):
        """RGBDiff: every segment holds new_length + 1 stacked RGB frames; the backbone sees the new_length differences
        of consecutive frames (one launch, csrc/frames.hip: ssn_frame_diff)."""
        if keep_rgb:
            raise NotImplementedError("keep_rgb=True is never used by the reference drivers")
        from . import kernels as K
        return K.frame_diff(input.contiguous().float(), self.new_length, 3)

    # ---- /root/reference/ssn_models.py:345-376 (keep_rgb = False)
    def _construct_diff_model(self, base_model, keep_rgb=False):
        """First-conv surgery for RGB differences: as for flow, over 3 * new_length input channels.  (The reference's
        own method subscripts a ``filter`` object -- a Python-2 idiom -- and only runs with ``filter`` shimmed to return
        a list, which is how oracle/make_golden.py obtains the fixture this is tested against.)"""
        if keep_rgb:
            raise NotImplementedError("keep_rgb=True is never used by the reference drivers")
        return SSN._first_conv_surgery(base_model, 3 * self.new_length)

    # ---- /root/reference/ssn_models.py:318-343
    def _construct_flow_model(self, base_model):
        """First-conv surgery for stacked optical flow: the RGB kernel averaged over its input channels and repeated
        over the 2 * new_length flow channels, bias kept; the new layer replaces the old one under the same name."""
        return SSN._first_conv_surgery(base_model, 2 * self.new_length)

    @staticmethod
    def _first_conv_surgery(base_model, c_flow):
        name, old = next((n, m) for n, m in base_model.named_modules() if isinstance(m, nn.Conv2d))
        holder = base_model
        *path, leaf = name.split(".")
        for part in path:
            holder = getattr(holder, part)
        has_bias = old.bias is not None
        new = nn.Conv2d(c_flow, old.out_channels, old.kernel_size, old.stride, old.padding, bias=has_bias)
        with torch.no_grad():
            new.weight.copy_(old.weight.mean(dim=1, keepdim=True).expand(-1, c_flow, -1, -1))
            if has_bias:
                new.bias.copy_(old.bias)
        setattr(holder, leaf, new)
        return base_model

    # ---- /root/reference/ssn_models.py:378-395
    @property
    def crop_size(self):
        return self.input_size

    @property
    def scale_size(self):
        return self.input_size * 256 // 224

    def get_augmentation(self):
        """The training augmentation the driver composes in front of Stack / ToTorchFormatTensor / GroupNormalize
        (ssn_train.py:65, 106-111): scale-jittered fixed-position crop resized to the input size, then a random
        horizontal flip -- PIL images on the loader workers, like the reference (`transforms` is this package's own
        implementation, RNG-compatible with the reference's; see action_detection_amd/transforms.py)."""
        from . import transforms as T
        if self.modality == 'RGB':
            return T.Compose([T.GroupMultiScaleCrop(self.input_size, [1, .875, .75, .66]),
                              T.GroupRandomHorizontalFlip(is_flow=False)])
        if self.modality == 'Flow':
            return T.Compose([T.GroupMultiScaleCrop(self.input_size, [1, .875, .75]),
                              T.GroupRandomHorizontalFlip(is_flow=True)])
        if self.modality == 'RGBDiff':
            return T.Compose([T.GroupMultiScaleCrop(self.input_size, [1, .875, .75]),
                              T.GroupRandomHorizontalFlip(is_flow=False)])
        raise ValueError("unknown modality {}".format(self.modality))
