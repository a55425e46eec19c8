"""Mirror of the reference's ``ops/ssn_ops.py`` (same class names, argument meaning and error
behaviour) whose arithmetic runs in the gfx950 kernels of include/ssn_hip.h.

Reference: /root/reference/ops/ssn_ops.py -- Identity :8-10, parse_stage_config :13-19,
StructuredTemporalPyramidPooling :22-79, STPPReorgainzed :82-170, OHEMHingeLoss :173-213,
CompletenessLoss :216-239, ClassWiseRegressionLoss :242-258.
(The reference's spellings ``standalong_classifier`` / ``STPPReorgainzed`` are kept: they are
part of the API its drivers use.)
"""
import numpy as np
import torch
from torch import nn

from .. import functional as FN
from .. import kernels as K


class Identity(torch.nn.Module):
    def forward(self, input):
        return input


def parse_stage_config(stage_cfg):
    if isinstance(stage_cfg, int):
        return (stage_cfg,), stage_cfg
    elif isinstance(stage_cfg, tuple) or isinstance(stage_cfg, list):
        return stage_cfg, sum(stage_cfg)
    else:
        raise ValueError("Incorrect STPP config {}".format(stage_cfg))


def _stage_ticks(stage_len, n_part):
    # The reference's own expression (ops/ssn_ops.py:53-55) evaluated once on the host, so the
    # integer segment assignment is identical by construction.
    ticks = torch.arange(0, stage_len + 1e-5, stage_len / n_part)
    return [int(t) for t in ticks]


class StructuredTemporalPyramidPooling(torch.nn.Module):
    """STPP operator for training (one HIP launch forward, one backward)."""

    def __init__(self, feat_dim, standalong_classifier=False, configs=(1, (1, 2), 1)):
        super(StructuredTemporalPyramidPooling, self).__init__()
        self.sc = standalong_classifier
        self.feat_dim = feat_dim

        starting_parts, starting_mult = parse_stage_config(configs[0])
        course_parts, course_mult = parse_stage_config(configs[1])
        ending_parts, ending_mult = parse_stage_config(configs[2])

        self.feat_multiplier = starting_mult + course_mult + ending_mult
        self.parts = (starting_parts, course_parts, ending_parts)
        self.norm_num = (starting_mult, course_mult, ending_mult)
        self._tables = {}

    def part_table(self, seg_split):
        """[(seg_lo, seg_hi, norm, scale_col)] in output order for ``seg_split = [x1, x2, n_seg]``."""
        x1, x2, n_seg = seg_split
        bounds = ((0, x1, 0), (x1, x2, -1), (x2, n_seg, 1))
        rows = []
        for (lo, hi, col), parts, norm in zip(bounds, self.parts, self.norm_num):
            for n_part in parts:
                ticks = _stage_ticks(hi - lo, n_part)
                for i in range(n_part):
                    rows.append((lo + ticks[i], lo + ticks[i + 1], norm, col))
        return rows

    def table_for(self, seg_split):
        key = tuple(int(v) for v in seg_split)
        if key not in self._tables:
            self._tables[key] = K.make_stpp_table(self.part_table(key), key[2], key[0], key[1])
        return self._tables[key]

    def forward(self, ft, scaling, seg_split):
        key = tuple(int(v) for v in seg_split)
        act_ft, stpp_ft = FN.StppFn.apply(ft, scaling, self.table_for(key), key[2])
        if not self.sc:
            return stpp_ft, stpp_ft
        return act_ft, stpp_ft

    def activity_feat_dim(self):
        if self.sc:
            return self.feat_dim
        else:
            return self.feat_dim * self.feat_multiplier

    def completeness_feat_dim(self):
        return self.feat_dim * self.feat_multiplier


class STPPReorgainzed:
    """Re-organised dense-test pooling (/root/reference/ops/ssn_ops.py:82-170) on the GPU.

    The per-proposal row ranges are computed on the host with the reference's own float
    arithmetic (np.arange + int(), :137-147) -- a few hundred integers per video -- and the
    pooling itself is one launch with one workgroup per proposal.
    """

    def __init__(self, feat_dim, act_score_len, comp_score_len, reg_score_len,
                 standalong_classifier=False, with_regression=True, stpp_cfg=(1, 1, 1)):
        self.sc = standalong_classifier
        self.act_len = act_score_len
        self.comp_len = comp_score_len
        self.reg_len = reg_score_len
        self.with_regression = with_regression
        self.feat_dim = feat_dim

        starting_parts, starting_mult = parse_stage_config(stpp_cfg[0])
        course_parts, course_mult = parse_stage_config(stpp_cfg[1])
        ending_parts, ending_mult = parse_stage_config(stpp_cfg[2])

        feature_multiplie = starting_mult + course_mult + ending_mult
        self.feat_multiplier = feature_multiplie
        self.stpp_cfg = (starting_parts, course_parts, ending_parts)

        self.act_slice = slice(0, self.act_len if self.sc else (self.act_len * feature_multiplie))
        self.comp_slice = slice(self.act_slice.stop, self.act_slice.stop + self.comp_len * feature_multiplie)
        self.reg_slice = slice(self.comp_slice.stop, self.comp_slice.stop + self.reg_len * feature_multiplie)
        cols = []
        for stage_idx, stage_cfg in enumerate(self.stpp_cfg):
            col = 0 if stage_idx == 0 else (1 if stage_idx == len(self.stpp_cfg) - 1 else -1)
            cols.extend([col] * sum(stage_cfg))
        self._part_cols = cols

    def host_ranges(self, proposal_ticks, n_rows):
        """int32 [P, n_parts, 2] row ranges (pr<=pl = skipped) and [P, 2] activity ranges."""
        ticks_all = np.asarray(proposal_ticks).astype(np.int64)
        n_out = ticks_all.shape[0]
        n_parts = self.feat_multiplier
        ranges = np.zeros((n_out, n_parts, 2), np.int32)
        act = np.zeros((n_out, 2), np.int32)
        for i in range(n_out):
            ticks = ticks_all[i]
            act[i, 0] = ticks[1]
            act[i, 1] = max(ticks[1] + 1, ticks[2])
            offset = 0
            for stage_idx, stage_cfg in enumerate(self.stpp_cfg):
                stage_cnt = sum(stage_cfg)
                left = ticks[stage_idx]
                right = max(ticks[stage_idx] + 1, ticks[stage_idx + 1])
                if right <= 0 or left >= n_rows:
                    offset += stage_cnt
                    continue
                for n_part in stage_cfg:
                    part_ticks = np.arange(left, right + 1e-5, (right - left) / n_part)
                    for j in range(n_part):
                        pl = int(part_ticks[j])
                        pr = int(part_ticks[j + 1])
                        if pr - pl >= 1:
                            ranges[i, offset] = (pl, pr)
                        offset += 1
        return ranges, act

    def forward(self, scores, proposal_ticks, scaling):
        assert scores.size(1) == self.feat_dim
        dev = scores.device
        n_out = proposal_ticks.size(0) if torch.is_tensor(proposal_ticks) else len(proposal_ticks)
        pt = proposal_ticks.cpu().numpy() if torch.is_tensor(proposal_ticks) else np.asarray(proposal_ticks)
        ranges, act = self.host_ranges(pt, scores.size(0))
        # The reference slices with Python semantics (ops/ssn_ops.py:149,157): a range END past the last score row is
        # clamped to it (the mean then runs over the rows that exist); a range that STARTS at or past the last row is a
        # mean over nothing -- NaN, which the reference adds into the proposal's scores (marked (-1, -1) for the
        # kernel).  Negative starts would count rows from the END in Python; nothing sensible, so they raise.
        t_rows = scores.size(0)
        if ranges.size:
            live = ranges[..., 1] > ranges[..., 0]
            if (act[:, 0] < 0).any() or (ranges[..., 0][live] < 0).any():
                raise IndexError("proposal ticks start before the first of the %d score rows" % t_rows)
            np.minimum(ranges[..., 1], t_rows, out=ranges[..., 1])
            np.minimum(act[:, 1], t_rows, out=act[:, 1])
            ranges[live & (ranges[..., 0] >= t_rows)] = -1
            act[act[:, 0] >= t_rows] = -1
        sc = scaling if torch.is_tensor(scaling) else torch.as_tensor(np.asarray(scaling))
        sc = sc.to(device=dev, dtype=torch.float32).reshape(-1, 2).contiguous()
        out_act = torch.empty((n_out, self.act_len), device=dev, dtype=torch.float32)
        out_comp = torch.empty((n_out, self.comp_len), device=dev, dtype=torch.float32)
        out_reg = torch.empty((n_out, self.reg_len), device=dev, dtype=torch.float32) if self.with_regression else None
        scores = scores.contiguous().float()
        ranges_d, act_d = torch.from_numpy(ranges).to(dev), torch.from_numpy(act).to(dev)
        cols = torch.tensor(self._part_cols, dtype=torch.int32, device=dev)
        if self.sc:
            K.stpp_reorg(scores, ranges_d, act_d, sc, cols, self.act_len, self.comp_len, self.reg_len, out_act, out_comp, out_reg)
        else:
            # no stand-alone classifier (ops/ssn_ops.py:160-161): the activity scores are pooled by stages and parts like the other two,
            # out of a block of act_len * multiplier columns -- the same launch on that block as its "completeness" block, then on the
            # columns behind it for the real completeness / regression blocks (no activity columns in either)
            none = torch.empty(1, device=dev, dtype=torch.float32)
            K.stpp_reorg(scores[:, self.act_slice], ranges_d, act_d, sc, cols, 0, self.act_len, 0, none, out_act, None)
            K.stpp_reorg(scores[:, self.comp_slice.start:], ranges_d, act_d, sc, cols, 0, self.comp_len, self.reg_len, none, out_comp, out_reg)
        return out_act, out_comp, out_reg


class OHEMHingeLoss(torch.autograd.Function):
    """Class-wise hinge loss with online hard example mining over groups of ``group_size`` rows.

    Same call signature as the reference Function (ops/ssn_ops.py:179-180); returns a [1] tensor.
    """

    @staticmethod
    def forward(ctx, pred, labels, is_positive, ohem_ratio, group_size):
        n_sample = pred.size()[0]
        assert n_sample == len(labels), "mismatch between sample size and label size"
        keep_num = int(group_size * ohem_ratio)
        pred = pred.contiguous()
        labels = labels.contiguous().long()
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        coef = torch.empty(n_sample, device=pred.device, dtype=torch.float32)
        ws = torch.empty(2 * n_sample, device=pred.device, dtype=torch.float32)
        split = group_size if is_positive > 0 else 0
        K.completeness_loss_fwd(pred, labels, loss, coef, ws, group_size, split, keep_num, keep_num, 1.0)
        ctx.save_for_backward(labels, coef)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        labels, coef = ctx.saved_tensors
        d = torch.empty(ctx.shape, device=coef.device, dtype=torch.float32)
        K.completeness_loss_bwd(labels, coef, grad_output.contiguous().reshape(1), d, 1.0)
        return d, None, None, None, None


class CompletenessLoss(torch.nn.Module):
    """/root/reference/ops/ssn_ops.py:216-239, one launch forward and one backward.

    ``global_rows`` (optional) lets a data-parallel rank reproduce the reference's denominator,
    which DataParallel computes from the gathered batch: pass the number of
    completeness rows over ALL ranks.  The loss returned is then ``local hinge sum / (global denominator / world)``,
    so that the average over the ranks (what the gradient all-reduce computes) equals the gathered-batch value.
    """

    def __init__(self, ohem_ratio=0.17):
        super(CompletenessLoss, self).__init__()
        self.ohem_ratio = ohem_ratio
        self.sigmoid = nn.Sigmoid()

    def forward(self, pred, labels, sample_split, sample_group_size, global_rows=None):
        pred_dim = pred.size()[1]
        pred = pred.reshape(-1, pred_dim)
        n_rows = pred.size(0)
        if n_rows % sample_group_size:
            raise RuntimeError("%d completeness rows cannot be viewed as groups of %d" % (n_rows, sample_group_size))
        pos_group_size = sample_split
        neg_group_size = sample_group_size - sample_split
        keep_pos = int(pos_group_size * 1.0)
        keep_neg = int(neg_group_size * self.ohem_ratio)
        rows = n_rows if global_rows is None else global_rows
        n_groups = rows // sample_group_size
        pos_cnt = n_groups * pos_group_size
        neg_cnt = int(n_groups * neg_group_size * self.ohem_ratio)
        # per-rank losses (and gradients) are AVERAGED over the ranks, like the per-rank means of the other two
        # losses: this rank's share of the global denominator makes that average equal the gathered-batch loss
        den = float(pos_cnt + neg_cnt) * (float(n_rows) / float(rows))
        return FN.CompletenessFn.apply(pred, labels.reshape(-1), sample_group_size, sample_split, keep_pos, keep_neg,
                                       den)


class SSNObjective(torch.nn.Module):
    """[r6] The training objective of /root/reference/ssn_train.py:210-214 in one launch each way:

        loss = CrossEntropyLoss()(act, act_target) + comp_loss_weight * CompletenessLoss()(comp, comp_target, sample_split, group)
               + reg_loss_weight * ClassWiseRegressionLoss()(reg, reg_labels, reg_target)

    ``forward(*ssn_outputs)`` takes the tuple ``SSN.forward`` returns (7 tensors, or 4 without regression) and returns the total;
    ``self.parts`` then holds the three component losses (a [3] tensor) for the driver's meters.  Not a class of the reference: its
    driver mixes the three criterions with Python arithmetic (five tiny kernels forward, six backward); the criterions themselves
    (ActivityLoss / CompletenessLoss / ClassWiseRegressionLoss) remain and give bit-identical components."""

    def __init__(self, comp_loss_weight=0.1, reg_loss_weight=0.1, ohem_ratio=0.17):
        super(SSNObjective, self).__init__()
        self.comp_loss_weight, self.reg_loss_weight, self.ohem_ratio = comp_loss_weight, reg_loss_weight, ohem_ratio
        self.parts = None

    def forward(self, act, act_target, comp, comp_target, reg=None, reg_labels=None, reg_target=None, sample_split=1,
                sample_group_size=7, global_rows=None):
        comp = comp.reshape(-1, comp.size()[1])
        n_rows = comp.size(0)
        if n_rows % sample_group_size:
            raise RuntimeError("%d completeness rows cannot be viewed as groups of %d" % (n_rows, sample_group_size))
        if reg is not None and (reg.dim() != 3 or reg_labels.dim() != 1):
            raise IndexError("ClassWiseRegressionLoss needs pred [n, C, 2] and labels [n] with n >= 2")      # (as the reference: ops/ssn_ops.py:253)
        neg_group_size = sample_group_size - sample_split
        keep_pos, keep_neg = int(sample_split * 1.0), int(neg_group_size * self.ohem_ratio)
        rows = n_rows if global_rows is None else global_rows
        n_groups = rows // sample_group_size
        den = float(n_groups * sample_split + int(n_groups * neg_group_size * self.ohem_ratio)) * (float(n_rows) / float(rows))
        total, self.parts = FN.TotalLossFn.apply(act, act_target, comp, comp_target.reshape(-1), reg, reg_labels, reg_target,
                                                 sample_group_size, sample_split, keep_pos, keep_neg, den, self.comp_loss_weight,
                                                 self.reg_loss_weight)
        return total


class ClassWiseRegressionLoss(torch.nn.Module):
    """Location regression loss for each class (/root/reference/ops/ssn_ops.py:242-258)."""

    def __init__(self):
        super(ClassWiseRegressionLoss, self).__init__()

    def forward(self, pred, labels, targets):
        if pred.dim() != 3 or labels.dim() != 1:
            # the reference fails here too when a forward holds a single foreground row
            # (squeeze() at ssn_models.py:282 -> IndexError at ops/ssn_ops.py:253)
            raise IndexError("ClassWiseRegressionLoss needs pred [n, C, 2] and labels [n] with n >= 2")
        return FN.ClassWiseRegressionFn.apply(pred, labels, targets)


class ActivityLoss(torch.nn.Module):
    """torch.nn.CrossEntropyLoss() as used at /root/reference/ssn_train.py:133,210, in one HIP launch."""

    def forward(self, logits, target):
        return FN.CrossEntropyFn.apply(logits, target)
