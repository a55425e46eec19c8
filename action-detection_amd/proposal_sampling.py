"""Which frames to load: the index / label side of the reference's ``SSNDataSet``
(/root/reference/ssn_dataset.py:11-132 records, :194-216 pools, :258-345 sampling, :347-380 per-proposal data,
:393-452 test ticks, :455-488 batch assembly), without the image I/O.

``ProposalSampler.sample_video(i)`` returns, for the 8 proposals the reference draws from video ``i`` (1 foreground,
6 incomplete, 1 background by default), everything ``SSN.forward`` takes except the pixels -- frame indices of the
9 snippets, label, normalised regression target, boundary scaling, proposal type -- consuming the numpy RNG in the
reference's order, so the same seed selects the same proposals and frames.  ``test_ticks(video)`` is the tester's
counterpart (frame ticks, relative spans, proposal ticks, scaling).  The decoded frames then go through
``input_pipeline.GpuFrameTransform``.
"""
import math
from collections import namedtuple

import numpy as np
from numpy.random import randint

from .proposal_io import load_proposal_file

FG, INCOMPLETE, BG = 0, 1, 2

SampledProposal = namedtuple("SampledProposal", "video_id frame_indices label reg_target scaling stage_split prop_type")


def temporal_iou(a, b):
    """IoU of two (start, end) spans (/root/reference/ops/utils.py:40-53)."""
    lo, hi = max(a[0], b[0]), min(a[1], b[1])
    if lo >= hi:
        return 0
    return float(hi - lo) / float(max(a[1], b[1]) - min(a[0], b[0]))


class Instance(object):
    """A ground-truth instance or a proposal of one video (frame units)."""

    def __init__(self, start_frame, end_frame, video_frame_count, label=None, best_iou=None, overlap_self=None):
        self.start_frame = start_frame
        self.end_frame = min(end_frame, video_frame_count)
        self.label = label if label is not None else -1
        self.coverage = (end_frame - start_frame) / video_frame_count
        self.best_iou = best_iou
        self.overlap_self = overlap_self
        self.loc_reg = None
        self.size_reg = None

    def set_regression_targets(self, gt_list, fg_thresh):
        """Centre shift (in proposal lengths) and log size ratio w.r.t. the best-overlapping ground truth."""
        if self.best_iou < fg_thresh:
            return
        ious = [temporal_iou((self.start_frame, self.end_frame), (g.start_frame, g.end_frame)) for g in gt_list]
        gt = gt_list[int(np.argmax(ious))]
        size, gt_size = self.end_frame - self.start_frame + 1, gt.end_frame - gt.start_frame + 1
        self.loc_reg = ((gt.start_frame + gt.end_frame) / 2 - (self.start_frame + self.end_frame) / 2) / size
        self.size_reg = math.log(gt_size / size)

    @property
    def regression_targets(self):
        return [self.loc_reg, self.size_reg] if self.loc_reg is not None else [0, 0]


class VideoRecord(object):
    """One record of a processed proposal list (``proposal_io.load_proposal_file``)."""

    def __init__(self, record):
        self.id, n = record[0], int(record[1])
        self.num_frames = n
        self.gt = [Instance(int(x[1]), int(x[2]), n, label=int(x[0]), best_iou=1.0)
                   for x in record[2] if int(x[2]) > int(x[1])]
        self.gt = [g for g in self.gt if g.start_frame < n]
        self.proposals = [Instance(int(x[3]), int(x[4]), n, label=int(x[0]), best_iou=float(x[1]),
                                   overlap_self=float(x[2])) for x in record[3] if int(x[4]) > int(x[3])]
        self.proposals = [p for p in self.proposals if p.start_frame < n]

    def foreground(self, fg_thresh, with_gt=True):
        fg = [p for p in self.proposals if p.best_iou > fg_thresh]
        if with_gt:
            fg.extend(self.gt)
        for p in fg:
            p.set_regression_targets(self.gt, fg_thresh)
        return fg

    def negatives(self, incomplete_iou_thresh, bg_iou_thresh, bg_coverage_thresh=0.01, incomplete_overlap_thresh=0.7):
        """-> (incomplete, background): low IoU but mostly inside a ground truth / low IoU and not tiny."""
        incomplete = [p for p in self.proposals
                      if p.best_iou < incomplete_iou_thresh and p.overlap_self > incomplete_overlap_thresh]
        taken = set(id(p) for p in incomplete)
        background = [p for p in self.proposals
                      if id(p) not in taken and p.best_iou < bg_iou_thresh and p.coverage > bg_coverage_thresh]
        return incomplete, background


class ProposalSampler(object):
    def __init__(self, prop_file=None, records=None, body_seg=5, aug_seg=2, video_centric=True, new_length=1,
                 random_shift=True, prop_per_video=8, fg_ratio=1, bg_ratio=1, incomplete_ratio=6,
                 fg_iou_thresh=0.7, bg_iou_thresh=0.01, incomplete_iou_thresh=0.3, bg_coverage_thresh=0.02,
                 incomplete_overlap_thresh=0.7, gt_as_fg=True, reg_stats=None, test_interval=6, exclude_empty=True):
        self.body_seg, self.aug_seg, self.new_length = body_seg, aug_seg, new_length
        self.video_centric, self.random_shift, self.test_interval = video_centric, random_shift, test_interval
        self.fg_iou_thresh, self.bg_iou_thresh, self.incomplete_iou_thresh = fg_iou_thresh, bg_iou_thresh, incomplete_iou_thresh
        self.bg_coverage_thresh, self.incomplete_overlap_thresh = bg_coverage_thresh, incomplete_overlap_thresh
        self.gt_as_fg = gt_as_fg
        self.starting_ratio = self.ending_ratio = 0.5
        total = fg_ratio + bg_ratio + incomplete_ratio
        self.fg_per_video = int(prop_per_video * (fg_ratio / total))
        self.bg_per_video = int(prop_per_video * (bg_ratio / total))
        self.incomplete_per_video = prop_per_video - self.fg_per_video - self.bg_per_video

        records = load_proposal_file(prop_file) if records is None else records
        self.video_list = [VideoRecord(r) for r in records]
        if exclude_empty:
            self.video_list = [v for v in self.video_list if len(v.gt) > 0]
        self.video_dict = {v.id: v for v in self.video_list}
        self.fg_pool, self.bg_pool, self.incomp_pool = [], [], []
        for v in self.video_list:
            self.fg_pool.extend((v.id, p) for p in v.foreground(fg_iou_thresh, gt_as_fg))
            inc, bg = self._negatives(v)
            self.incomp_pool.extend((v.id, p) for p in inc)
            self.bg_pool.extend((v.id, p) for p in bg)
        if reg_stats is None:
            targets = [list(p.regression_targets) for v in self.video_list for p in v.foreground(fg_iou_thresh, False)]
            self.stats = np.array((np.mean(targets, axis=0), np.std(targets, axis=0)))
        else:
            self.stats = reg_stats

    def __len__(self):
        return len(self.video_list)

    def _negatives(self, video):
        return video.negatives(self.incomplete_iou_thresh, self.bg_iou_thresh, self.bg_coverage_thresh,
                               self.incomplete_overlap_thresh)

    # ---- which proposals (ssn_dataset.py:258-291)
    def _draw(self, prop_type, video, pool, count, dataset_pool):
        if len(pool) == 0:      # nothing of this type in the video: borrow from the whole data set
            return [(dataset_pool[x], prop_type) for x in np.random.choice(len(dataset_pool), count, replace=False)]
        picks = np.random.choice(len(pool), count, replace=len(pool) < count)
        return [((video.id, pool[x]), prop_type) for x in picks]

    def pick_proposals(self, video):
        if not self.video_centric:
            out = [(x, FG) for x in np.random.choice(self.fg_pool, self.fg_per_video, replace=False)]
            out += [(x, INCOMPLETE) for x in np.random.choice(self.incomp_pool, self.incomplete_per_video, replace=False)]
            return out + [(x, BG) for x in np.random.choice(self.bg_pool, self.bg_per_video, replace=False)]
        fg = video.foreground(self.fg_iou_thresh, self.gt_as_fg)
        inc, bg = self._negatives(video)
        return (self._draw(FG, video, fg, self.fg_per_video, self.fg_pool)
                + self._draw(INCOMPLETE, video, inc, self.incomplete_per_video, self.incomp_pool)
                + self._draw(BG, video, bg, self.bg_per_video, self.bg_pool))

    # ---- which frames (ssn_dataset.py:293-345)
    def _segment_offsets(self, valid_length, num_seg):
        if not self.random_shift:       # validation: segment centres
            if valid_length > num_seg:
                tick = valid_length / float(num_seg)
                return np.array([int(tick / 2.0 + tick * x) for x in range(num_seg)])
            return np.zeros((num_seg,))
        span = (valid_length + 1) // num_seg
        if span > 0:
            return np.multiply(list(range(num_seg)), span) + randint(span, size=num_seg)
        if valid_length > num_seg:
            return np.sort(randint(valid_length, size=num_seg))
        return np.zeros((num_seg,))

    def snippet_indices(self, prop, frame_cnt):
        """-> (frame index of each of the aug + body + aug snippets, starting scale, ending scale, stage split)"""
        start, end = prop.start_frame + 1, prop.end_frame
        duration = end - start + 1
        assert duration != 0, (prop.start_frame, prop.end_frame, prop.best_iou)
        first = max(1, start - int(duration * self.starting_ratio))
        last = min(frame_cnt - self.new_length + 1, end + int(duration * self.ending_ratio))
        len_start = start - first - self.new_length + 1
        len_end = last - end - self.new_length + 1
        scale_start = (len_start + self.new_length - 1) / (duration * self.starting_ratio)
        scale_end = (len_end + self.new_length - 1) / (duration * self.ending_ratio)
        offsets = np.concatenate((self._segment_offsets(len_start, self.aug_seg) + first,
                                  self._segment_offsets(duration - self.new_length, self.body_seg) + start,
                                  self._segment_offsets(len_end, self.aug_seg) + end))
        split = [self.aug_seg, self.aug_seg + self.body_seg, self.aug_seg * 2 + self.body_seg]
        return offsets, scale_start, scale_end, split

    def describe(self, picked):
        """One picked ((video id, instance), type) pair -> SampledProposal (ssn_dataset.py:347-380 without the images)."""
        (vid, inst), prop_type = picked
        frame_cnt = self.video_dict[vid].num_frames
        offsets, s0, s1, split = self.snippet_indices(inst, frame_cnt)
        if prop_type not in (FG, INCOMPLETE, BG):
            raise ValueError()
        label = 0 if prop_type == BG else inst.label
        frames = [min(frame_cnt, int(seg) + x) for seg in offsets for x in range(self.new_length)]
        if prop_type == FG:
            t = inst.regression_targets
            reg = ((t[0] - self.stats[0][0]) / self.stats[1][0], (t[1] - self.stats[0][1]) / self.stats[1][1])
        else:
            reg = (0.0, 0.0)
        return SampledProposal(vid, frames, label, reg, (s0, s1), split, prop_type)

    def sample_video(self, index):
        """The proposals of one training sample, in the reference's order (foreground, incomplete, background), and the
        per-video arrays ``get_training_data`` returns besides the frames (scaling, type, labels, reg targets)."""
        video = self.video_list[index % len(self.video_list)]
        props = [self.describe(p) for p in self.pick_proposals(video)]
        return props, {
            "scaling": np.array([p.scaling for p in props], dtype=np.float32),
            "prop_type": np.array([p.prop_type for p in props]),
            "labels": np.array([p.label for p in props]),
            "reg_targets": np.array([p.reg_target for p in props], dtype=np.float32),
        }

    # ---- testing (ssn_dataset.py:393-452 without the images)
    def test_ticks(self, video):
        """-> (frame ticks, relative spans [P, 2], proposal ticks [P, 4], scaling [P, 2]) of one video"""
        frame_cnt = video.num_frames
        ticks = np.arange(0, frame_cnt - self.new_length, self.test_interval, dtype=int) + 1
        n = len(ticks)
        props = video.proposals if len(video.proposals) else [Instance(0, frame_cnt - 1, frame_cnt)]
        rel, pticks, scaling = [], [], []
        for p in props:
            r0, r1 = p.start_frame / frame_cnt, p.end_frame / frame_cnt
            d_start, d_end = (r1 - r0) * self.starting_ratio, (r1 - r0) * self.ending_ratio
            lo, hi = max(0.0, r0 - d_start), min(1.0, r1 + d_end)
            rel.append((r0, r1))
            pticks.append((int(lo * n), int(r0 * n), int(r1 * n), int(hi * n)))
            scaling.append(((r0 - lo) / d_start, (hi - r1) / d_end))
        return ticks, np.array(rel), np.array(pticks), np.array(scaling)

    def all_gt(self):
        return [[v.id, g.label - 1, g.start_frame / v.num_frames, g.end_frame / v.num_frames]
                for v in self.video_list for g in v.gt]
