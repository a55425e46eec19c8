"""Detections of one video from the tester's outputs: the per-video part of the reference's evaluation script
(/root/reference/eval_detection_results.py:91-178 with ops/utils.py:56-82) in one GPU call.

Input = what ``ssn_test.py:92`` puts on the result queue (``DenseTester.score_video``): relative proposal spans,
activity / completeness / regression scores.  Output = ``{class: array [n, 5]}`` of (start, end, score, loc, dur)
after score fusion, the optional top-k over all (proposal, class) pairs, per-class temporal NMS and location
regression, in the reference's order (descending score) -- the rows ``dataset_detections[cls][video_id]`` holds
before they are handed to the ActivityNet toolkit.  ``video_cls_score`` selects the ``--cls_scores`` branch (:130-144):
detections only for the ``cls_top_k`` classes an external video-level classifier ranks highest, every proposal kept,
fused score from the raw class scores (or their softmax with ``softmax_before_filter``).
"""
import numpy as np
import torch

from . import kernels as K


class DetectionPostProcessor(object):
    def __init__(self, num_class, nms_threshold, top_k=0, no_regression=False, cls_top_k=1, softmax_before_filter=False):
        self.num_class = num_class
        self.nms_threshold = float(nms_threshold)
        self.top_k = int(top_k) if top_k else 0
        self.no_regression = bool(no_regression)
        self.cls_top_k = int(cls_top_k)                              # --cls_top_k (default 1, :32)
        self.softmax_before_filter = bool(softmax_before_filter)     # --softmax_before_filter (:28)

    @torch.no_grad()
    def process_video(self, rel_prop, act_scores, comp_scores, reg_scores=None, device=None, video_cls_score=None):
        """video_cls_score: the [C] scores of this video from the external classifier's pickle (``--cls_scores``), or None."""
        dev = torch.device(device) if device is not None else act_scores.device

        def f32(t):
            return (t if torch.is_tensor(t) else torch.as_tensor(np.asarray(t))).to(dev, torch.float32).contiguous()
        rp = (rel_prop if torch.is_tensor(rel_prop) else torch.as_tensor(np.asarray(rel_prop)))
        rp = rp.reshape(-1, 2).to(dev, torch.float64).contiguous()        # (1, P, 2) in the pickles: squeezed, :93-96
        act, comp = f32(act_scores), f32(comp_scores)
        reg = None if reg_scores is None else f32(reg_scores).reshape(-1, self.num_class, 2)
        # top_k <= 0: softmax over all C+1 activity scores (:98); top_k > 0: over the C class scores (:113)
        if video_cls_score is not None:
            # every (proposal, class) pair is a candidate (top_k plays no role in this branch, :130-144); the classes the
            # external classifier does not rank among its cls_top_k best are dropped from the result
            mode = 1 if self.softmax_before_filter else 2
            combined, dets, counts = K.detections(act, comp, reg, rp, 0, mode, self.nms_threshold, not self.no_regression)
            vs = np.asarray(video_cls_score.detach().cpu() if torch.is_tensor(video_cls_score) else video_cls_score).reshape(-1)
            assert vs.shape[0] == self.num_class
            # the reference's own expression (eval_detection_results.py:137: default-kind np.argsort): with exact ties in the
            # classifier's scores the same numpy picks the same classes
            classes = sorted(int(c) for c in np.argsort(vs,)[-self.cls_top_k:])
        else:
            combined, dets, counts = K.detections(act, comp, reg, rp, self.top_k, self.top_k <= 0, self.nms_threshold,
                                                  not self.no_regression)
            classes = range(self.num_class)
        counts = counts.cpu().numpy()
        dets = dets.cpu().numpy()
        return {c: dets[c, :counts[c]].copy() for c in classes if counts[c] > 0}, combined
