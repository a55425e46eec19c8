"""Dense testing of one video: the loop body of the reference's tester
(/root/reference/ssn_test.py:66-92) on the MI355X kernels.

The reference scores every sampled frame ("tick") of a video with ``SSN.test_forward``
(backbone -> folded ``test_fc``), averages the ``num_crop`` crops of a tick, and hands the
per-tick score matrix to ``STPPReorgainzed`` together with the proposal ticks; regression
outputs are de-normalised with the training-set statistics.  Same inputs, same outputs here:

* the backbone runs on as many ticks per call as ``tick_batch`` asks for (the reference's
  generator yields 4 ticks x 10 crops = 40 frames per call, ssn_dataset.py:393; a 288 GB GPU
  takes hundreds), whatever batching the frame source uses;
* ``test_fc`` is linear, so ``fc(x).view(crops, -1, D).mean(0)`` (ssn_test.py:84-85) is computed
  as ``fc(mean over crops of x)``: one ``ssn_crop_mean`` launch on the 1024-d features, then the
  folded FC on 10x fewer rows;
* ``STPPReorgainzed`` is one launch per video (ops/ssn_ops.py:109-170 runs Python loops over
  proposals, stages and parts); the de-normalisation is ``ssn_reg_denorm``.
"""
import torch

from . import kernels as K
from .ops.ssn_ops import STPPReorgainzed


class DenseTester(object):
    """``net``: an ``SSN(..., test_mode=True)`` with ``prepare_test_fc()`` done, in eval mode, on a HIP device.
    ``stats``: the 2x2 regression statistics array of the checkpoint (``[[mean0, mean1], [std0, std1]]``)."""

    def __init__(self, net, num_class, stpp_cfg=(1, 1, 1), stats=None, tick_batch=32):
        if net.test_fc is None:
            raise RuntimeError("call net.prepare_test_fc() first (ssn_test.py:61)")
        self.net = net
        self.num_class = num_class
        self.with_regression = net.with_regression
        self.output_dim = net.test_fc.out_features
        self.reorg = STPPReorgainzed(self.output_dim, num_class + 1, num_class, num_class * 2, True,
                                     with_regression=net.with_regression, stpp_cfg=stpp_cfg)
        self.stats = stats
        self.tick_batch = int(tick_batch)
        self.length = (3 if net.modality == "RGB" else 2) * net.new_length

    @torch.no_grad()
    def frame_scores(self, frames_gen, frame_cnt, num_crop):
        """[frame_cnt, output_dim] crop-averaged test_fc scores; ``frames_gen`` yields crop-major frame batches
        (``[num_crop * b, length, H, W]`` or anything that views to it), as ssn_dataset.get_test_data does."""
        dev = self.net.test_fc.weight.device
        output = torch.empty((frame_cnt, self.output_dim), device=dev, dtype=torch.float32)
        cnt = 0
        pending, pending_ticks = [], 0

        def flush():
            nonlocal cnt, pending, pending_ticks
            if not pending:
                return
            # re-pack crop-major sub-batches [crop][tick] into one crop-major batch
            if len(pending) == 1:
                x = pending[0]
            else:
                x = torch.cat([p.reshape((num_crop, -1) + tuple(p.shape[1:])) for p in pending], dim=1)
                x = x.reshape((-1,) + tuple(pending[0].shape[1:]))
            base = self.net._backbone(x.contiguous())                       # [num_crop * b, feat]
            feat = torch.empty((pending_ticks, base.shape[1]), device=dev, dtype=torch.float32)
            K.crop_mean(base.contiguous(), num_crop, feat)
            sc = self.net.test_fc(feat)
            output[cnt:cnt + pending_ticks] = sc
            cnt += pending_ticks
            pending, pending_ticks = [], 0

        for frames in frames_gen:
            x = frames.to(dev, non_blocking=True).reshape((-1, self.length) + tuple(frames.shape[-2:]))
            if x.shape[0] % num_crop:
                raise ValueError("a frame batch of %d images is not a multiple of %d crops" % (x.shape[0], num_crop))
            pending.append(x)
            pending_ticks += x.shape[0] // num_crop
            if pending_ticks >= self.tick_batch:
                flush()
        flush()
        if cnt != frame_cnt:
            raise ValueError("the frame source gave %d ticks, expected %d" % (cnt, frame_cnt))
        return output

    @torch.no_grad()
    def score_video(self, frames_gen, frame_cnt, prop_ticks, prop_scaling, num_crop=10):
        """-> (act_scores [P, C+1], comp_scores [P, C], reg_scores [P, C, 2] or None, output [frame_cnt, D])
        -- the tensors ssn_test.py:92 puts on the result queue."""
        output = self.frame_scores(frames_gen, frame_cnt, num_crop)
        act, comp, reg = self.reorg.forward(output, prop_ticks, prop_scaling)
        if reg is not None:
            reg = reg.reshape(-1, self.num_class, 2)
            if self.stats is not None:
                K.reg_denorm(reg, self.stats[0][0], self.stats[1][0], self.stats[0][1], self.stats[1][1])
        return act, comp, reg, output
