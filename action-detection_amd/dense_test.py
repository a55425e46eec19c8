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
  proposals, stages and parts); the de-normalisation is ``ssn_reg_denorm``;
* the range guard of the backbone's delayed scales (planes_exec.py) is polled ONCE per video instead of once per backbone call:
  every call leaves its fault word in a per-call slot, the inputs of the video's calls stay referenced, and after the last call
  one host read says which calls (if any) have to be repeated -- with the per-call poll the host-side preparation of every call
  sat exposed behind a drained queue (8 % of a video: 19.7 -> 21.4 k frames/s on Inception-v3).
"""
import torch

from . import kernels as K
from .ops.ssn_ops import STPPReorgainzed


class DenseTester(object):
    """``net``: an ``SSN(..., test_mode=True)`` with ``prepare_test_fc()`` done, in eval mode, on a HIP device.
    ``stats``: the 2x2 regression statistics array of the checkpoint (``[[mean0, mean1], [std0, std1]]``)."""

    def __init__(self, net, num_class, stpp_cfg=(1, 1, 1), stats=None, tick_batch=32, max_keep_bytes=None):
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
        # inputs kept referenced for a possible repeat (the once-per-video poll); beyond the cap the calls poll one by one.  None: half of
        # the device memory that is free when a video starts, at most 24 GiB
        self.max_keep_bytes = max_keep_bytes
        self.repeated_calls = 0             # backbone calls the once-per-video poll had to repeat

    @torch.no_grad()
    def frame_scores(self, frames_gen, frame_cnt, num_crop):
        """[frame_cnt, output_dim] crop-averaged test_fc scores; ``frames_gen`` yields crop-major frame batches
        (``[num_crop * b, length, H, W]`` or anything that views to it), as ssn_dataset.get_test_data does."""
        dev = self.net.test_fc.weight.device
        output = torch.empty((frame_cnt, self.output_dim), device=dev, dtype=torch.float32)
        cnt = 0
        pending, pending_ticks = [], 0
        bm = self.net.base_model
        lag = {"on": (getattr(bm, "scale_guard", "") == "sync" and dev.type == "cuda" and getattr(bm, "layout", "") == "planes"),
               "kept": [], "marks": [], "bytes": 0}
        word = bm.planes_flag(dev)[0:1] if lag["on"] else None
        # the fault word is shared by every user of the backbone on this device (a graph owner polls it): what it held before this video
        # is put back at the end -- the tester only ever clears what its own calls set
        prior = word.clone() if lag["on"] else None
        keep_cap = self.max_keep_bytes
        if keep_cap is None:
            keep_cap = min(24 << 30, torch.cuda.mem_get_info(dev)[0] // 2) if dev.type == "cuda" else 0

        def score(x, row0, ticks):
            base = self.net._backbone(x)                                    # [num_crop * b, feat]
            feat = torch.empty((ticks, base.shape[1]), device=dev, dtype=torch.float32)
            K.crop_mean(base.contiguous(), num_crop, feat)
            output[row0:row0 + ticks] = self.net.test_fc(feat)

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
            x = x.contiguous()
            if lag["on"] and lag["bytes"] + x.numel() * x.element_size() > keep_cap:
                lag["on"] = False                                           # (a very long video: poll call by call from here on)
            if lag["on"]:
                bm.scale_guard = "deferred"                                 # the call launches its range check, polls nothing
                try:
                    score(x, cnt, pending_ticks)
                finally:
                    bm.scale_guard = "sync"
                lag["marks"].append(word.clone())                           # this call's fault word (device-side copy, no sync)
                word.zero_()
                lag["kept"].append((x, cnt, pending_ticks))
                lag["bytes"] += x.numel() * x.element_size()
            else:
                score(x, cnt, pending_ticks)
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
        if lag["marks"]:
            bad = [i for i, v in enumerate(torch.cat(lag["marks"]).tolist()) if v]      # the one host read of the video
            for i in bad:                                                   # repeat exactly the calls that left their range
                self.repeated_calls += 1
                score(*lag["kept"][i])                                      # (sync guard: repairs itself)
        if prior is not None:
            torch.maximum(word, prior, out=word)
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
