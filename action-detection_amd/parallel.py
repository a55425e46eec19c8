"""Data parallelism for the SSN hot path: one process per GPU, RCCL all-reduce over xGMI.

The reference uses single-process ``torch.nn.DataParallel`` (/root/reference/ssn_train.py:67):
parameters are re-broadcast every forward, outputs are gathered to GPU 0 where all losses are
computed, and gradients are reduced to GPU 0.  Here every rank owns whole videos (8 proposals
each -- CompletenessLoss groups rows per video, ops/ssn_ops.py:225-226), computes its own
losses, and the gradients are summed with ``torch.distributed`` (backend "nccl" == RCCL):

* backbone gradients live in one flat buffer; ``BNInception._run_backward`` reports contiguous
  tail ranges as soon as an Inception block's wgrads have been launched, and each range is
  all-reduced asynchronously while earlier blocks are still in backward;
* the three head layers are reduced in one flat bucket after ``loss.backward()``.

To reproduce the reference's numbers exactly, the completeness loss must use the GLOBAL
denominator (``CompletenessLoss(..., global_rows=)`` divides by this rank's share of it, so that the
average over the ranks is the gathered-batch loss); cross-entropy and smooth-L1 are means over equal
per-rank counts, so averaging per-rank gradients is exact.
"""
import os

import torch
import torch.distributed as dist

from . import kernels as K
from . import _lib


def _scale_inplace(t, coef):
    if t.is_cuda or _lib.emulator_active():
        K.scale_(t, None, coef)
    else:  # gloo CPU tensors in the multi-process CPU tests (no HIP device to launch on)
        t.mul_(coef)


class GradReducer:
    """Bucketed, overlapped gradient averaging for ``SSN`` across ``torch.distributed`` ranks."""

    def __init__(self, model, process_group=None, min_bucket_elems=1 << 20, deferred=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # SSN_FORCE_ALLREDUCE=1 issues the collectives even on a 1-rank group (lets a single-GPU box exercise
        # the RCCL + hipGraph-capture path the multi-GPU runs take)
        self.force = os.environ.get("SSN_FORCE_ALLREDUCE") == "1"
        self.min_bucket = min_bucket_elems
        # deferred: no collective inside the backward; the caller sums everything afterwards with reduce_all() -- ONE
        # all-reduce of the backbone's flat gradient buffer in place plus one bucket for the heads (the "separate" mode of
        # bench.py: graph(forward + backward) -> eager RCCL all-reduces -> graph(optimizer), nothing captured)
        self.deferred = deferred
        self._deferred_flat = None
        self._handles = []
        self._flat = None
        self._pend = None  # (start, end) of ranges reported but not yet launched
        model.base_model.grad_ready_hook = self
        self.launched = []  # (start, end) history of the last backward, for tests / profiling

    # --- protocol used by BNInception._run_backward
    def range_ready(self, flat, start, end):
        if self.deferred:
            self._deferred_flat = flat
            return
        if self._flat is None:
            self._flat = flat
            self.launched = []
        if self._pend is None:
            self._pend = (start, end)
        else:
            assert end == self._pend[0], "ranges must arrive tail-first and contiguous"
            self._pend = (start, self._pend[1])
        if self._pend[1] - self._pend[0] >= self.min_bucket or start == 0:
            self._launch()

    def _launch(self):
        s, e = self._pend
        self._pend = None
        if self.world > 1 or self.force:
            self._handles.append(dist.all_reduce(self._flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
        self.launched.append((s, e))

    def finish(self):
        """Called at the end of the backbone backward, before autograd consumes the gradients."""
        if self.deferred:
            return
        if self._pend is not None:
            self._launch()
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._flat is not None and (self.world > 1 or self.force):
            _scale_inplace(self._flat, 1.0 / self.world)
        self._flat = None

    def agree(self, flag):
        """Logical OR of a per-rank boolean over the ranks (one 4-byte all-reduce; every rank must call it at the same point).
        The backward executor asks it after a pass whose bucket all-reduces are already in flight: the range guard's decision
        to repeat that pass -- and issue the collectives again -- has to be the same on every rank."""
        if self.world == 1 and not self.force:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self._agree_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def agree_flag_(self, flag_tensor):
        """In-place MAX of a device fault word over the ranks (no host sync; capturable): what a graph-captured step runs in
        front of its optimizer launches, so that all ranks skip a flagged step together."""
        if self.world > 1 or self.force:
            dist.all_reduce(flag_tensor, op=dist.ReduceOp.MAX, group=self.group)
        return flag_tensor

    def _agree_device(self):
        p = next(self.model.parameters())
        backend = dist.get_backend(self.group)
        return p.device if (p.is_cuda and backend == "nccl") else torch.device("cpu")

    def reduce_all(self, average=False):
        """Deferred mode, after loss.backward(): SUM (average=True: mean) of every gradient over the ranks.  The backbone's
        parameter gradients are views of the executor's flat buffer (autograd adopts them without a copy), so that buffer is
        reduced in place -- no gather / scatter of the 42 MB; anything that does not alias it (the three heads) goes in one
        small bucket."""
        if self.world == 1 and not self.force:
            return
        flat = self._deferred_flat
        rest = []
        lo = hi = None
        if flat is not None:
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        for p in self.model.parameters():
            if p.grad is None:
                continue
            if flat is not None and lo <= p.grad.data_ptr() < hi and p.grad.is_contiguous():
                continue                       # lives in the flat buffer
            rest.append(p)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if average:
                _scale_inplace(flat, 1.0 / self.world)
        if rest:
            bucket = torch.cat([p.grad.reshape(-1) for p in rest])
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            if average:
                _scale_inplace(bucket, 1.0 / self.world)
            torch._foreach_copy_([p.grad for p in rest], [c.view_as(p.grad) for c, p in zip(bucket.split([p.grad.numel() for p in rest]), rest)])
        self.last_reduce = (0 if flat is None else flat.numel(), sum(p.grad.numel() for p in rest))

    # --- heads
    def head_parameters(self):
        """The parameters whose gradients do not live in the backbone's flat buffer: the three heads, and -- bn_mode 'partial' /
        'full' -- gamma / beta of the BatchNorm2d layers SSN.train() leaves in training mode (ssn_models.py:156-174)."""
        ps = []
        for name in ("activity_fc", "completeness_fc", "regressor_fc"):
            fc = getattr(self.model, name, None)
            if fc is not None:
                ps.extend(p for p in fc.parameters() if p.grad is not None)
        base = getattr(self.model, "base_model", None)
        for lid in (base._train_bn_ids() if hasattr(base, "_train_bn_ids") else ()):
            bn = getattr(base, lid + "_bn")
            ps.extend(p for p in (bn.weight, bn.bias) if p.grad is not None)
        return ps

    def reduce_heads(self):
        """Average the gradients outside the flat buffer -- heads, training-mode BatchNorm parameters (call after loss.backward())."""
        if self.world == 1 and not self.force:
            return
        ps = self.head_parameters()
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        _scale_inplace(flat, 1.0 / self.world)
        off = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def shard_videos(num_videos_global, rank, world):
    """Whole-video sharding: rank r owns videos [lo, hi)."""
    if num_videos_global % world:
        raise ValueError("%d videos do not split evenly over %d ranks" % (num_videos_global, world))
    per = num_videos_global // world
    return rank * per, (rank + 1) * per
