"""GPU side of the input pipeline: the reference's transform chain after image decoding and scaling.

``ssn_test.py:101-112`` builds ``GroupOverSample(input_size, scale_size) -> Stack(roll) ->
ToTorchFormatTensor(div=False) -> GroupNormalize(mean, std)`` (``ssn_train.py:106-111`` the same tail behind a
random crop + flip) and runs it image by image in PIL on the loader workers.  Here the decoded, scaled frames go
to the GPU as uint8 and one launch (``ssn_frames_crop_normalize``) produces the network input in the layout
``SSN.forward`` / ``DenseTester`` expect (crop-major ``[crop][tick]`` rows).  Decoding and PIL's antialiased resize
stay on the CPU side (not arithmetic of this repo's path).
"""
import torch

from . import kernels as K


def fill_fix_offset(more_fix_crop, image_w, image_h, crop_w, crop_h):
    """/root/reference/transforms.py:184-206 (GroupMultiScaleCrop.fill_fix_offset)."""
    w_step = (image_w - crop_w) // 4
    h_step = (image_h - crop_h) // 4
    ret = [(0, 0), (4 * w_step, 0), (0, 4 * h_step), (4 * w_step, 4 * h_step), (2 * w_step, 2 * h_step)]
    if more_fix_crop:
        ret += [(0, 2 * h_step), (4 * w_step, 2 * h_step), (2 * w_step, 4 * h_step), (2 * w_step, 0 * h_step),
                (1 * w_step, 1 * h_step), (3 * w_step, 1 * h_step), (1 * w_step, 3 * h_step), (3 * w_step, 3 * h_step)]
    return ret


class GpuFrameTransform(object):
    """``mean`` / ``std`` / ``roll`` as the reference passes them (``net.input_mean``, ``net.input_std``,
    ``roll = arch in ('BNInception', 'InceptionV3')``); ``is_flow``: single-channel images, x component inverted
    under a flip."""

    def __init__(self, crop_size, mean, std, roll=True, is_flow=False, device="cuda:0"):
        self.crop_h, self.crop_w = (crop_size, crop_size) if isinstance(crop_size, int) else crop_size
        self.roll = bool(roll) and not is_flow
        self.is_flow = bool(is_flow)
        self.mean = torch.tensor([float(m) for m in mean], dtype=torch.float32, device=device)
        self.std = torch.tensor([float(s) for s in std], dtype=torch.float32, device=device)

    def _run(self, frames, crops):
        if frames.dtype != torch.uint8 or frames.dim() != 4:
            raise ValueError("frames: uint8 [n_img, H, W, C] as decoded")
        frames = frames.to(self.mean.device).contiguous()
        n_img, _, _, c = frames.shape
        out = torch.empty((len(crops), n_img, c, self.crop_h, self.crop_w), dtype=torch.float32, device=frames.device)
        K.frames_crop_normalize(frames, out, crops, self.roll, self.is_flow, self.mean, self.std)
        return out

    def oversample(self, frames):
        """GroupOverSample: 5 fixed crops x {as is, flipped} -> [10 * n_img * C, crop_h, crop_w], the tensor
        ``ssn_dataset.get_test_data`` yields per frame batch (view it as (-1, length, H, W))."""
        offs = fill_fix_offset(False, frames.shape[2], frames.shape[1], self.crop_w, self.crop_h)
        crops = []
        for ow, oh in offs:
            crops += [(ow, oh, 0), (ow, oh, 1)]
        return self._run(frames, crops).reshape(-1, self.crop_h, self.crop_w)

    def crop(self, frames, off_w, off_h, flip):
        """One crop (+ optional flip) of every frame: the tail of the training chain -> [n_img * C, crop_h, crop_w]."""
        return self._run(frames, [(off_w, off_h, flip)]).reshape(-1, self.crop_h, self.crop_w)


class GpuTrainAugment(GpuFrameTransform):
    """``SSN.get_augmentation()`` + the tail of the training chain on the GPU (/root/reference/ssn_models.py:386-395,
    ssn_train.py:106-111): ``GroupMultiScaleCrop(input_size, scales)`` -> ``GroupRandomHorizontalFlip(is_flow)`` -> ``Stack(roll)`` ->
    ``ToTorchFormatTensor(div=False)`` -> ``GroupNormalize`` on DECODED uint8 frames -- crop, PIL's bilinear resize (restated
    bit-exactly, csrc/frames.hip), flip and normalisation in two launches; the loader workers only decode.

    Crop boxes and flips are drawn per GROUP (the reference transforms the frames of one proposal together,
    ssn_dataset.py:347-380) through Python's ``random`` in the reference's order -- two draws for the box, one for the flip -- so a
    seeded run picks the reference's crops (tests/golden/ref_transforms.npz)."""

    def __init__(self, input_size, mean, std, scales, roll=True, is_flow=False, device="cuda:0", max_distort=1, fix_crop=True,
                 more_fix_crop=True):
        super().__init__(input_size, mean, std, roll=roll, is_flow=is_flow, device=device)
        self.scales = list(scales)
        self.max_distort, self.fix_crop, self.more_fix_crop = max_distort, fix_crop, more_fix_crop

    def sample(self, frame_wh, n_groups):
        """-> (boxes [n_groups][4] = (x0, y0, crop_w, crop_h), flips [n_groups]) with the reference's draws per group."""
        import random
        from .transforms import GroupMultiScaleCrop
        msc = GroupMultiScaleCrop([self.crop_w, self.crop_h], self.scales, self.max_distort, self.fix_crop, self.more_fix_crop)
        boxes, flips = [], []
        for _ in range(n_groups):
            cw, ch, ow, oh = msc.sample_crop(frame_wh)
            boxes.append((ow, oh, cw, ch))
            flips.append(random.random() < 0.5)
        return boxes, flips

    def __call__(self, frames, group_size, boxes=None, flips=None):
        """frames uint8 [n_img, H, W, C] (n_img = groups x group_size, decoded size) -> fp32 [n_img * C, crop_h, crop_w]."""
        if frames.dtype != torch.uint8 or frames.dim() != 4:
            raise ValueError("frames: uint8 [n_img, H, W, C] as decoded")
        n_img, h, w, c = frames.shape
        if n_img % group_size:
            raise ValueError("%d frames do not split into groups of %d" % (n_img, group_size))
        if boxes is None:
            boxes, flips = self.sample((w, h), n_img // group_size)
        per_box = [b for b in boxes for _ in range(group_size)]
        per_flip = [f for f in flips for _ in range(group_size)]
        frames = frames.to(self.mean.device).contiguous()
        out = K.frames_crop_resize_normalize(frames, per_box, per_flip, (self.crop_h, self.crop_w), self.roll, self.is_flow,
                                             self.mean, self.std)
        return out.reshape(-1, self.crop_h, self.crop_w)


class TrainingBatchPrefetcher(object):
    """Keeps the GPU fed during training.

    The reference's loader workers produce finished fp32 tensors and the training loop uploads them synchronously
    (``ssn_train.py:205-208``; its ``Data`` meter is that wait, 4 bytes per pixel over PCIe).  Here the workers stop after
    the PIL part -- decode, scale-jittered crop + resize, flip -- and hand over the uint8 frames; a background thread stages
    batch i + 1 in pinned memory, a side HIP stream uploads it (1 byte per pixel) and runs the ``Stack(roll) ->
    ToTorchFormatTensor(div=False) -> GroupNormalize`` tail as one launch (``ssn_frames_crop_normalize``) while the compute
    stream is busy with batch i; ``next()`` only makes the compute stream wait on that batch's event.

    ``source``: iterable of ``(frames, scaling, target, reg_target, prop_type)`` with ``frames`` uint8
    ``[videos, images per video, H, W, C]`` (numpy or torch, already at the network's input size); ``transform``: a
    ``GpuFrameTransform`` for the same device.  Yields ``(input [videos, images * C, H, W] fp32, scaling, target, reg_target,
    prop_type)`` on the device -- the five arguments of ``SSN.forward``.  ``depth`` batches are in flight (>= 2).  On a
    non-HIP device (the host emulator of the tests) it degrades to a synchronous loop with the same results.
    """

    def __init__(self, source, transform, depth=2, group_size=None):
        import queue
        import threading
        self.transform = transform
        # with a GpuTrainAugment the frames arrive UN-cropped at their decoded size and the scale-jittered crop + resize + flip run
        # on the GPU too; group_size = images that share one crop box (the frames of one proposal: segments x new_length)
        self.group_size = group_size
        if isinstance(transform, GpuTrainAugment) and not group_size:
            raise ValueError("GpuTrainAugment needs group_size (images per proposal)")
        self.device = transform.mean.device
        self.cuda = self.device.type == "cuda"
        self.depth = max(2, int(depth))
        self._source = iter(source)
        self._slots = [dict(pinned=None, dev=None, uploaded=None) for _ in range(self.depth)]
        self._q = queue.Queue(maxsize=self.depth)
        self._stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._stop = False
        self._done = None
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._thread.start()

    # -- producer thread: stage, upload, transform ------------------------------------------------------------------
    def _produce(self):
        try:
            i = 0
            for item in self._source:
                if self._stop:
                    break
                slot = self._slots[i % self.depth]
                i += 1
                if not self._put(("ok", self._stage(slot, item))):     # waits while `depth` batches are queued
                    return
            self._put(("end", None))
        except BaseException as e:      # surfaced on the consumer's thread
            self._put(("err", e))

    def _put(self, item):
        """queue.put that gives up when close() was called (a producer blocked on a full queue must be able to exit)."""
        import queue
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _stage(self, slot, item):
        frames, scaling, target, reg_target, prop_type = item
        frames = torch.as_tensor(frames)
        if frames.dtype != torch.uint8 or frames.dim() != 5:
            raise ValueError("frames: uint8 [videos, images, H, W, C]")
        v, n_img, h, w, c = frames.shape
        small = [torch.as_tensor(t) for t in (scaling, target, reg_target, prop_type)]
        if not self.cuda:
            out = self._transform(frames.reshape(v * n_img, h, w, c))
            ch, cw = out.shape[-2], out.shape[-1]      # the transform's crop size (the frames may be larger)
            return (out.reshape(v, n_img * c, ch, cw),) + tuple(t.to(self.device) for t in small), None
        if slot["pinned"] is None or slot["pinned"].shape != frames.shape:
            slot["pinned"] = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
            slot["dev"] = torch.empty(frames.shape, dtype=torch.uint8, device=self.device)
        if slot["uploaded"] is not None:
            slot["uploaded"].synchronize()      # the upload that last read this pinned buffer (`depth` batches ago) is done
        slot["pinned"].copy_(frames)
        with torch.cuda.stream(self._stream):
            slot["dev"].copy_(slot["pinned"], non_blocking=True)    # (the device buffer is only touched on this stream)
            slot["uploaded"] = torch.cuda.Event()
            slot["uploaded"].record(self._stream)
            out = self._transform(slot["dev"].reshape(v * n_img, h, w, c))
            out = out.reshape(v, n_img * c, out.shape[-2], out.shape[-1])      # the transform's crop size
            rest = tuple(t.pin_memory().to(self.device, non_blocking=True) for t in small)
            ready = torch.cuda.Event()
            ready.record(self._stream)
        return (out,) + rest, ready

    def _transform(self, frames):
        if isinstance(self.transform, GpuTrainAugment):
            return self.transform(frames, self.group_size)
        return self.transform.crop(frames, 0, 0, False)

    # -- consumer ------------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        if self._done is not None:          # exhausted / failed / closed: every further next() says so again
            raise self._done
        kind, payload = self._q.get()
        if kind == "end":
            self._done = StopIteration()
            raise StopIteration
        if kind == "err":
            self._done = StopIteration()
            raise payload
        batch, ready = payload
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
            for t in batch:
                t.record_stream(torch.cuda.current_stream(self.device))   # allocated on the side stream, used here
        return batch

    def close(self):
        """Stop the producer thread and release its staging buffers (safe to call more than once)."""
        import queue
        self._stop = True
        if self._done is None:
            self._done = StopIteration()
        while True:                          # unblock a producer waiting on the full queue, drop what it staged
            try:
                self._q.get_nowait()
            except queue.Empty:
                break
        self._thread.join(timeout=5.0)
        self._slots = []
