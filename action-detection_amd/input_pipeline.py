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
