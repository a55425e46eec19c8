"""Host-side image-group transforms the reference's drivers compose around the model.

The reference builds its loader pipelines from ``transforms.py`` (inherited from tsn-pytorch) and from
``SSN.get_augmentation()`` (/root/reference/ssn_models.py:386-395, called at ssn_train.py:65; the chains are wired at
ssn_train.py:106-125 and ssn_test.py:101-112).  They run on PIL images in DataLoader workers -- CPU code in the
reference, CPU code here; the arithmetic AFTER decoding and scaling (crop, flip, channel roll, float conversion,
normalisation) also exists as one GPU launch, ``input_pipeline.GpuFrameTransform``.

Everything that draws random numbers does so through Python's ``random`` module in the reference's order, so a
seeded run picks the same crops and flips (tests/golden/ref_transforms.npz, generated from the reference's classes).
"""
import random

import numpy as np
import torch
from PIL import Image, ImageOps

from .input_pipeline import fill_fix_offset


def _pair(size):
    return (int(size), int(size)) if isinstance(size, (int, float)) else tuple(size)


class Compose(object):
    """torchvision.transforms.Compose: apply the transforms in order."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class GroupScale(object):
    """Resize so that the SHORTER edge becomes ``size`` (the old torchvision ``Scale``), bilinear; transforms.py:83-96."""

    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def _one(self, img):
        if not isinstance(self.size, int):
            return img.resize(tuple(self.size)[::-1], self.interpolation)
        w, h = img.size
        if (w <= h and w == self.size) or (h <= w and h == self.size):
            return img
        if w < h:
            return img.resize((self.size, int(self.size * h / w)), self.interpolation)
        return img.resize((int(self.size * w / h), self.size), self.interpolation)

    def __call__(self, img_group):
        return [self._one(im) for im in img_group]


class GroupCenterCrop(object):
    """transforms.py:41-46 (torchvision CenterCrop per image)."""

    def __init__(self, size):
        self.th, self.tw = _pair(size)

    def __call__(self, img_group):
        out = []
        for im in img_group:
            w, h = im.size
            x1, y1 = int(round((w - self.tw) / 2.0)), int(round((h - self.th) / 2.0))
            out.append(im.crop((x1, y1, x1 + self.tw, y1 + self.th)))
        return out


class GroupRandomCrop(object):
    """transforms.py:15-38: one random window for the whole group."""

    def __init__(self, size):
        self.th, self.tw = _pair(size)

    def __call__(self, img_group):
        w, h = img_group[0].size
        x1 = random.randint(0, w - self.tw)
        y1 = random.randint(0, h - self.th)
        if (w, h) == (self.tw, self.th):
            return list(img_group)
        return [im.crop((x1, y1, x1 + self.tw, y1 + self.th)) for im in img_group]


class GroupRandomHorizontalFlip(object):
    """transforms.py:49-64: with probability 1/2 mirror every image; for optical flow the x component (even positions
    of the group) changes sign, i.e. the 8-bit image is inverted."""

    def __init__(self, is_flow=False):
        self.is_flow = is_flow

    def __call__(self, img_group, is_flow=False):
        if random.random() >= 0.5:
            return img_group
        flipped = [im.transpose(Image.FLIP_LEFT_RIGHT) for im in img_group]
        if self.is_flow:
            flipped = [ImageOps.invert(im) if i % 2 == 0 else im for i, im in enumerate(flipped)]
        return flipped


class GroupMultiScaleCrop(object):
    """Scale-jittering crop of the training chain (transforms.py:135-206): a crop whose sides are drawn from
    ``scales`` x the shorter frame edge (neighbouring scales may pair up: aspect distortion), placed at one of 13 fixed
    positions, resized to the network input with bilinear interpolation."""

    def __init__(self, input_size, scales=None, max_distort=1, fix_crop=True, more_fix_crop=True):
        self.scales = scales if scales is not None else [1, 875, .75, .66]    # (the reference's default has this typo)
        self.max_distort = max_distort
        self.fix_crop = fix_crop
        self.more_fix_crop = more_fix_crop
        self.input_size = [input_size, input_size] if isinstance(input_size, int) else list(input_size)
        self.interpolation = Image.BILINEAR

    def sample_crop(self, im_size):
        """-> (crop_w, crop_h, offset_w, offset_h); consumes two draws of ``random`` like the reference."""
        image_w, image_h = im_size
        base = min(image_w, image_h)
        sides = [int(base * s) for s in self.scales]

        def snap(v, target):        # a side within 3 pixels of the network input is taken as the input size
            return target if abs(v - target) < 3 else v
        heights = [snap(v, self.input_size[1]) for v in sides]
        widths = [snap(v, self.input_size[0]) for v in sides]
        candidates = [(w, h) for i, h in enumerate(heights) for j, w in enumerate(widths)
                      if abs(i - j) <= self.max_distort]
        crop_w, crop_h = random.choice(candidates)
        if self.fix_crop:
            off_w, off_h = random.choice(fill_fix_offset(self.more_fix_crop, image_w, image_h, crop_w, crop_h))
        else:
            off_w = random.randint(0, image_w - crop_w)
            off_h = random.randint(0, image_h - crop_h)
        return crop_w, crop_h, off_w, off_h

    _sample_crop_size = sample_crop      # the reference's name

    def __call__(self, img_group):
        cw, ch, ow, oh = self.sample_crop(img_group[0].size)
        target = (self.input_size[0], self.input_size[1])
        return [im.crop((ow, oh, ow + cw, oh + ch)).resize(target, self.interpolation) for im in img_group]


class GroupOverSample(object):
    """Ten-crop testing (transforms.py:103-132): four corners + centre, each also mirrored (flow x inverted)."""

    def __init__(self, crop_size, scale_size=None):
        self.crop_w, self.crop_h = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self.scale_worker = GroupScale(scale_size) if scale_size is not None else None

    def __call__(self, img_group):
        if self.scale_worker is not None:
            img_group = self.scale_worker(img_group)
        image_w, image_h = img_group[0].size
        out = []
        for ow, oh in fill_fix_offset(False, image_w, image_h, self.crop_w, self.crop_h):
            crops = [im.crop((ow, oh, ow + self.crop_w, oh + self.crop_h)) for im in img_group]
            mirrored = [c.transpose(Image.FLIP_LEFT_RIGHT) for c in crops]
            mirrored = [ImageOps.invert(c) if (c.mode == 'L' and i % 2 == 0) else c for i, c in enumerate(mirrored)]
            out += crops + mirrored
        return out


class Stack(object):
    """Images -> one H x W x (sum of channels) array; ``roll``: RGB -> BGR per image (transforms.py:256-268)."""

    def __init__(self, roll=False):
        self.roll = roll

    def __call__(self, img_group):
        if img_group[0].mode == 'L':
            return np.stack([np.asarray(im) for im in img_group], axis=2)
        arrs = [np.asarray(im) for im in img_group]
        if self.roll:
            arrs = [a[:, :, ::-1] for a in arrs]
        return np.concatenate(arrs, axis=2)


class ToTorchFormatTensor(object):
    """H x W x C uint8 -> C x H x W float (optionally / 255), transforms.py:271-288."""

    def __init__(self, div=True):
        self.div = div

    def __call__(self, pic):
        if isinstance(pic, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(pic)).permute(2, 0, 1).contiguous()
        else:
            t = torch.from_numpy(np.asarray(pic).reshape(pic.size[1], pic.size[0], len(pic.mode)).copy())
            t = t.permute(2, 0, 1).contiguous()
        t = t.float()
        return t.div(255) if self.div else t


class GroupNormalize(object):
    """Per-channel (x - mean) / std with mean / std repeated over the stacked images (transforms.py:67-80)."""

    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, tensor):
        c = tensor.size(0)
        mean = torch.tensor(self.mean * (c // len(self.mean)), dtype=tensor.dtype).view(-1, 1, 1)
        std = torch.tensor(self.std * (c // len(self.std)), dtype=tensor.dtype).view(-1, 1, 1)
        k = min(mean.shape[0], std.shape[0])     # the reference's zip(tensor, rep_mean, rep_std) stops at the shortest list
        tensor[:k].sub_(mean[:k]).div_(std[:k])
        return tensor


class IdentityTransform(object):
    def __call__(self, data):
        return data
