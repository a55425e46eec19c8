"""Input-side transforms against tests/golden/ref_transforms.npz, which holds what the REFERENCE's own transforms.py
classes produced on seeded PIL images (oracle/make_golden.py:golden_transforms):

* the oracle's numpy restatement of the test chain (``O.oversample_transform``)  -- (a)-tier;
* the product's host-side transform classes (``action_detection_amd.transforms``): test chain, and the training
  augmentation ``SSN.get_augmentation()`` returns, with the same ``random`` seeds (same crops, same flips, same pixels);
* the GPU launch (``GpuFrameTransform``) for the test chain, bit-exact (emulator on CPU, libssn_hip.so with -m gpu).
"""
import os
import random

import numpy as np
import torch
from PIL import Image

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd import transforms as T
from action_detection_amd.input_pipeline import GpuFrameTransform

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_transforms.npz"))
CASES = (("over_rgb_roll", "rgb", (24, 24), [104, 117, 128], [1], True),
         ("over_rgb_std", "rgb", (28, 20), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], False),   # (crop_w, crop_h)
         ("over_flow", "flow", (22, 22), [128], [1], True))


def test_oracle_oversample_matches_reference_transforms():
    for key, kind, (cw, ch), mean, std, roll in CASES:
        frames = G[kind + "_frames"]
        got = O.oversample_transform([f for f in frames], cw, ch, mean, std, roll, kind == "flow")
        assert np.array_equal(got.numpy(), G[key]), key


def test_host_transform_classes_match_reference_transforms():
    for key, kind, (cw, ch), mean, std, roll in CASES:
        imgs = [Image.fromarray(f, "RGB" if kind == "rgb" else "L") for f in G[kind + "_frames"]]
        chain = T.Compose([T.GroupOverSample((cw, ch)), T.Stack(roll=roll), T.ToTorchFormatTensor(div=False),
                           T.GroupNormalize(mean, std)])
        assert np.array_equal(chain(imgs).numpy(), G[key]), key


def test_get_augmentation_matches_reference_with_the_same_seeds():
    from action_detection_amd.ssn_models import SSN
    for tag, modality, mode in (("rgb", "RGB", "RGB"), ("flow", "Flow", "L")):
        net = SSN(20, 2, 5, 2, modality, dropout=0.8)
        net.input_size = 56                                   # the fixture's network input (small frames)
        aug = net.get_augmentation()
        imgs = [Image.fromarray(f, mode) for f in G["aug_%s_frames" % tag]]
        for seed in range(6):
            random.seed(seed)
            out = np.stack([np.asarray(im) for im in aug(imgs)])
            assert np.array_equal(out, G["aug_%s_%d" % (tag, seed)]), (tag, seed)
    # the crop-parameter sampler alone over real frame sizes (and the flip coin that follows it)
    msc = T.GroupMultiScaleCrop(224, [1, .875, .75, .66])
    for w, h, seed, cw, ch, ow, oh, flip in G["crop_params"]:
        random.seed(int(seed))
        assert msc.sample_crop((int(w), int(h))) == (cw, ch, ow, oh)
        assert int(random.random() < 0.5) == flip


def test_gpu_oversample_matches_reference_transforms(backend):
    for key, kind, (cw, ch), mean, std, roll in CASES:
        frames = G[kind + "_frames"]
        if kind == "flow":
            frames = frames[..., None]
        tf = GpuFrameTransform((ch, cw), mean, std, roll=roll, is_flow=(kind == "flow"), device=backend.device)
        got = tf.oversample(backend.put(torch.from_numpy(frames)))
        assert np.array_equal(got.cpu().numpy(), G[key]), key


def _tail(u8, mean, std, roll):
    """Stack(roll) -> ToTorchFormatTensor(div=False) -> GroupNormalize of the reference's uint8 output images (numpy)."""
    if u8.ndim == 3:
        u8 = u8[..., None]
    x = u8[..., ::-1] if (roll and u8.shape[-1] == 3) else u8
    x = np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)       # [n_img, C, H, W]
    n, c = x.shape[:2]
    flat = x.reshape(n * c, *x.shape[2:])
    m = np.array((list(mean) * (n * c))[:n * c], np.float32).reshape(-1, 1, 1)
    s = np.array((list(std) * (n * c))[:n * c], np.float32).reshape(-1, 1, 1)
    return (flat - m) / s


def test_gpu_training_augmentation_is_bit_exact_with_pil(backend):
    """GroupMultiScaleCrop (crop + PIL bilinear resize) + GroupRandomHorizontalFlip + the tail of the training chain as
    ssn_frames_crop_resize_normalize, against what the REFERENCE's own classes produced with the same ``random`` seeds: same crop
    boxes (index-exact), same pixels (bit-exact: the kernel restates Pillow's fixed-point resampler), at the fixture's small frames
    (RGB, flow with the inverted x component) and at the real sizes (256 x 340 -> 224: 5-tap shrink, 3-tap enlarge, copy)."""
    from action_detection_amd.input_pipeline import GpuTrainAugment
    for tag, scales, is_flow, mean in (("rgb", [1, .875, .75, .66], False, [104, 117, 128]), ("flow", [1, .875, .75], True, [128])):
        frames = G["aug_%s_frames" % tag]
        if is_flow:
            frames = frames[..., None]
        aug = GpuTrainAugment(56, mean, [1], scales, roll=True, is_flow=is_flow, device=backend.device)
        for seed in range(6):
            random.seed(seed)
            got = aug(backend.put(torch.from_numpy(frames)), group_size=frames.shape[0]).cpu().numpy()
            want = _tail(G["aug_%s_%d" % (tag, seed)], mean, [1], True)
            assert np.array_equal(got, want), (tag, seed, np.abs(got - want).max())
    aug = GpuTrainAugment(224, [104, 117, 128], [1], [1, .875, .75, .66], device=backend.device)
    full = torch.from_numpy(G["aug_full_frames"])
    for (cw, ch, ow, oh), seed in zip(G["aug_full_boxes"], (0, 1, 3, 4, 7)):
        random.seed(100 + seed)
        boxes, flips = aug.sample((340, 256), 1)
        assert boxes[0] == (ow, oh, cw, ch), (seed, boxes)
        got = aug(backend.put(full), group_size=1, boxes=boxes, flips=flips).cpu().numpy()
        want = _tail(G["aug_full_%d" % seed], [104, 117, 128], [1], True)
        assert np.array_equal(got, want), (seed, (cw, ch), np.abs(got - want).max(), (got != want).mean())
    # several groups with their own boxes in one call == the groups one by one
    random.seed(5)
    two = torch.from_numpy(np.concatenate([G["aug_full_frames"]] * 4))
    b, f = aug.sample((340, 256), 2)
    both = aug(backend.put(two), group_size=2, boxes=b, flips=f).cpu().numpy().reshape(4, 3, 224, 224)
    for g in range(2):
        one = aug(backend.put(two[2 * g:2 * g + 2]), group_size=2, boxes=[b[g]], flips=[f[g]]).cpu().numpy().reshape(2, 3, 224, 224)
        assert np.array_equal(both[2 * g:2 * g + 2], one)
    with np.testing.assert_raises(ValueError):
        aug(backend.put(full), group_size=1, boxes=[(200, 0, 224, 224)], flips=[False])       # leaves the 340-wide frame
