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
