"""GPU input transforms (csrc/frames.hip) against the oracle's restatement of the reference chain
GroupOverSample -> Stack(roll) -> ToTorchFormatTensor(div=False) -> GroupNormalize: bit-exact (uint8 -> fp32)."""
import numpy as np
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd.input_pipeline import GpuFrameTransform, fill_fix_offset


def test_oversample_rgb_and_flow_bit_exact(backend):
    rs = np.random.RandomState(3)
    dev = backend.device
    # RGB, BNInception conventions: 256x340 frames -> 224 crops, BGR roll, mean [104, 117, 128], std [1]
    frames = rs.randint(0, 256, size=(3, 32, 43, 3)).astype(np.uint8)
    tf = GpuFrameTransform(24, [104, 117, 128], [1], roll=True, is_flow=False, device=dev)
    got = tf.oversample(backend.put(torch.from_numpy(frames)))
    ref = O.oversample_transform([f for f in frames], 24, 24, [104, 117, 128], [1], True, False)
    assert got.shape == ref.shape == (10 * 3 * 3, 24, 24)
    assert torch.equal(got.cpu(), ref)
    # no roll, non-trivial std (resnet-style statistics)
    tf = GpuFrameTransform((20, 28), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], roll=False, device=dev)
    got = tf.oversample(backend.put(torch.from_numpy(frames)))
    ref = O.oversample_transform([f for f in frames], 28, 20, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], False, False)
    assert torch.equal(got.cpu(), ref)
    # flow: 'L' images, x component (even index) inverted in the flipped crops, mean [128]
    flow = rs.randint(0, 256, size=(10, 30, 40, 1)).astype(np.uint8)
    tf = GpuFrameTransform(22, [128], [1], roll=True, is_flow=True, device=dev)
    got = tf.oversample(backend.put(torch.from_numpy(flow)))
    ref = O.oversample_transform([f[:, :, 0] for f in flow], 22, 22, [128], [1], True, True)
    assert torch.equal(got.cpu(), ref)
    # single crop + flip (tail of the training chain) == the matching slice of the oversampled group
    offs = fill_fix_offset(False, 43, 32, 24, 24)
    tf = GpuFrameTransform(24, [104, 117, 128], [1], roll=True, device=dev)
    one = tf.crop(backend.put(torch.from_numpy(frames)), offs[3][0], offs[3][1], True)
    allc = tf.oversample(backend.put(torch.from_numpy(frames))).reshape(10, 9, 24, 24)
    assert torch.equal(one.cpu(), allc[7].cpu())


def test_training_batch_prefetcher(backend):
    """uint8 frames -> the five arguments of SSN.forward, batch after batch, equal to the synchronous transform chain (on the
    GPU: staged through pinned memory and a side stream while the consumer works on the previous batch)."""
    from action_detection_amd.input_pipeline import TrainingBatchPrefetcher
    rs = np.random.RandomState(11)
    dev = backend.device
    tf = GpuFrameTransform(16, [104, 117, 128], [1], roll=True, device=dev)
    items = []
    for _ in range(5):
        frames = rs.randint(0, 256, size=(2, 6, 16, 16, 3)).astype(np.uint8)
        items.append((frames, rs.rand(2, 8, 2).astype(np.float32), rs.randint(0, 5, (2, 8)), rs.randn(2, 8, 2).astype(np.float32),
                      np.tile(np.array([0, 1, 1, 1, 1, 1, 1, 2]), (2, 1))))
    seen = 0
    for got, want in zip(TrainingBatchPrefetcher(iter(items), tf, depth=2), items):
        inp, scaling, target, reg_target, prop_type = got
        ref = O.oversample_transform([f for f in want[0].reshape(12, 16, 16, 3)], 16, 16, [104, 117, 128], [1], True, False)
        ref = ref.reshape(10, 12 * 3, 16, 16)[4].reshape(2, 18, 16, 16)      # crop 4 = the centre crop = the whole 16x16 frame
        assert inp.shape == (2, 18, 16, 16) and torch.equal(inp.cpu(), ref)
        assert torch.equal(scaling.cpu(), torch.from_numpy(want[1])) and torch.equal(target.cpu(), torch.from_numpy(want[2]))
        assert torch.equal(prop_type.cpu(), torch.from_numpy(want[4])) and str(inp.device).startswith(str(dev)[:4])
        seen += 1
    assert seen == 5

    def broken():
        yield items[0]
        raise RuntimeError("decoder died")
    it = TrainingBatchPrefetcher(broken(), tf)
    next(it)
    import pytest
    with pytest.raises(RuntimeError):
        next(it)


def test_training_batch_prefetcher_with_gpu_augmentation(backend):
    """Un-cropped decoded frames in, network input out: the prefetcher runs SSN.get_augmentation() (scale-jittered crop, PIL's
    bilinear resize, flip) + the normalisation tail on the device, one box / flip per proposal group drawn through `random` in the
    reference's order -- equal to the host chain of action_detection_amd.transforms (itself pinned to the reference's classes,
    tests/test_transforms.py) with the same seed."""
    import random
    from PIL import Image
    from action_detection_amd import transforms as T
    from action_detection_amd.input_pipeline import GpuTrainAugment, TrainingBatchPrefetcher
    rs = np.random.RandomState(12)
    dev = backend.device
    aug = GpuTrainAugment(32, [104, 117, 128], [1], [1, .875, .75, .66], roll=True, device=dev)
    items = []
    for _ in range(3):
        frames = rs.randint(0, 256, size=(2, 6, 40, 52, 3)).astype(np.uint8)        # 2 videos x (2 proposals x 3 frames)
        items.append((frames, rs.rand(2, 2, 2).astype(np.float32), rs.randint(0, 5, (2, 2)), rs.randn(2, 2, 2).astype(np.float32),
                      np.tile(np.array([0, 2]), (2, 1))))
    random.seed(77)
    got = [b[0].cpu() for b in TrainingBatchPrefetcher(iter(items), aug, depth=2, group_size=3)]
    random.seed(77)
    host = T.Compose([T.GroupMultiScaleCrop(32, [1, .875, .75, .66]), T.GroupRandomHorizontalFlip(is_flow=False)])
    tail = T.Compose([T.Stack(roll=True), T.ToTorchFormatTensor(div=False), T.GroupNormalize([104, 117, 128], [1])])
    for g, item in zip(got, items):
        frames = item[0].reshape(4, 3, 40, 52, 3)                                   # 4 proposal groups per batch, in order
        want = torch.cat([tail(host([Image.fromarray(f, "RGB") for f in grp])) for grp in frames])
        assert g.shape == (2, 18, 32, 32) and torch.equal(g.reshape(-1, 32, 32), want)
    import pytest
    with pytest.raises(ValueError):
        TrainingBatchPrefetcher(iter(items), aug)                                  # group_size is required
