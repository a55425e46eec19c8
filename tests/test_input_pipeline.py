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
