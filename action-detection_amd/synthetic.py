"""Seeded synthetic weights and THUMOS14-shape batches.

There is no network for checkpoints or datasets, so parity tests, the oracle and the
benchmark all draw weights/inputs from here.  numpy ``RandomState`` is used (not the torch
generator) so the streams are identical on every box regardless of torch build.

Input conventions follow what ``SSNDataSet.get_training_data`` emits after the reference's
transform stack (/root/reference/ssn_dataset.py:455-490, /root/reference/ssn_train.py:106-111):
pixels 0..255 minus the per-channel mean, std 1, proposals ordered [fg, incomplete x6, bg]
per video (/root/reference/ssn_dataset.py:181-183,273-276).
"""
import numpy as np
import torch
from torch import nn


PIXEL_STD = 74.0  # std of (uniform 0..255 pixel - mean): the first conv is scaled for this input range


def init_backbone_synthetic(base_model, seed=1234, negative_gamma_frac=0.0):
    """He-normal conv weights, small biases, non-trivial frozen-BN statistics.

    A real BN-Inception checkpoint absorbs the 0..255 pixel scale in its first conv / BN; the
    synthetic one does the same by dividing the first conv's He-normal weights by PIXEL_STD, so
    activations stay O(1) through all 69 layers and a few SGD steps at the reference's default
    lr=0.001 remain finite.  ``negative_gamma_frac``: fraction of BN scales whose sign is flipped (trained
    checkpoints do contain negative gammas; the fused ReLU/BN backward must keep their sign).
    """
    rng = np.random.RandomState(seed)
    mods = sorted(((n, m) for n, m in base_model.named_modules()), key=lambda t: t[0])
    with torch.no_grad():
        for name, m in mods:
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                w = rng.standard_normal(m.weight.shape).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_in))
                if name.startswith("conv1_") or name.startswith("conv_1a_"):   # the layer that sees 0..255 pixels
                    w = w / np.float32(PIXEL_STD)
                m.weight.copy_(torch.from_numpy(w))
                if m.bias is not None:
                    m.bias.copy_(torch.from_numpy((rng.standard_normal(m.bias.shape) * 0.01).astype(np.float32)))
            elif isinstance(m, nn.BatchNorm2d):
                c = m.num_features
                gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
                if negative_gamma_frac > 0:
                    flip = np.random.RandomState(seed + 7 + c).uniform(0, 1, c) < negative_gamma_frac
                    gamma = np.where(flip, -gamma, gamma).astype(np.float32)
                m.weight.copy_(torch.from_numpy(gamma))
                m.bias.copy_(torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32)))
                m.running_mean.copy_(torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32)))
                m.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
    return base_model


def init_heads_synthetic(model, std=0.05, seed=4321):
    """Head weights with enough spread that hinge / CE / smooth-L1 branches are all exercised."""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name in ("activity_fc", "completeness_fc", "regressor_fc"):
            fc = getattr(model, name, None)
            if fc is None:
                continue
            fc.weight.copy_(torch.from_numpy((rng.standard_normal(fc.weight.shape) * std).astype(np.float32)))
            fc.bias.copy_(torch.from_numpy((rng.standard_normal(fc.bias.shape) * std).astype(np.float32)))
    return model


def make_batch(num_videos, modality="RGB", num_class=20, seed=0, input_size=224,
               num_segments=9, prop_per_video=8, new_length=None):
    """Return (input, aug_scaling, target, reg_target, prop_type) as CPU tensors."""
    rng = np.random.RandomState(seed)
    if new_length is None:
        new_length = 1 if modality == "RGB" else 5
    c = 3 * new_length if modality == "RGB" else (3 * (new_length + 1) if modality == "RGBDiff" else 2 * new_length)
    v, p, s = num_videos, prop_per_video, num_segments
    pix = rng.randint(0, 256, size=(v, p * s, c, input_size, input_size)).astype(np.float32)
    if modality == "RGB":
        mean = np.array([104, 117, 128] * new_length, np.float32)
    elif modality == "RGBDiff":      # ssn_models.py:128-129: the RGB means repeated over the new_length + 1 stacked frames
        mean = np.array([104, 117, 128] * (new_length + 1), np.float32)
    else:
        mean = np.full((c,), 128, np.float32)
    pix -= mean.reshape(1, 1, c, 1, 1)
    inp = torch.from_numpy(pix.reshape(v, p * s * c, input_size, input_size))
    scaling = torch.from_numpy(rng.uniform(0, 1, size=(v, p, 2)).astype(np.float32))
    prop_type = torch.tensor([[0] + [1] * (p - 2) + [2]] * v, dtype=torch.int64)
    target = torch.from_numpy(rng.randint(1, num_class + 1, size=(v, p)).astype(np.int64))
    target[:, -1] = 0
    reg = np.zeros((v, p, 2), np.float32)
    reg[:, 0, :] = rng.standard_normal((v, 2)).astype(np.float32)
    reg_target = torch.from_numpy(reg)
    return inp, scaling, target, reg_target, prop_type
