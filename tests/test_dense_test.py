"""Dense testing (ssn_test.py:66-92): crop mean / regression de-normalisation kernels on every backend, and the
whole per-video loop of the product (DenseTester) against the oracle's restatement of the reference loop (-m gpu)."""
import numpy as np
import pytest
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd import kernels as K
from test_kernels import rel_err


def test_crop_mean_and_reg_denorm(backend):
    g = torch.Generator().manual_seed(5)
    for crops, t, d in ((10, 7, 1024), (1, 3, 33), (10, 4, 201), (3, 1, 5)):
        x = torch.randn(crops * t, d, generator=g)
        out = backend.put(torch.empty(t, d))
        K.crop_mean(backend.put(x), crops, out)
        ref = x.view(crops, t, d).mean(dim=0)
        assert rel_err(out, ref) < 1e-6, (crops, t, d)
    reg = torch.randn(9, 20, 2, generator=g)
    dev = backend.put(reg.clone())
    K.reg_denorm(dev, 0.3, 1.7, -0.2, 0.9)
    ref = reg.clone()
    ref[:, :, 0] = ref[:, :, 0] * 1.7 + 0.3
    ref[:, :, 1] = ref[:, :, 1] * 0.9 + (-0.2)
    assert rel_err(dev, ref) < 1e-6


def synthetic_video(n_ticks, num_crop, length, gen_batch, seed, size=224):
    """Frame batches in the reference's layout (ssn_dataset.py:433-452 + GroupOverSample): crop-major [crop][tick]."""
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (n_ticks, num_crop, length, size, size), generator=g).float() - 110.0
    batches = []
    for t0 in range(0, n_ticks, gen_batch):
        b = frames[t0:t0 + gen_batch]                              # [b, crop, length, H, W]
        batches.append(b.transpose(0, 1).reshape(-1, size, size))  # crop-major, channel-stacked like Stack()
    return batches


@pytest.mark.gpu
@pytest.mark.parametrize("modality,num_crop,tick_batch", [("RGB", 10, 5), ("Flow", 1, 3)])
def test_dense_tester_matches_reference_loop(hip_library, modality, num_crop, tick_batch):
    from action_detection_amd.dense_test import DenseTester
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic
    num_class, n_ticks = 20, 11
    torch.manual_seed(0)
    net = SSN(num_class, 2, 5, 2, modality, test_mode=True, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.05)
    oracle = O.OracleSSN(num_class, 2, 5, 2, modality, test_mode=True, stpp_cfg=(1, 1, 1))
    oracle.load_state_dict(net.state_dict())
    net.prepare_test_fc()
    oracle.prepare_test_fc()
    net.to("cuda:0").eval()
    oracle.eval()
    length = 3 if modality == "RGB" else 10
    batches = synthetic_video(n_ticks, num_crop, length, 4, seed=3)
    # proposals: inside, touching both ends, reaching outside (skipped stages), single tick
    ticks = np.array([[1, 3, 7, 9], [0, 0, 11, 11], [0, 2, 4, 11], [5, 5, 5, 6], [-2, 0, 3, 5]], dtype=np.int64)
    ticks = np.clip(ticks, 0, n_ticks)
    scaling = np.array([[1.0, 1.0], [0.0, 0.0], [0.4, 1.0], [1.0, 0.25], [0.5, 0.5]])
    stats = np.array([[0.1, -0.3], [1.5, 0.7]])
    tester = DenseTester(net, num_class, stpp_cfg=(1, 1, 1), stats=stats, tick_batch=tick_batch)
    act, comp, reg, output = tester.score_video(iter(batches), n_ticks, torch.from_numpy(ticks),
                                                torch.from_numpy(scaling), num_crop=num_crop)
    r_act, r_comp, r_reg, r_out = O.dense_test_video(oracle, iter(batches), n_ticks, ticks, scaling, num_class,
                                                     num_crop=num_crop, stats=stats)
    assert rel_err(output, torch.from_numpy(r_out)) < 1e-4
    assert rel_err(act, torch.from_numpy(r_act)) < 1e-4
    assert rel_err(comp, torch.from_numpy(r_comp)) < 1e-4
    assert rel_err(reg, torch.from_numpy(r_reg)) < 1e-4


def test_empty_inputs(backend):
    """Zero proposals / zero ticks / zero pairs: every entry point of the test path returns empty results, no launch."""
    from action_detection_amd.detection_post import DetectionPostProcessor
    from action_detection_amd.ops.ssn_ops import STPPReorgainzed
    dev = backend.device
    out = backend.put(torch.empty(0, 16))
    K.crop_mean(backend.put(torch.empty(0, 16)), 10, out)
    reorg = STPPReorgainzed(21 + 20 * 3 + 40 * 3, 21, 20, 40, True, stpp_cfg=(1, 1, 1))
    scores = backend.put(torch.randn(9, 201))
    act, comp, reg = reorg.forward(scores, torch.zeros((0, 4), dtype=torch.int64), torch.zeros((0, 2)))
    assert act.shape == (0, 21) and comp.shape == (0, 20) and reg.shape == (0, 40)
    dets, comb = DetectionPostProcessor(20, 0.2, 2000).process_video(
        torch.zeros((0, 2), dtype=torch.float64), backend.put(torch.zeros(0, 21)), backend.put(torch.zeros(0, 20)),
        backend.put(torch.zeros(0, 20, 2)), device=dev)
    assert dets == {} and comb.shape == (0, 20)
    K.reg_denorm(backend.put(torch.empty(0, 20, 2)), 0.0, 1.0, 0.0, 1.0)


@pytest.mark.gpu
def test_dense_tester_repeats_the_calls_that_left_their_range(hip_library):
    """ssn_test.py:78-92 on a video whose tick batches alternate dark (x 1/40) and bright (x 8) frames: every activation of the
    backbone moves 320x up and down between consecutive calls.  The tester polls the range guard once per video and repeats
    exactly the calls that left the range of their delayed scales; the scores of every tick must match the oracle's."""
    from action_detection_amd.dense_test import DenseTester
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic
    num_class, n_ticks, num_crop = 20, 12, 2
    torch.manual_seed(0)
    net = SSN(num_class, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.05)
    oracle = O.OracleSSN(num_class, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1))
    oracle.load_state_dict(net.state_dict())
    net.prepare_test_fc()
    oracle.prepare_test_fc()
    net.to("cuda:0").eval()
    oracle.eval()
    batches = synthetic_video(n_ticks, num_crop, 3, 3, seed=8)
    batches = [b * k for b, k in zip(batches, (1.0, 1.0 / 40.0, 8.0, 1.0 / 40.0))]
    ticks = np.array([[1, 3, 7, 9], [0, 0, 12, 12]], dtype=np.int64)
    scaling = np.array([[1.0, 1.0], [0.3, 0.6]])
    tester = DenseTester(net, num_class, stpp_cfg=(1, 1, 1), stats=None, tick_batch=3)
    act, comp, reg, output = tester.score_video(iter(batches), n_ticks, torch.from_numpy(ticks), torch.from_numpy(scaling),
                                                num_crop=num_crop)
    r_act, r_comp, r_reg, r_out = O.dense_test_video(oracle, iter(batches), n_ticks, ticks, scaling, num_class, num_crop=num_crop)
    assert tester.repeated_calls >= 1 and not net.scale_fault()
    for t0 in range(0, n_ticks, 3):         # per call: a clamped batch would be off by far more than this
        assert rel_err(output[t0:t0 + 3], torch.from_numpy(r_out[t0:t0 + 3])) < 1e-4, t0
    assert rel_err(act, torch.from_numpy(r_act)) < 1e-4 and rel_err(comp, torch.from_numpy(r_comp)) < 1e-4
