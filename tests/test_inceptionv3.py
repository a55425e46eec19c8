"""Inception-v3 backbone (BASELINE.json configs[4], forward only): product executor vs the oracle's block-by-block
restatement -- through the host emulator at a reduced input size (CPU tier) and at 299x299 on the MI355X (-m gpu,
SSN test_forward + the dense-testing loop)."""
import numpy as np
import pytest
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic
from test_kernels import rel_err


def pair(input_size):
    from action_detection_amd.inceptionv3 import InceptionV3
    torch.manual_seed(0)
    prod = InceptionV3(num_classes=10, input_size=input_size)
    init_backbone_synthetic(prod)
    orc = O.OracleInceptionV3(num_classes=10)
    orc.load_state_dict(prod.state_dict())
    return prod.eval(), orc.eval()


def test_inceptionv3_features_emulated(emu):
    """75x75 is the smallest input the topology accepts (1x1 at the last stage): every layer shape class (3x3 s2 p0,
    5x5, 1x7, 7x1, 1x3, 3x1, pools, slices of the block outputs) runs through the emulated kernels."""
    prod, orc = pair(75)
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (1, 3, 75, 75), generator=g).float() - 110.0
    with torch.no_grad():
        feat = prod.features(x)
        ref = orc.features(x)
    assert feat.shape == (1, 2048)
    assert rel_err(feat, ref) < 1e-4
    with pytest.raises(NotImplementedError):
        prod.features(x.requires_grad_())


@pytest.mark.gpu
def test_inceptionv3_ssn_test_forward_and_dense_loop(hip_library):
    from action_detection_amd.dense_test import DenseTester
    from action_detection_amd.ssn_models import SSN
    num_class = 100
    torch.manual_seed(0)
    net = SSN(num_class, 2, 5, 2, "RGB", base_model="InceptionV3", test_mode=True, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.05)
    oracle = O.OracleSSN(num_class, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1), base_model="InceptionV3")
    oracle.load_state_dict(net.state_dict())
    net.prepare_test_fc()
    oracle.prepare_test_fc()
    net.to("cuda:0").eval()
    oracle.eval()
    assert net.test_fc.out_features == 1001
    g = torch.Generator().manual_seed(4)
    x = torch.randint(0, 256, (6, 3, 299, 299), generator=g).float() - 110.0
    with torch.no_grad():
        sc, base = net(x.cuda(), None, None, None, None)
        r_sc, r_base = oracle(x, None, None, None, None)
    assert base.shape == (6, 2048)
    assert rel_err(base, r_base) < 1e-4
    assert rel_err(sc, r_sc) < 1e-4
    # the per-video loop on 3 ticks x 2 crops
    batches = [x.view(3, 2, 3, 299, 299).transpose(0, 1).reshape(-1, 299, 299)]
    ticks = np.array([[0, 1, 2, 3], [0, 0, 3, 3]], dtype=np.int64)
    scaling = np.array([[1.0, 0.5], [0.0, 0.0]])
    tester = DenseTester(net, num_class, stats=np.array([[0.1, -0.3], [1.5, 0.7]]), tick_batch=2)
    act, comp, reg, out = tester.score_video(iter(batches), 3, torch.from_numpy(ticks), torch.from_numpy(scaling), num_crop=2)
    r = O.dense_test_video(oracle, iter(batches), 3, ticks, scaling, num_class, num_crop=2,
                           stats=np.array([[0.1, -0.3], [1.5, 0.7]]))
    for a, b in zip((act, comp, reg, out), r):
        assert rel_err(a, torch.from_numpy(b)) < 1e-4
