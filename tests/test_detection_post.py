"""Detection post-processing (csrc/detect.hip) against the oracle's restatement of eval_detection_results.py +
ops/utils.temporal_nms: same kept boxes per class in the same order, fused scores / regressed spans to 1e-6."""
import numpy as np
import torch

import action_detection_amd  # noqa: F401
import ssn_oracle as O
from action_detection_amd.detection_post import DetectionPostProcessor


def synthetic_video(rs, p, c):
    start = rs.uniform(0, 0.8, p)
    length = rs.uniform(0.02, 0.5, p)
    rel = np.stack([start, np.minimum(start + length, 1.0)], axis=1)            # float64, as in the pickles
    act = rs.standard_normal((p, c + 1)).astype(np.float32) * 2
    comp = rs.standard_normal((p, c)).astype(np.float32)
    reg = (rs.standard_normal((p, c, 2)) * 0.3).astype(np.float32)
    return rel, act, comp, reg


def check(backend, p, c, top_k, thr, no_reg, seed, with_reg=True):
    rs = np.random.RandomState(seed)
    rel, act, comp, reg = synthetic_video(rs, p, c)
    ref, ref_comb = O.detections_for_video(rel[None], act, comp, reg if with_reg else None, c, thr, top_k, no_reg)
    post = DetectionPostProcessor(c, thr, top_k, no_reg)
    got, comb = post.process_video(torch.from_numpy(rel[None]), backend.put(torch.from_numpy(act)),
                                   backend.put(torch.from_numpy(comp)),
                                   backend.put(torch.from_numpy(reg)) if with_reg else None, device=backend.device)
    assert np.allclose(comb.cpu().numpy(), ref_comb, rtol=2e-6, atol=1e-30)
    assert sorted(got) == sorted(ref), (sorted(got), sorted(ref))
    for cls in ref:
        assert got[cls].shape == ref[cls].shape, (cls, got[cls].shape, ref[cls].shape)
        assert np.allclose(got[cls], ref[cls], rtol=2e-6, atol=1e-9), cls


def test_detections_match_reference_logic(backend):
    if backend.is_gpu:
        # THUMOS14-style (top_k 2000 over all pairs, NMS 0.2), ActivityNet-style (top_k 60, NMS 0.6), all-pairs branch
        check(backend, 700, 20, 2000, 0.2, False, 1)
        check(backend, 187, 100, 60, 0.6, False, 2)
        check(backend, 300, 20, 0, 0.4, False, 3)
        check(backend, 1500, 3, 0, 0.7, False, 7)
    # (the emulator runs every barrier of the sort / suppression loops through fibers: small cases only)
    check(backend, 24, 6, 50, 0.2, False, 1)
    check(backend, 20, 4, 0, 0.4, False, 3)
    check(backend, 16, 5, 10 ** 6, 0.3, True, 4)          # top_k larger than the number of pairs, no regression
    check(backend, 12, 3, 20, 0.5, False, 5, with_reg=False)
    check(backend, 1, 3, 2, 0.5, False, 6)
