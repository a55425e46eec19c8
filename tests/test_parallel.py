"""CPU tier: the N>1 path with world_size-2 gloo processes.

Checks (1) the bucketed tail-first range protocol of GradReducer averages a flat gradient buffer
exactly, (2) whole-video sharding + the global completeness denominator reproduce the single-process
loss and gradients of the reference semantics (losses on the gathered batch).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import action_detection_amd  # noqa: F401
    import ssn_oracle as O
    from action_detection_amd.parallel import GradReducer, shard_videos

    class FakeBackbone(torch.nn.Module):
        grad_ready_hook = None

        def __init__(self):
            super().__init__()
            self.conv1_bn = torch.nn.BatchNorm2d(4)      # bn_mode 'partial': its gamma / beta have gradients of their own
            self.conv2_bn = torch.nn.BatchNorm2d(4)      # frozen: never touched

        def _train_bn_ids(self):
            return ["conv1"]

    class FakeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.base_model = FakeBackbone()
            self.activity_fc = torch.nn.Linear(8, 3)
            self.completeness_fc = torch.nn.Linear(24, 2)
            self.regressor_fc = None

    torch.manual_seed(0)
    model = FakeModel()
    red = GradReducer(model, min_bucket_elems=1000)
    # (1) tail-first contiguous ranges, coalesced into buckets, averaged
    total = 5000
    flat = torch.arange(total, dtype=torch.float32) * (rank + 1)
    for s, e in ((4200, 5000), (3900, 4200), (2500, 3900), (400, 2500), (0, 400)):
        model.base_model.grad_ready_hook.range_ready(flat, s, e)
    model.base_model.grad_ready_hook.finish()
    want = torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok1 = torch.allclose(flat, want)
    buckets = list(red.launched)
    # heads
    for p in model.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    red.reduce_heads()
    frozen = [model.base_model.conv2_bn.weight, model.base_model.conv2_bn.bias]
    ok2 = all(torch.allclose(p.grad, torch.full_like(p, (world + 1) / 2.0 if not any(p is q for q in frozen) else float(rank + 1)))
              for p in model.parameters())      # heads + the training-mode BatchNorm's gamma / beta averaged, nothing else touched

    # (2) loss semantics: shard 4 videos over 2 ranks vs the gathered batch
    rng = np.random.RandomState(3)
    v, c = 4, 20
    pred = torch.from_numpy((rng.standard_normal((7 * v, c)) * 1.5).astype(np.float32))
    labels = rng.randint(1, c + 1, size=7 * v)
    full_loss, full_grad = O.completeness_loss(pred, labels, 1, 7)
    lo, hi = shard_videos(v, rank, world)
    my_pred, my_lab = pred[7 * lo:7 * hi], labels[7 * lo:7 * hi]
    # per-rank: numerators local, denominator global (what CompletenessLoss(global_rows=) implements)
    l_loc, g_loc = O.completeness_loss(my_pred, my_lab, 1, 7)
    den_loc = (hi - lo) * 1 + int((hi - lo) * 6 * 0.17)
    den_glob = v * 1 + int(v * 6 * 0.17)
    contrib = torch.tensor([float(l_loc) * den_loc / den_glob])
    dist.all_reduce(contrib)
    g_scaled = torch.from_numpy(g_loc * den_loc / den_glob)
    ok3 = abs(contrib.item() - float(full_loss)) < 1e-6 and np.allclose(g_scaled.numpy(), full_grad[7 * lo:7 * hi])
    ret[rank] = (ok1, ok2, ok3, buckets)
    dist.destroy_process_group()


def test_two_process_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        ok1, ok2, ok3, buckets = ret[r]
        assert ok1, "flat-range all-reduce mismatch"
        assert ok2, "head gradient averaging mismatch"
        assert ok3, "sharded completeness loss does not reproduce the gathered-batch loss"
        assert buckets == [(3900, 5000), (2500, 3900), (400, 2500), (0, 400)], buckets


def test_shard_videos():
    from action_detection_amd.parallel import shard_videos
    assert [shard_videos(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(ValueError):
        shard_videos(10, 0, 4)
