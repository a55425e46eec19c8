"""The range guard of the planes path's delayed scales (planes_exec.py) on data that CHANGES between calls.

The reference's loop alternates different batches and train / eval passes (/root/reference/ssn_train.py:191-253, validate()
:278-362); its cuDNN path stores fp32, so magnitude changes cost it nothing.  The planes executor stores every activation /
gradient with the power-of-two scale the PREVIOUS call's maxima suggested (2 bits of head-room).  These tests run what the
round-3 tests never did: calibrate on batch A, then -- without touching any calibration switch -- another batch, 8 x A and 256 x A (overflows
the head-room), A / 64, a feature gradient 3000 x larger and 3e6 x smaller (drains the low plane), an evaluation forward of another batch
size between training steps; EVERY call must meet the parity bars (features / logits vs the oracle, every gradient tensor vs the
float64 referee that takes the forward's ReLU / max-pool decisions), the guard must have repeated the passes that needed it, and
no fault may be left behind.  Also: the deferred protocol of a hipGraph owner (fault word -> optimizer skips -> recalibrate ->
retry), weight reloads, and -- two ranks over gloo -- that a fault on ONE rank repeats the pass on BOTH (matched collectives,
bit-identical averaged gradients).

CPU tier: a five-layer backbone (tests/tiny_backbone.py) on the product's executor through the host emulator.  ``-m gpu``: the
real SSN at 224 x 224 against oracle/ssn_oracle.py.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import action_detection_amd  # noqa: F401
from action_detection_amd.optim import SSNSGD
from tiny_backbone import TinyBackbone, TinyRef, init_tiny

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


def _pair(dev):
    net = init_tiny(TinyBackbone()).to(dev).train()
    net.debug_keep_saved = True
    ref = TinyRef()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    return net, ref


def _check_call(net, ref, x, w, dev, tol_f=3e-6, tol_g=2e-5, train=True):
    """One forward (+ backward) of the product against the float64 referee forced onto the product's decisions."""
    net.zero_grad(set_to_none=True)
    ref.zero_grad(set_to_none=True)
    ref.forced = None
    if not train:
        with torch.no_grad():
            f = net.features(x.to(dev))
        assert rel_err(f, ref(x)) < 1e-5
        return
    f = net.features(x.to(dev))
    free = ref(x)
    assert rel_err(f, free) < 1e-5, ("features vs independent float64 forward", rel_err(f, free))
    relu, pool = net.export_decisions()
    ref.forced = ({k: t.cpu() for k, t in relu.items()}, {k: t.cpu() for k, t in pool.items()})
    fr = ref(x)
    assert rel_err(f, fr) < tol_f, ("features vs forced referee", rel_err(f, fr))
    (f * w.to(dev)).sum().backward()
    (fr * w.double()).sum().backward()
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        if p2.grad is None:
            assert p1.grad is None
            continue
        assert rel_err(p1.grad, p2.grad) < tol_g, (n1, rel_err(p1.grad, p2.grad))
    assert not net.scale_fault(), "the sync guard must not leave a fault behind"


def _data(n=3, size=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, size, size, generator=g) * 40.0, torch.randn(n, 32, generator=g)


def test_guard_repairs_every_call_on_changing_data(emu):
    dev = torch.device("cpu")
    net, ref = _pair(dev)
    a, w = _data(seed=1)
    b, wb = _data(seed=2)
    _check_call(net, ref, a, w, dev)                     # calibrates
    st = next(iter(net._planes_states.values()))
    assert st.fwd_calibrated and st.bwd_calibrated and net.guard_stats() == {"fwd": 0, "bwd": 0}
    _check_call(net, ref, b * 1.7, wb, dev)              # another batch, inside the head-room: nothing repeated
    assert net.guard_stats() == {"fwd": 0, "bwd": 0}
    _check_call(net, ref, a * 8.0, w, dev)               # 8 x A: at the edge of the head-room (2 - 3 bits, by the octave)
    _check_call(net, ref, a * 8.0 * 32.0, w, dev)        # another 32 x: beyond it in any case -> the forward is repeated
    s1 = net.guard_stats()
    assert s1["fwd"] >= 1
    e, we = _data(n=1, seed=3)
    _check_call(net, ref, e * 0.5, we, dev, train=False)  # an evaluation forward of ANOTHER batch size between training steps
    _check_call(net, ref, a / 64.0, w, dev)              # A / 64: absorbed by the format (no repeat needed, still exact)
    _check_call(net, ref, a / 4096.0 / 64.0, w, dev)     # (activations bottom out at their BatchNorm shifts: nothing drains)
    _check_call(net, ref, a, w * 3000.0, dev)            # same frames, feature gradient 3000 x larger -> backward repeated
    s2 = net.guard_stats()
    assert s2["bwd"] >= 1
    _check_call(net, ref, b, wb * 1e-3, dev)             # ... and 3e6 x smaller: the low plane would drain -> repeated (fault bit 1)
    s3 = net.guard_stats()
    assert s3["bwd"] > s2["bwd"]
    _check_call(net, ref, a, w, dev)
    # a checkpoint loaded after the first forward re-calibrates (magnitudes of every activation change)
    sd = {k: (v * 30.0 if k.endswith("conv1_3x3.weight") else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    ref.load_state_dict({k: v.double() for k, v in sd.items()})
    assert not st.fwd_calibrated and not st.bwd_calibrated
    _check_call(net, ref, a, w, dev)
    assert st.fwd_calibrated


def test_unguarded_pass_clamps_and_deferred_protocol_recovers(emu):
    """What the guard prevents, and the protocol of a caller that cannot poll inside its step (a hipGraph replay): the pass only
    LAUNCHES the check, the optimizer kernel skips the flagged step, the host sees the word afterwards, recalibrates, retries."""
    dev = torch.device("cpu")
    net, ref = _pair(dev)
    a, w = _data(seed=4)
    _check_call(net, ref, a, w, dev)
    net.scale_guard = "deferred"
    opt = SSNSGD([{"params": [p for p in net.parameters() if p.requires_grad], "lr_mult": 1, "decay_mult": 1, "name": "w"}],
                 lr=0.01, momentum=0.9, weight_decay=5e-4)
    flag = net.planes_flag(dev)

    def step(x):
        net.zero_grad(set_to_none=True)
        f = net.features(x.to(dev))
        (f * w.to(dev)).sum().backward()
        opt.step(skip_flag=flag)
        return f.detach().clone()

    before = [p.detach().clone() for p in net.parameters()]
    step(a)                                              # clean step: the update lands
    assert not net.scale_fault()
    assert any(not torch.equal(p, q) for p, q in zip(net.parameters(), before))
    before = [p.detach().clone() for p in net.parameters()]
    mom = [opt.state[p]["momentum_buffer"].clone() for p in net.parameters() if p in opt.state]
    f_bad = step(a * 64.0)                               # overflows: values were clamped ...
    ref.forced = None
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    assert rel_err(f_bad, ref(a * 64.0)) > 1e-3          # ... (this is the silent divergence round 3 shipped) ...
    assert net.scale_fault()                             # ... the device word says so ...
    assert all(torch.equal(p, q) for p, q in zip(net.parameters(), before)), "a flagged step must not move the weights"
    assert all(torch.equal(opt.state[p]["momentum_buffer"], m) for p, m in
               zip([p for p in net.parameters() if p in opt.state], mom)), "... nor the momentum"
    net.recalibrate()                                    # host side of the protocol: clear, recalibrate, retry the step
    f_ok = step(a * 64.0)
    assert not net.scale_fault()
    assert rel_err(f_ok, ref(a * 64.0)) < 1e-5
    assert any(not torch.equal(p, q) for p, q in zip(net.parameters(), before))
    # and with the guard off nothing is even launched: the word stays clear while results are wrong (why "off" is not a default)
    net.scale_guard = "off"
    step(a * 4096.0)
    assert not net.scale_fault()


# ---------------------------------------------------------------------------------------------- two ranks (gloo)
class _Wrap(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.base_model = net


def _rank_batches(rank):
    """step 0 calibrates both ranks on their own data; step 1: rank 0's frames are 16 x larger (fault on rank 0 only); step 2:
    rank 1's feature gradient is 2000 x larger (fault on rank 1 only, in the backward); step 3: quiet."""
    x, w = _data(n=2, seed=10 + rank)
    return [(x, w), (x * (16.0 if rank == 0 else 1.0), w), (x, w * (2000.0 if rank == 1 else 1.0)), (x * 0.7, w)]


def _worker(rank, world, port, out_dir, deferred):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from action_detection_amd import _lib
    _lib.use_library_for_testing(_lib.SsnLibrary(os.path.join(ROOT, "tests", "emu", "libssn_emu.so"), is_emulator=True))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from action_detection_amd.parallel import GradReducer
    dev = torch.device("cpu")
    net = init_tiny(TinyBackbone()).to(dev).train()
    red = GradReducer(_Wrap(net), min_bucket_elems=256, deferred=deferred)
    out = []
    for x, w in _rank_batches(rank):
        net.zero_grad(set_to_none=True)
        f = net.features(x)
        (f * w).sum().backward()
        if deferred:
            red.reduce_all(average=True)
        out.append({n: p.grad.detach().clone().numpy() for n, p in net.named_parameters() if p.grad is not None})
        assert not net.scale_fault()
    torch.save({"grads": out, "stats": net.guard_stats(), "launched": list(red.launched)}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("deferred", [False, True])
def test_fault_on_one_rank_repeats_the_pass_on_all_ranks(emu_library, tmp_path, deferred):
    """SURVEY 8(e) / ssn_train.py:67: gradients are averaged over the ranks.  The guard fires on ONE rank only (different data =
    different delayed scales); with the overlapping reducer the bucket all-reduces of the faulty pass are already in flight, so
    the repeat must be collective (GradReducer.agree) -- a mismatch would hang or mix passes.  Both ranks must end every step
    with bit-identical gradients that equal the float64 average of the per-rank referees."""
    world = 2
    port = 29300 + (os.getpid() % 500) + (7 if deferred else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), deferred), nprocs=world, join=True)
    ranks = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r), weights_only=False) for r in range(world)]
    assert ranks[0]["stats"]["fwd"] >= 1 and ranks[1]["stats"]["fwd"] == 0, [r["stats"] for r in ranks]
    assert ranks[1]["stats"]["bwd"] >= 1, [r["stats"] for r in ranks]
    ref = TinyRef()
    ref.load_state_dict({k: v.double() for k, v in init_tiny(TinyBackbone()).state_dict().items()})
    for step in range(4):
        g0, g1 = ranks[0]["grads"][step], ranks[1]["grads"][step]
        want = None
        for rank in range(world):
            x, w = _rank_batches(rank)[step]
            ref.zero_grad(set_to_none=True)
            (ref(x) * w.double()).sum().backward()
            g = {n: p.grad.clone() / world for n, p in ref.named_parameters() if p.grad is not None}
            want = g if want is None else {n: want[n] + g[n] for n in g}
        for n in want:
            assert np.array_equal(g0[n], g1[n]), (step, n, "ranks disagree")
            e = rel_err(torch.from_numpy(g0[n]), want[n])
            assert e < 1e-4, (step, n, e)


# ---------------------------------------------------------------------------------------------- the real model (-m gpu)
@pytest.mark.gpu
def test_real_ssn_parity_on_changing_data_gpu(hip_library):
    """BNInception SSN at 224 x 224 (2 videos = 144 frames per call so that six calls fit the GPU tier's budget): calibrate on batch
    A, then B != A, 8 x A, an eval() forward of a 1-video batch, A / 64, B again -- logits / losses <= 1e-4 against the fp32 CPU
    oracle and every gradient tensor <= 5e-5 against the mask-forced float64 referee on EVERY call, no calibration switch touched."""
    import ssn_oracle as O
    from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch
    torch.manual_seed(0)
    v = 2
    m = SSN(20, 2, 5, 2, "RGB", dropout=0.0, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(m.base_model)
    init_heads_synthetic(m)
    o = O.OracleSSN(20, 2, 5, 2, "RGB", dropout=0.0, stpp_cfg=(1, 1, 1))
    o.load_state_dict(m.state_dict())
    o64 = O.OracleSSN(20, 2, 5, 2, "RGB", dropout=0, stpp_cfg=(1, 1, 1)).double()
    o64.load_state_dict({k: t.double() for k, t in o.state_dict().items()})
    m.to("cuda:0").train()
    o.train()
    o64.train()
    m.base_model.debug_keep_saved = True
    crit = (ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss())

    def call(batch, what):
        m.zero_grad(set_to_none=True)
        o64.zero_grad(set_to_none=True)
        out = m(*[t.cuda() for t in batch])
        with torch.no_grad():
            ref = o(*batch)
        for i in (0, 2, 4):
            assert rel_err(out[i], ref[i]) < 1e-4, (what, i, rel_err(out[i], ref[i]))
        loss = crit[0](out[0], out[1]) + 0.1 * crit[1](out[2], out[3], 1, 7) + 0.1 * crit[2](out[4], out[5], out[6])
        rt = O.ssn_total_loss(ref, v)[0]
        assert abs(loss.item() - rt.item()) <= 1e-4 * abs(rt.item()) + 1e-7, (what, loss.item(), rt.item())
        loss.backward()
        relu, pools = m.base_model.export_decisions()
        o64.base_model.forced = ({k: t.cpu() for k, t in relu.items()}, [t.cpu() for t in pools.values()])
        b64 = [t.double() if t.is_floating_point() else t for t in batch]
        out_m = o64(*b64)
        for i in (0, 2, 4):
            assert rel_err(out[i], out_m[i]) < 2e-6, (what, "forced forward", i, rel_err(out[i], out_m[i]))
        O.ssn_total_loss(out_m, v)[0].backward()
        worst = ("", 0.0)
        for (n1, p1), (n2, p2) in zip(m.named_parameters(), o64.named_parameters()):
            if p2.grad is None:
                continue
            e = rel_err(p1.grad, p2.grad)
            worst = (n1, e) if e > worst[1] else worst
        print("%-24s worst gradient vs mask-forced float64 referee: %s %.2e   guard %s" % (what, worst[0], worst[1],
                                                                                      m.base_model.guard_stats()))
        assert worst[1] < 5e-5, (what, worst)
        assert not m.scale_fault()

    def scaled(batch, k):
        return [batch[0] * k] + list(batch[1:])

    a = make_batch(v, "RGB", 20, seed=21)
    b = make_batch(v, "RGB", 20, seed=22)
    call(a, "A (calibrates)")
    assert m.base_model.guard_stats() == {"fwd": 0, "bwd": 0}
    call(b, "B != A")
    call(scaled(a, 8.0), "8 x A")                      # the edge of the head-room (2 - 3 bits, by where the maximum sits in its octave)
    call(scaled(a, 8.0 * 32.0), "256 x A")             # beyond it in any case
    assert m.base_model.guard_stats()["fwd"] >= 1, "a 32 x jump must overflow the head-room and be repeated"
    m.eval()
    o.eval()
    e1 = make_batch(1, "RGB", 20, seed=23)
    with torch.no_grad():
        oe = m(*[t.cuda() for t in e1])
        re_ = o(*e1)
    for i in (0, 2, 4):
        assert rel_err(oe[i], re_[i]) < 1e-4, ("eval forward between training steps", i)
    m.train()
    o.train()
    call(scaled(a, 1.0 / 64.0), "A / 64")
    call(b, "B again")


@pytest.mark.gpu
def test_dense_tester_on_dark_and_bright_tick_batches_gpu(hip_library):
    """ssn_test.py:78-92: the tick batches of one video go through the backbone one after the other.  A video whose batches
    alternate dark (x 1/30) and bright (x 6) frames moves every activation's magnitude 180 x up and down between consecutive
    calls; every batch's scores must match the oracle's."""
    import ssn_oracle as O
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic
    torch.manual_seed(0)
    net = SSN(20, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.01)
    o = O.OracleSSN(20, 2, 5, 2, "RGB", test_mode=True, stpp_cfg=(1, 1, 1))
    o.load_state_dict({k: t.cpu() for k, t in net.state_dict().items()})
    net.prepare_test_fc()
    o.prepare_test_fc()
    net.to("cuda:0").eval()
    o.eval()
    g = torch.Generator().manual_seed(5)
    base = torch.randint(0, 256, (8, 3, 224, 224), generator=g).float() - 110.0
    for k in (1.0, 1.0 / 30.0, 6.0, 1.0 / 30.0, 6.0, 1.0):
        x = base * k
        with torch.no_grad():
            s, f = net(x.cuda())
            rs, rf = o(x)
        assert rel_err(f, rf) < 1e-4 and rel_err(s, rs) < 1e-4, (k, rel_err(f, rf), rel_err(s, rs))
        assert not net.scale_fault()
    assert net.base_model.guard_stats()["fwd"] >= 2


def test_inference_cache_follows_parameter_updates(emu):
    """No-grad forwards reuse the packed weights / folded BatchNorm vectors of the previous one (dense testing: ten calls per
    video) -- and must notice every way the parameters change in the reference's loop: this package's optimizer kernels (raw
    pointers, invisible to torch's version counters), torch in-place updates, load_state_dict, BatchNorm buffers."""
    dev = torch.device("cpu")
    net, ref = _pair(dev)
    net.debug_keep_saved = False
    x, w = _data(seed=9)

    def check():
        ref.forced = None
        ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
        with torch.no_grad():
            f = net.features(x)
        assert rel_err(f, ref(x)) < 1e-5
    check()
    key0 = net.__dict__["_infer_cache"][0]
    check()
    assert net.__dict__["_infer_cache"][0] == key0                     # second call: cache hit
    opt = SSNSGD([{"params": [p for p in net.parameters() if p.requires_grad], "lr_mult": 1, "decay_mult": 1, "name": "w"}],
                 lr=0.05, momentum=0.9, weight_decay=5e-4)
    net.zero_grad(set_to_none=True)
    (net.features(x) * w).sum().backward()
    opt.step()
    check()                                                            # after the package's own SGD kernel
    with torch.no_grad():
        net.conv1_3x3.weight.mul_(1.5)
    check()                                                            # after a torch in-place update
    net.branch_3x3_bn.running_var.mul_(2.0)
    check()                                                            # BatchNorm buffers are part of the key
    net.load_state_dict({k: v * 0.5 if k.endswith("branch_3x3.bias") else v for k, v in net.state_dict().items()})
    check()
    # [r6] a write through `.data` (what checkpoint-averaging / EMA scripts do): no version counter moves, the key still matches --
    # the device-side checksum of the parameter bits notices, the pass is repeated from the parameters, the cache is rebuilt
    check()
    key1 = net.__dict__["_infer_cache"][0]
    v0 = net.conv1_3x3.weight._version
    net.conv1_3x3.weight.data.mul_(0.7)
    net.branch_3x3_bn.running_mean.data.add_(0.05)
    assert net.conv1_3x3.weight._version == v0
    check()
    assert sum(getattr(st, "stale_cache_hits", 0) for st in net._planes_states.values()) == 1
    assert net.__dict__["_infer_cache"][0] == key1 and not net.scale_fault()
    check()                                                            # (a hit again: nothing to repeat)
    assert sum(getattr(st, "stale_cache_hits", 0) for st in net._planes_states.values()) == 1


@pytest.mark.gpu
def test_graph_replay_protocol_on_changing_data_gpu(hip_library):
    """The protocol of a training loop that REPLAYS a captured step (what bench.py runs): the range check and the optimizer's skip
    word are part of the hipGraph; the static input buffer is overwritten between replays with batches of very different
    magnitude.  A replay whose tensors left the range of their delayed scales must leave weights and momentum untouched and raise
    the device word; after ``recalibrate()`` + an eager redo the weights must equal those of a twin model that ran the same
    sequence of batches eagerly (sync guard) -- step for step, to rounding."""
    import copy
    dev = torch.device("cuda:0")
    net = init_tiny(TinyBackbone()).to(dev).train()
    twin = copy.deepcopy(net)
    assert twin.planes_flag(dev) is not net.planes_flag(dev)

    def opt_for(m):
        return SSNSGD([{"params": [p for p in m.parameters() if p.requires_grad], "lr_mult": 1, "decay_mult": 1, "name": "w"}],
                      lr=1e-3, momentum=0.9, weight_decay=5e-4)
    opt, opt2 = opt_for(net), opt_for(twin)
    x0, w0 = _data(n=4, seed=40)
    xs, ws = x0.to(dev), w0.to(dev)                       # static buffers of the captured step
    flag = net.planes_flag(dev)

    def step(m, o, x, w, skip=None):
        o.zero_grad(set_to_none=True)
        (m.features(x) * w).sum().backward()
        o.step(skip_flag=skip)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                                 # eager warm-up: calibrates the scales
            step(net, opt, xs, ws, flag)
            step(twin, opt2, xs, ws)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(net, opt, xs, ws, flag)
    step(twin, opt2, xs, ws)                               # the twin's third step; the capture itself executed nothing ...

    def same(tol=2e-6):
        for (n1, p1), (n2, p2) in zip(net.named_parameters(), twin.named_parameters()):
            assert rel_err(p1, p2) < tol, (n1, rel_err(p1, p2))
    graph.replay()                                         # ... so the first replay is the net's third step

    torch.cuda.synchronize()
    same()
    for k in (1.0, 0.5, 1.0 / 64.0, 600.0, 1.0, 2000.0):   # the static batch changes magnitude between replays
        xs.copy_((x0 * k).to(dev))
        before = [p.detach().clone() for p in net.parameters()]
        graph.replay()
        torch.cuda.synchronize()
        if net.scale_fault():                              # the replay flagged itself: nothing may have moved ...
            assert all(torch.equal(p, q) for p, q in zip(net.parameters(), before)), k
            net.recalibrate()                              # ... host side: clear, recalibrate, redo the step eagerly
            step(net, opt, xs, ws, flag)
            assert not net.scale_fault()
        step(twin, opt2, xs, ws)
        torch.cuda.synchronize()
        same()
    assert twin.guard_stats()["fwd"] >= 1                 # the eager twin repaired the same jumps by itself


@pytest.mark.gpu
def test_graph_replay_protocol_on_the_real_ssn_gpu(hip_library):
    """The same protocol on the model and the step bench.py runs: the real BN-Inception SSN (2 videos = 144 frames of 224 x 224, three
    losses, the reference's parameter groups), forward + losses + backward + SGD captured into ONE hipGraph -- with the two-lane branch
    schedule, the grouped weight gradients and the stem on planes inside the capture -- and replayed on a static batch that is
    overwritten with frames of three magnitudes (x 1, x 12, x 1/40).  A replay that flagged itself must not have moved a weight; after
    ``recalibrate_scales()`` + the eager redo the weights must equal those of a twin that ran the same sequence eagerly (sync guard)."""
    import copy
    from action_detection_amd.ops.ssn_ops import ActivityLoss, ClassWiseRegressionLoss, CompletenessLoss
    from action_detection_amd.ssn_models import SSN
    from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = SSN(20, 2, 5, 2, "RGB", dropout=0, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(net.base_model)
    init_heads_synthetic(net, std=0.05)
    net.to(dev).train()
    twin = copy.deepcopy(net)
    net.base_model.scale_guard = "deferred"
    assert twin.scale_fault_flag(dev) is not net.scale_fault_flag(dev)

    def opt_for(m):
        return SSNSGD(m.get_optim_policies(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
    opt, opt2 = opt_for(net), opt_for(twin)
    batch0 = make_batch(2, "RGB", 20, seed=13)
    static = [t.to(dev) for t in batch0]
    flag = net.scale_fault_flag(dev)
    crit = (ActivityLoss(), CompletenessLoss(), ClassWiseRegressionLoss())

    def step(m, o, skip=None):
        o.zero_grad(set_to_none=True)
        out = m(*static)
        loss = crit[0](out[0], out[1]) + 0.1 * crit[1](out[2], out[3], 1, 7) + 0.1 * crit[2](out[4], out[5], out[6])
        loss.backward()
        o.step(skip_flag=skip)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            net.base_model.scale_guard = "sync"           # (eager warm-up calibrates; the capture then runs deferred)
            step(net, opt, flag)
            step(twin, opt2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    net.base_model.scale_guard = "deferred"
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(net, opt, flag)
    # (the capture executed nothing: both models have taken two steps)

    prev = {"net": [p.detach().clone() for p in net.parameters()], "twin": [p.detach().clone() for p in twin.parameters()]}

    def same():
        """The step just taken moved every parameter tensor of the net like the twin's: the UPDATES agree to 1e-2 of the largest
        update of the tensor (a skipped, doubled or clamped step is off by ~1; two fp32 executions of one step differ by the
        ReLU-flip level ~1e-3: the net's scales after a recalibration are not bit for bit the twin's).  (No bound on the weights
        themselves: a bias tensor of magnitude 0.03 takes updates of 1e-3 per step here, so 1e-2 of an update is already 3e-4 of it.)"""
        worst = 0.0
        for (n1, p1), p2, q1, q2 in zip(net.named_parameters(), twin.parameters(), prev["net"], prev["twin"]):
            d1, d2 = (p1.detach() - q1).double(), (p2.detach() - q2).double()
            if float(d2.abs().max()) > 0:
                e = float((d1 - d2).abs().max() / d2.abs().max())
                worst = max(worst, e)
                assert e < 1e-2, ("update of", n1, e)
        prev["net"] = [p.detach().clone() for p in net.parameters()]
        prev["twin"] = [p.detach().clone() for p in twin.parameters()]
        return worst
    flagged = [0]

    def replay_and_repair(k):
        """One replayed step of the net (+ the host side of the protocol when it flagged itself) and the twin's eager step."""
        before = [p.detach().clone() for p in net.parameters()]
        graph.replay()
        torch.cuda.synchronize()
        if net.scale_fault():
            flagged[0] += 1
            assert all(torch.equal(p, q) for p, q in zip(net.parameters(), before)), k
            net.recalibrate_scales()
            net.base_model.scale_guard = "sync"
            step(net, opt, flag)
            net.base_model.scale_guard = "deferred"
            assert not net.scale_fault()
        step(twin, opt2)
        torch.cuda.synchronize()
        same()

    replay_and_repair(1.0)      # (gradient maxima still move several-fold in the first steps on fresh head weights: may flag)
    for k in (1.0, 12.0, 1.0 / 40.0, 1.0):
        static[0].copy_((batch0[0] * k).to(dev))
        replay_and_repair(k)
    assert flagged[0] >= 1, "the magnitude jumps were meant to trip the range guard inside the replayed graph"


def test_a_pass_that_overflows_fp32_behaves_like_the_reference(emu):
    """weights that make the ACTIVATIONS overflow fp32 (a diverged run).  The reference carries inf / NaN forward and
    keeps running; the planes path used to raise "scales did not settle".  Now: no exception, non-finite features (what fp32 storage
    would hold), non-finite gradients, the fault word raised so that SSNSGD.step(skip_flag=) leaves the weights untouched -- and once the
    weights are sane again the very next call is correct and clears the word."""
    dev = torch.device("cpu")
    net, ref = _pair(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 16, 16, generator=g) * 40
    f0 = net.features(x)                                   # a sane, calibrated state first
    w = torch.randn(f0.shape, generator=g)
    (f0 * w).sum().backward()
    good = {k: v.clone() for k, v in net.state_dict().items()}
    opt = SSNSGD([{"params": [p for p in net.parameters() if p.requires_grad], "lr_mult": 1, "decay_mult": 1, "name": "all"}], lr=0.01,
                 momentum=0.9, weight_decay=0.0)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("weight") and p_.dim() == 4:
                p_.mul_(1e20)                              # two layers of this: ~1e22, then ~1e43 > 3.4e38 -- fp32 overflows to inf
    net.zero_grad(set_to_none=True)
    f = net.features(x)                                    # must not raise
    assert not torch.isfinite(f).any()
    assert net.scale_fault()
    (f * w).sum().backward()                               # must not raise either
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert grads and all(not torch.isfinite(g_).any() for g_ in grads)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt.step(skip_flag=net.planes_flag(dev)[0:1])          # the flagged step's update is skipped
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k
    # sane weights again (a reloaded checkpoint): the next call recalibrates by itself, is correct, and clears the word
    net.load_state_dict(good)
    ref.load_state_dict({k: v.double() for k, v in good.items()})
    _check_call(net, ref, x, w, dev)
    assert not net.scale_fault()

