"""Backbone executor on planes tensors (csrc/planes.h): the launch plan of ``bninception.BNInception`` (same manifest, same
plan transforms, same parameter surface and flat gradient layout) run on the kernels that keep every activation and every
activation gradient as two f16 planes in the channel-blocked NC8HW8 layout -- the operand format of the matrix cores.

What differs from the fp32-layout executor (bninception._run_forward / _run_backward):

* producers split, consumers multiply: the convolution / pool epilogues emit the planes, so no kernel of the step converts
  operands in its inner loop (the round-2 kernels spent 7-13 VALU per MFMA there);
* scales are delayed (``planes.SlotPool``): a tensor's power-of-two scale for this step comes from the largest magnitude its
  producers recorded in the previous step; the first step of an executor state is CALIBRATED by running the pass until no scale
  moves any more (2-3 passes);
* the RANGE GUARD makes that safe on data that changes from call to call (another batch, an evaluation pass between training
  steps, a reloaded checkpoint, a loss spike): producers record the magnitude BEFORE their clamp to the f16 range, one
  ``ssn_pl_range_check`` launch at the end of every pass raises a device word when a tensor left the range of the scale it was
  stored with, and a flagged pass is REPEATED with fresh scales before anything consumes it -- ``scale_guard = "sync"`` (default for
  eager calls): the pass polls the word itself (one 8-byte read per pass) and never returns a clamped result; inside a hipGraph
  capture nothing can be polled, so the check is captured, ``SSNSGD.step(skip_flag=...)`` refuses to apply a flagged step's
  gradients, and the owner of the graph polls the word behind the replay (``BNInception.scale_fault()`` /
  ``recalibrate()``; bench.py does).  With an overlapping gradient reducer attached the decision to repeat a backward is taken by
  ALL ranks (``hook.agree``), so the collectives stay matched;
* one stream: the launches of a step already fill the GPU (round 2 measured 1.5 % from four-stream overlap), so the planes path
  keeps the whole step on the caller's stream -- one amax / scale slot per tensor suffices and results stay deterministic.
  (Round 3 carried a switch that moved the weight gradients to a side stream: +1.0 ... +1.4 % on two boxes then, -0.3 % on the
  round-4 box -- 17.03 vs 16.98 ms, profiles/r4_bench_wgrad_side_stream_ab.txt -- i.e. nothing; removed.)
* the ReLU / frozen-BN backward of a layer is fused into whichever launch writes its output gradient last and reads only the
  SIGN of the activation's high plane (2 bytes per element instead of the 4 of an fp32 ``y``).

Supported: frozen BatchNorm (the reference's default ``bn_mode='frozen'``; /root/reference/ssn_models.py:95-105) on the square /
rectangular-tap plans.  Training-mode BatchNorm keeps the fp32-layout executor.
"""
import contextlib
import itertools
import math

import torch

from . import kernels as K
from . import planes as P
from .planes import PSlice, PlaneTensor


class PlanesState:
    """Scale / amax slots of one backbone (persistent across steps) and the calibration status."""

    MAX_TENSORS = 512
    GRAD_EXTRA_BITS = 3      # gradients: maximum lands in [2^9, 2^10) (5 - 6 bits to the f16 ceiling), kept while inside [2^7, 2^11)

    def __init__(self, device, flag=None):
        self.pool = P.SlotPool(2 * self.MAX_TENSORS, device, flag)
        # gradient tensors (the odd slots) keep GRAD_EXTRA_BITS more bits of head-room than activations: their largest element
        # moves > 5x between consecutive steps on the SAME batch (dropout mask, OHEM selection), an activation's barely at all
        self.pool.odd_extra_bits = self.GRAD_EXTRA_BITS
        self.act_slot = {}      # activation tensor name -> slot
        self.grad_slot = {}     # gradient tensor name -> slot
        self.fwd_calibrated = False
        self.bwd_calibrated = False
        self.calibration_passes = [0, 0]
        self.recalibrations = [0, 0]     # passes the range guard had repeated (forward, backward)
        self.fault_log = []              # diagnostics: the last few faults as (pass, [(tensor, amax * scale), ...])
        self.nonfinite = False           # the last settle() gave up on a pass whose maxima were inf / NaN

    def slot(self, name, grad):
        table = self.grad_slot if grad else self.act_slot
        if name not in table:
            # activations take the even slots, gradients the odd ones: both sets stay contiguous ranges for the update kernel
            table[name] = 2 * len(table) + (1 if grad else 0)
            assert table[name] < self.pool.amax.numel()
            self.pool.used = max(self.pool.used, table[name] + 1)
        return table[name]

    def update(self):
        """One launch over all slots (slots nobody wrote keep their scale)."""
        self.pool.update(first=0, count=self.pool.used)

    def check(self):
        """Range guard: one launch over the slots, raises the fault word (no host sync; capturable)."""
        self.pool.range_check()

    def fault(self):
        """The fault word (host sync): bit 0 = a tensor was clamped, bit 1 = a tensor fell far below its scale."""
        return int(self.pool.flag[0].item())

    def overflowed(self):
        return bool(self.fault() & 1)

    def describe_fault(self, which):
        """Diagnostics (host sync): the tensors whose recorded maximum left the range of their scale in the pass just run."""
        a = (self.pool.amax[:self.pool.used] * self.pool.scale[:self.pool.used]).tolist()
        names = {v: ("grad:" if g else "act:") + k for g, tab in ((0, self.act_slot), (1, self.grad_slot)) for k, v in tab.items()}
        floor = [16.0, 16.0 / (1 << self.pool.odd_extra_bits)]
        bad = [(names.get(i, "slot%d" % i), v) for i, v in enumerate(a) if v != v or v >= 65504.0 or 0.0 < v < floor[i & 1]]
        self.fault_log = (self.fault_log + [(which, bad)])[-8:]
        return bad

    def settle(self, relaunch, what):
        """Repeat a pass (``relaunch``) until no scale moves and nothing leaves its range; returns the number of repeats.  The
        pass must already have run once.  (Host syncs: eager only.)"""
        trail = []
        self.nonfinite = False
        for it in range(12):
            # a NON-FINITE recorded maximum is not a scale problem: the pass itself produced inf / NaN (diverged weights, an inf in
            # the input) -- fp32 storage would carry it forward, no scale can hold it.  Stop repeating, leave the fault word raised
            # (the optimizer skips the step), and let the caller poison its result like the reference's would be.
            if not bool(torch.isfinite(self.pool.amax[:self.pool.used]).all()):
                self.nonfinite = True
                self.pool.amax[:self.pool.used].zero_()
                self.pool.flag[0] = 1
                self.fault_log = (self.fault_log + [(what, [("non-finite maximum: pass not repeated", float("inf"))])])[-8:]
                return it
            if it >= 9:      # diagnostics for the error below: which tensors are still out of range / moving
                a = (self.pool.amax[:self.pool.used] * self.pool.scale[:self.pool.used]).tolist()
                names = {v: ("grad:" if g else "act:") + k for g, tab in ((0, self.act_slot), (1, self.grad_slot))
                         for k, v in tab.items()}
                trail.append([(names.get(i, "slot%d" % i), v) for i, v in enumerate(a)
                              if v != v or v >= 65504.0 or (0.0 < v < 1024.0) or v >= 16384.0][:8])
            self.pool.flag.zero_()
            self.update()
            fault, moved = self.pool.flag.tolist()
            if not moved and not (fault & 1):
                self.pool.flag.zero_()
                return it
            relaunch()
        raise RuntimeError("planes executor: %s scales did not settle; tensors outside [2^10, 2^14) x scale in the last passes: %r"
                           % (what, trail))


def _state(net, x):
    key = (x.device, x.shape[1], x.shape[2])
    st = net._planes_states.get(key)
    if st is None:
        st = net._planes_states[key] = PlanesState(x.device, net.planes_flag(x.device))
    return st


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def supported(net, plan):
    if net._train_bn_ids() and not net.planes_train_bn:
        return False
    for idx, op in enumerate(plan):
        if op["kind"] == "pool" and op["pool"] != "max":
            # a plain average pool (plans with training-mode BatchNorm keep it in front of its projection): stride 1, and the LAST
            # reader of its input, so that its backward is the first writer of that gradient (the stencil kernel does not accumulate)
            if op["pool"] != "avg" or op["s"] != 1 or any(q.get("src") == op["src"] for q in plan[idx + 1:]):
                return False
        if op["kind"] == "bn_train" and any(c % 8 for c in op["couts"]):
            return False
        if op["kind"] == "conv" and (op["cout"] % 8 or (op["src"] != "data" and op["cin"] % 8)):
            return False
    return True


def _conv_taps(op):
    return op.get("kh", op["k"]), op.get("kw", op["k"]), op.get("ph", op["p"]), op.get("pw", op["p"])


def _is_rect(op):
    kh, kw, _, _ = _conv_taps(op)
    return kh != kw or kh not in (1, 3, 7)


def _fold_bn(net, plan, shapes, dev):
    """tscale: per tensor folded-BN scale of every channel (NaN: not a conv+ReLU output); shift_of: per layer shift vectors."""
    tscale, shift_of = {}, {}
    # one NaN-filled buffer for the scale vectors of all tensors (one fill launch per forward, not one per tensor)
    offs, o = {}, 0
    for name_, v in shapes.items():
        if not isinstance(v, tuple):      # (the plan's bookkeeping entries)
            continue
        offs[name_] = o
        o += (v[0] + 7) // 8 * 8          # (every vector starts 32-byte aligned, as separate allocations did)
    # (kept across passes: the entries nobody folds into stay NaN, the others are rewritten by every pass with what the parameters
    # say -- one fill launch per state instead of one per forward)
    # ... per PLAN: which entries are folded into depends on it (fused / per-layer block inputs, which BatchNorm layers are in training
    # mode, raw projection rows) -- an entry another plan wrote must read NaN again under this one
    sig = (dev, o, tuple((tuple(op["lids"]), bool(op.get("bn_train")), op.get("raw_from", -1), op.get("row_split", -1), op.get("row_gap", 0),
                          op["dst"], op["dst_c0"], op.get("final"), op.get("proj_final")) for op in plan if op["kind"] == "conv"))
    keep = net.__dict__.setdefault("_nan_scale_flat", {})
    flat = keep.get(sig)
    if flat is None:
        if len(keep) >= 8:
            keep.clear()
        flat = keep[sig] = torch.full((o,), float("nan"), device=dev, dtype=torch.float32)

    def scale_slice(name, c0, c):
        if name not in tscale:
            tscale[name] = flat[offs[name]:offs[name] + shapes[name][0]]
        return tscale[name][c0:c0 + c]

    shift_flat = torch.empty(sum(op["cout"] for op in plan if op["kind"] == "conv"), device=dev, dtype=torch.float32)
    fold = ([], [], [], [], [], [], [], [])
    soff = 0
    for op in plan:
        if op["kind"] != "conv":
            continue
        shift_of[op["lids"][0]] = shift_flat[soff:soff + op["cout"]]
        for lid_, o_ in zip(op["lids"], [sum(op["couts"][:q]) for q in range(len(op["lids"]))]):
            shift_of.setdefault(lid_, shift_flat[soff + o_:soff + o_ + getattr(net, lid_).out_channels])
        off = 0
        aff_dst, aff_c0 = op.get("final", (op["dst"], op["dst_c0"]))
        if op.get("bn_train"):
            # batch-statistics layer: nothing to fold; the NaN scales of its slice say "not a frozen ReLU / BN output", so the fused
            # backward epilogues of the consumers leave that gradient alone (the bn_train op owns its backward)
            scale_slice(aff_dst, aff_c0, op["cout"])
            soff += op["cout"]
            continue
        for lid, c in zip(op["lids"], op["couts"]):
            conv, bn = getattr(net, lid), getattr(net, lid + "_bn")
            if "raw_from" in op and off >= op["raw_from"]:
                sdst = scale_slice(op["proj_final"][0], op["proj_final"][1], c)
            else:
                sdst = scale_slice(aff_dst, aff_c0 + off + (op["row_gap"] if off >= op.get("row_split", 1 << 30) else 0), c)
            for lst, v in zip(fold, (conv.bias.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                     bn.eps, sdst, shift_flat[soff + off:soff + off + c])):
                lst.append(v)
            off += c
        soff += op["cout"]
    if fold[0]:
        K.bn_fold_multi(*fold)
    return tscale, shift_of, scale_slice


def _pool_out(h, k, s, p):
    """ceil-mode output size of the manifest's pools (torch rule: the last window must start inside the padded input)."""
    o = -(-(h + 2 * p - k) // s) + 1
    if (o - 1) * s >= h + p:
        o -= 1
    return o


def _branch_lanes(plan):
    """Two-lane schedule of the Inception blocks (net.branch_lanes): behind a block-input launch the plan holds the SHORT chain
    (the 3x3 branch, and the pool behind / beside its projection) and the LONG one (double_3x3_1 -> double_3x3_2), which depend on
    nothing but the block input.  Returns ({plan index: "fork" | "side"} for the ops of the short chain -- they go to a side stream --,
    {plan indices in front of which the side stream is joined}); forward order.  The backward walks the same sets in reverse."""
    side, joins = {}, set()
    i = 0
    while i < len(plan):
        op = plan[i]
        seg = plan[i + 1:i + 5]
        ok = (op["kind"] == "conv" and len(op["lids"]) >= 2 and len(seg) == 4 and all(q["kind"] == "conv" and len(q["lids"]) == 1 for q in seg[:3])
              and seg[3]["kind"] in ("pool", "pool_aff") and seg[2]["src"] == seg[1]["dst"] and seg[0]["src"] == op["dst"]
              and seg[1]["src"] == op["dst"] and seg[1]["dst"] not in (seg[0]["dst"], op["dst"])
              # the pool runs on the side lane with no wait on the main lane behind the fork: it may only read what existed at the
              # fork (the block input, the block-input launch's rows) or what its own lane wrote
              and seg[3]["src"] in (op["src"], op["dst"], seg[0]["dst"]))
        if ok:
            side[i + 1] = "fork"
            side[i + 4] = "side"
            joins.add(i + 5)
            i += 5
        else:
            i += 1
    return side, joins


def _stem_on_planes(net, x):
    """The weight gradient of the space-to-depth stem as a problem of the grouped launch (csrc/wgrad_pl.hip: wgrad_stem_body) -- on
    planes operands like every other layer: no fp32 copy of the frames, no fp32 output gradient.  Needs the grouped launches and
    rows of at most 114 (space-to-depth) pixels: the body's halo buffer."""
    return net.group_wgrad and net.stem_planes and x.shape[3] // 2 <= 114


def _dgrad_is_s2(op):
    return len(op["lids"]) == 1 and not op["rect"] and op["k"] == 3 and op["s"] == 2


def _pack_dgrad(net, plan):
    """The data-gradient operands of every convolution that has one (not the layers on the caller's frames)."""
    dg_ops = [op for op in plan if op["kind"] == "conv" and op["src"] != "data"]
    packed_dg = {}
    for op in dg_ops:
        if _dgrad_is_s2(op):
            packed_dg[op["lids"][0]] = K.pack_dgrad_s2(getattr(net, op["lids"][0]).weight.detach())
        elif op["s"] != 1:
            raise NotImplementedError("planes layout: data gradient of a %dx%d / stride-%d convolution" % (op["k"], op["k"], op["s"]))
    rect_ops = [op for op in dg_ops if op["rect"]]
    packed_dg.update(zip((op["lids"][0] for op in rect_ops),
                         K.pack_rect_multi([getattr(net, op["lids"][0]).weight.detach() for op in rect_ops], dgrad=True)))
    sq_ops = [op for op in dg_ops if not op["rect"] and not _dgrad_is_s2(op)]
    packed_dg.update(zip((op["lids"][0] for op in sq_ops), K.pack_weights_multi(
        [([getattr(net, lid).weight.detach() for lid in op["lids"]], 1) for op in sq_ops], x6=True)))
    return packed_dg


def run_forward(net, x, keep):
    plan, shapes = net._plan(x)
    n, dev = x.shape[0], x.device
    st = _state(net, x)
    # inference (nothing kept for a backward): the folded BatchNorm vectors and the packed weights only depend on the parameters --
    # reused while no parameter / buffer has been written (dense testing calls the backbone ten times per video: 3 pack launches,
    # the fold and ~2 ms of Python per call)
    # Validity: torch's version counters (optimizers, load_state_dict, init functions, anything under no_grad) + the counter of this
    # package's own optimizer kernels + the storage addresses; a write through `.data` is invisible to all three -- [r6] so every hit
    # also compares a device-side checksum of the parameter bits (kernels.ParamChecksum; with scale_guard "off" nothing polls the
    # word: call net.recalibrate() after such a write).  SSN_INFER_CACHE=0 switches the cache off.
    ckey = None
    hit = None
    if not keep and net.infer_cache:
        ckey = (dev, tuple(x.shape[1:]), K.PARAM_EPOCH, bool(net.training), tuple(net._train_bn_ids()),      # (the last two: the plan)
                tuple((t._version, t.data_ptr()) for t in itertools.chain(net.parameters(), net.buffers())))
        hit = net.__dict__.get("_infer_cache")
        if hit is not None and hit[0] == ckey:
            tscale, shift_of, scale_slice, packed = hit[1]
            # [r6] ... and a write through `.data` / a raw pointer (checkpoint averaging, EMA scripts) that none of the above sees: the
            # parameters' BITS are compared with the ones the cache was built from, on the device (no host read) -- a mismatch raises
            # bit 2 of the fault word and this pass is repeated without the cache by the guard's poll below
            if net.scale_guard != "off":
                hit[2].check(st.pool.flag, 4)
        else:
            hit = None
    if keep or hit is None:
        tscale, shift_of, scale_slice = _fold_bn(net, plan, shapes, dev)

    conv_ops = [op for op in plan if op["kind"] == "conv"]
    for op in conv_ops:
        op["rect"] = _is_rect(op)
        op["s2d"] = (op["src"] == "data" and len(op["lids"]) == 1 and (op["k"], op["s"], op["p"]) == (7, 2, 3)
                     and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
        if op["src"] == "data" and not op["s2d"] and not op["rect"] and op["k"] not in (1, 3):
            raise NotImplementedError("planes layout: first convolution %dx%d / stride %d" % (op["k"], op["k"], op["s"]))
    packed_dg = None
    if keep or hit is None:
        # every weight operand of the step -- forward, and with a backward to come the data-gradient ones too (the weights autograd
        # would have saved) -- recorded and issued in three launches (kernels.PackBatch; one state per kind of pass so that the
        # operand buffers, and with them the device-resident plan, repeat from step to step)
        pstate = net.__dict__.setdefault("_pack_states", {}).setdefault((dev, bool(keep), tuple(x.shape[1:])), {})
        batch = K.PackBatch(pstate, dev) if net.batch_packing else contextlib.nullcontext()
        with batch:
            packed = {}
            for op in conv_ops:
                if op["s2d"]:
                    packed[op["lids"][0]] = K.pack_weights_rect(K.s2d_weights(getattr(net, op["lids"][0]).weight.detach()))
            rect_ops = [op for op in conv_ops if op["rect"] and not op["s2d"]]
            packed.update(zip((op["lids"][0] for op in rect_ops),
                              K.pack_rect_multi([getattr(net, op["lids"][0]).weight.detach() for op in rect_ops])))
            sq_ops = [op for op in conv_ops if not op["rect"] and not op["s2d"]]
            packed.update(zip((op["lids"][0] for op in sq_ops), K.pack_weights_multi(
                [([getattr(net, lid).weight.detach() for lid in op["lids"]], 0) for op in sq_ops], x6=True)))
            if keep:
                packed_dg = _pack_dgrad(net, plan)
        if ckey is not None:
            fingerprint = K.ParamChecksum([t.detach() for t in itertools.chain(net.parameters(), net.buffers())
                                           if t.element_size() == 4 and t.numel() > 0], dev)
            net.__dict__["_infer_cache"] = (ckey, (tscale, shift_of, scale_slice, packed), fingerprint)

    bnstat, bn_before = {}, {}      # training-mode BatchNorm layers: layer id -> (batch mean of z, 1 / sqrt(var + eps))
    bn_factor = {}                  # ... with momentum=None: the cumulative-average factor of this call

    def launch_all(acts, argmax):
        # the scales this pass stores with, frozen: the pool's entries move on with the next pass (another sub-batch, an
        # evaluation forward), while this pass's tensors may still be read by its backward
        snap = st.pool.scale.clone()

        def get(name):
            if name not in acts:
                c, h, w = shapes[name]
                acts[name] = PlaneTensor(n, c, h, w, dev, st.pool, st.slot(name, False), snap)
            return acts[name]

        feat_box = [None]

        def run_op(op):
            if op["kind"] == "conv":
                cout, cin = op["cout"], op["cin"]
                raw = bool(op.get("raw"))
                shift = None if raw else shift_of[op["lids"][0]]
                scale = None if raw else scale_slice(op["dst"], op["dst_c0"], cout + op.get("row_gap", 0))
                kh, kw, ph, pw = _conv_taps(op)
                dst = PSlice(get(op["dst"]), op["dst_c0"], cout)
                wp = packed[op["lids"][0]]
                flops = 2.0 * n * shapes[op["dst"]][1] * shapes[op["dst"]][2] * cout * cin * kh * kw
                if op["s2d"]:
                    if "data_s2d" not in acts:
                        acts["data_s2d"] = PlaneTensor(n, 4 * cin, x.shape[2] // 2, x.shape[3] // 2, dev, st.pool,
                                                       st.slot("data_s2d", False), snap)
                        measured = None
                        if keep and not _stem_on_planes(net, x):
                            # the stem's weight gradient on the fp32-layout kernel (SSN_STEM_PLANES=0 / rows wider than the stem
                            # body's halo buffer; see run_backward): it reads an fp32 space-to-depth copy of the frames
                            acts["data_s2d_f32"] = K.space_to_depth2(x)
                            measured = acts["data_s2d_f32"]._ssn_amax      # (that pass over the frames also took their maximum)
                        # stem on planes: the frames take a DELAYED scale like every other tensor of the pass (their maximum barely
                        # moves between batches; the range guard covers the rest) -- no measuring pass over the 173 MB of frames
                        delayed = keep and measured is None and net.delayed_input_scale
                        P.from_f32(x, acts["data_s2d"], s2d=True, exact=not delayed, amax=measured)
                    src = P.pfull(acts["data_s2d"])
                    net._timed("conv_fwd_pl", op["lids"][0], flops, lambda: P.conv_fwd(
                        PSlice(src.t, 0, src.t.g * 8), wp, scale, shift, dst, 4, 4, 1, 2, 2, not raw, net._pl_tile("fwd", op, n, shapes)))
                else:
                    if op["src"] == "data":
                        if "data" not in acts:
                            acts["data"] = PlaneTensor(n, cin, x.shape[2], x.shape[3], dev, st.pool, st.slot("data", False), snap)
                            P.from_f32(x, acts["data"], exact=True)
                        src = PSlice(acts["data"], 0, acts["data"].g * 8)
                    else:
                        src = PSlice(acts[op["src"]], op["src_c0"], cin)
                    net._timed("conv_fwd_pl", op["lids"][0], flops, lambda: P.conv_fwd(
                        src, wp, scale, shift, dst, kh, kw, op["s"], ph, pw, not raw, net._pl_tile("fwd", op, n, shapes),
                        raw_from=op.get("raw_from", 0), row_split=op.get("row_split", 0), row_gap=op.get("row_gap", 0)))
            elif op["kind"] == "pool" and op["pool"] == "avg":
                c = op["c"]
                P.avgpool_affine(PSlice(acts[op["src"]], 0, c), PSlice(get(op["dst"]), op["dst_c0"], c), None, None, False,
                                 op["k"], op["p"])
            elif op["kind"] == "bn_train":
                # training-mode BatchNorm2d of the layer(s) whose bare convolution wrote z = acts[op["src"]] (planes_bn.hip)
                off = 0
                bws = torch.empty(P.bn_train_workspace_bytes(op["c"]) // 4, device=dev, dtype=torch.float32)
                for lid, c in zip(op["lids"], op["couts"]):
                    conv, bn = getattr(net, lid), getattr(net, lid + "_bn")
                    mean = torch.empty(c, device=dev, dtype=torch.float32)
                    invstd = torch.empty(c, device=dev, dtype=torch.float32)
                    zs = PSlice(acts[op["src"]], off, c)
                    track = bn.track_running_stats and bn.running_mean is not None
                    rm, rv = (None, None)
                    if track:
                        if lid not in bn_before:
                            bn_before[lid] = (bn.running_mean.clone(), bn.running_var.clone())
                            if bn.num_batches_tracked is not None:
                                bn.num_batches_tracked.add_(1)      # torch increments BEFORE it updates the running statistics
                            if bn.momentum is None:
                                # torch: a CUMULATIVE moving average, factor 1 / num_batches_tracked (counted first) -- one host
                                # read per layer and call, as torch's own module does; not capturable
                                if capturing:
                                    raise RuntimeError("BatchNorm2d(momentum=None) in training mode reads its batch counter on the "
                                                       "host and cannot be captured into a hipGraph")
                                bn_factor[lid] = 1.0 / float(bn.num_batches_tracked.item())
                        else:      # a pass repeated by the calibration / the range guard restarts from the statistics it found
                            bn.running_mean.copy_(bn_before[lid][0])
                            bn.running_var.copy_(bn_before[lid][1])
                        rm, rv = bn.running_mean, bn.running_var
                    P.bn_train_stats(zs, conv.bias.detach(), mean, invstd, rm, rv, bn.eps,
                                     bn_factor.get(lid, 0.0) if bn.momentum is None else bn.momentum, bws)
                    P.bn_train_apply(zs, PSlice(get(op["dst"]), op["dst_c0"] + off, c), mean, invstd, bn.weight.detach(),
                                     bn.bias.detach(), True)
                    bnstat[lid] = (mean, invstd)
                    off += c
            elif op["kind"] == "pool":
                c = op["c"]
                out = PSlice(get(op["dst"]), op["dst_c0"], c)
                _, ho, wo = shapes[op["dst"]]
                am = None
                if keep:
                    am = argmax.get(op["lid"])
                    if am is None:
                        am = argmax[op["lid"]] = torch.empty((n, c // 8, ho * wo, 8), device=dev, dtype=torch.uint8)
                P.maxpool_fwd(PSlice(acts[op["src"]], 0, c), out, am, op["k"], op["s"], op["p"])
            elif op["kind"] == "pool_aff":
                c = op["c"]
                P.avgpool_affine(PSlice(acts[op["src"]], op.get("src_c0", 0), c), PSlice(get(op["dst"]), op["dst_c0"], c),
                                 scale_slice(op["dst"], op["dst_c0"], c), shift_of[op["conv"]], True, op["k"], op["p"])
            else:
                feat_box[0] = torch.empty((n, op["c"]), device=dev, dtype=torch.float32)
                P.gap_fwd(PSlice(acts[op["src"]], 0, op["c"]), feat_box[0])

        side_ops, joins = _branch_lanes(plan) if (net.branch_lanes and x.is_cuda) else ({}, set())
        side = net._side_stream(dev) if side_ops else None
        main = torch.cuda.current_stream(dev) if side_ops else None
        for pidx, op in enumerate(plan):
            if pidx in joins:
                main.wait_stream(side)
            if pidx in side_ops:
                get(op["dst"])                          # (every allocation of the pass happens on the caller's stream)
                if side_ops[pidx] == "fork":
                    side.wait_stream(main)              # the block-input launch has been issued
                with torch.cuda.stream(side):
                    run_op(op)
            else:
                run_op(op)
        if side_ops:
            main.wait_stream(side)
        return feat_box[0]

    # delayed scales: this pass stores with the scales derived from the previous pass's maxima
    res = {"acts": {}, "feat": None}
    argmax = {}
    capturing = _capturing(x)

    def again():
        res["acts"] = {}
        res["feat"] = launch_all(res["acts"], argmax)

    st.nonfinite = False
    if not st.fwd_calibrated:
        # first pass of this state: no history.  Activations start at 1/4 (head-room up to 2.6e5), then the pass is repeated
        # until no scale moves (each pass fixes every tensor whose inputs were already right)
        if capturing:
            raise RuntimeError("planes executor: the first pass of a state calibrates its scales with host syncs and cannot be "
                               "captured into a hipGraph; run one eager step first")
        st.pool.scale[0::2] = 0.25
        again()
        st.calibration_passes[0] = 1 + st.settle(again, "forward")
        st.fwd_calibrated = True
    else:
        st.update()
        again()
        if net.scale_guard != "off":
            # range guard: did every tensor fit the scale last pass's maxima gave it?  (another batch, an evaluation pass after
            # training steps, reloaded weights ...)  Eager: poll and repeat the pass before anyone reads it; capture: the owner of
            # the graph polls the word (the optimizer kernel skips a flagged step)
            st.check()
            word = st.fault() if (net.scale_guard == "sync" and not capturing) else 0
            if word & 4:
                # the inference cache was built from other parameter bits (a `.data` write): drop it and run the pass from the parameters
                net.__dict__.pop("_infer_cache", None)
                st.pool.flag[0] = word & ~4
                st.stale_cache_hits = getattr(st, "stale_cache_hits", 0) + 1
                return run_forward(net, x, keep)
            if word:
                st.describe_fault("forward")
                st.recalibrations[0] += 1 + st.settle(again, "forward")
    acts, feat = res["acts"], res["feat"]
    if st.nonfinite:
        # the pass overflowed fp32 itself (settle() gave up): the planes hold clamped values, the reference would hold inf / NaN --
        # hand out what IT would: non-finite features (the loss and every gradient follow; the raised fault word makes
        # SSNSGD.step(skip_flag=) leave the weights alone, the next finite pass recalibrates and clears it)
        feat.fill_(float("nan"))
        st.fwd_calibrated = True
        st.nonfinite_passes = getattr(st, "nonfinite_passes", 0) + 1
    saved = (plan, shapes, acts, argmax, tscale, {"packed": packed, "packed_dg": packed_dg, "bnstat": bnstat, "nonfinite": st.nonfinite}, st) if keep else None
    return feat, saved


def run_backward(net, dfeat, saved, hook=True):
    plan, shapes, acts, argmax, tscale, extras, st = saved
    bnstat, bn_grads = extras["bnstat"], {}
    n, dev = dfeat.shape[0], dfeat.device
    layout, total = net.flat_grad_layout(plan)
    lay = {lid: (wo, wn, bo, bn) for lid, wo, wn, bo, bn in layout}
    flat = torch.empty(total, device=dev, dtype=torch.float32)

    # workspace: the split-K slabs of EVERY weight gradient of the pass (each layer its own region: the reductions are deferred and
    # issued together, one launch instead of one per layer; 2.2 GB at the bench batch), channel-sum scratch
    ws_off, ws_bytes = {}, 0
    # two lanes: the 3x3 branch's weight gradient (+ its reduction) runs on the side stream beside double_3x3_1/2's on the main one, so
    # per-layer launches must not share ONE slab region even when their reductions are not deferred
    lanes_on = bool(net.branch_lanes and dfeat.is_cuda and _branch_lanes(plan)[0])
    for op in plan:
        if op["kind"] != "conv":
            continue
        kh, kw, _, _ = _conv_taps(op)
        _, ho, wo = shapes[op["dst"]]
        if op.get("s2d"):
            need = max(P.wgrad_workspace_bytes(n, 4 * op["cin"], op["cout"], ho, wo, 4, 4, net._pl_tile("wgrad", op, n, shapes)),
                       K.wgrad_x6_workspace_bytes(n, 4 * op["cin"], op["cout"], ho, wo, 4))
        else:
            need = P.wgrad_workspace_bytes(n, op["cin"], op["cout"], ho, wo, kh, kw, net._pl_tile("wgrad", op, n, shapes))
        need = (need + 1023) // 1024 * 1024
        if net.group_wgrad and (not op.get("s2d") or "data_s2d_f32" not in acts):
            continue                                   # (grouped: the slabs live in the group's own workspace)
        if not net.defer_wgrad_reduce and not lanes_on:
            ws_off[op["lids"][0]] = (0, need)
            ws_bytes = max(ws_bytes, need)
        else:
            ws_off[op["lids"][0]] = (ws_bytes, need)
            ws_bytes += need
    ws_all = net._workspace(ws_bytes, dev)

    def ws_of(op):
        if op["lids"][0] not in ws_off:
            return None
        o, nb = ws_off[op["lids"][0]]
        return ws_all[o // 4:(o + nb) // 4]
    # (the bias sums of the projections in front of their pools are deferred with the reductions and issued together: the scratch
    # holds all of them)
    cs_ws = torch.empty(P.channel_sum_workspace_bytes(sum(op["cout"] for op in plan if op["kind"] == "conv" and
                                                           (op.get("raw") or "raw_from" in op)) + 8) // 4,
                        device=dev, dtype=torch.float32)

    dg_s2 = {op["lids"][0]: _dgrad_is_s2(op) for op in plan if op["kind"] == "conv" and op["src"] != "data"}
    packed_dg = extras["packed_dg"]      # (packed with the forward operands: the weights the forward saw)

    def src_key(op):
        return (op["src"], op.get("src_c0", 0))
    last_writer = {}
    for idx in range(len(plan) - 1, -1, -1):
        last_writer[src_key(plan[idx])] = idx
    first_conv = net._conv_ids[0]
    # the stem's output tensor, when a max pool (and nothing else) reads it: its gradient is produced in the fp32 layout
    stem_out = None
    for op in plan:
        if op["kind"] == "conv" and op.get("s2d") and "data_s2d_f32" in acts:
            readers = [q for q in plan if q.get("src") == op["dst"]]
            if len(readers) == 1 and readers[0]["kind"] == "pool" and readers[0]["pool"] == "max" and op["dst_c0"] == 0:
                stem_out = op["dst"]

    conv_of = {op["dst"]: op for op in plan if op["kind"] == "conv" and op.get("bn_train")}      # z tensor -> its convolution

    def launch_all(grads, fire_hook):
        masked, inited = {}, set()
        pending_end = total
        pending_reduce, pending_sums = [], []     # deferred split-K reductions / the channel sums that must follow them

        pending_wgrad = []                        # grouped mode: (planes.WgradJob, flops) of the weight gradients not yet launched
        wgrad_flops_total = sum(2.0 * n * shapes[q["dst"]][1] * shapes[q["dst"]][2] * q["cout"] * q["cin"] * _conv_taps(q)[0] * _conv_taps(q)[1]
                                for q in plan if q["kind"] == "conv")
        pending_post = []                         # ... and what has to follow their reduction (the stem: space-to-depth taps -> 7x7)

        def flush():
            if pending_wgrad:
                # every weight gradient recorded since the last flush: <= 5 launches over a device-resident problem table + ONE
                # reduction (ssn_conv_wgrad_pl_group) -- their output gradients are all final by now, and they depend on nothing else
                jobs = [j for j, _ in pending_wgrad]
                net._timed("conv_wgrad_pl", "group", sum(f for _, f in pending_wgrad),
                           lambda: P.conv_wgrad_group(jobs, *net._wgrad_group_buffers(jobs, dev)))
            if pending_reduce:       # (timed with the weight-gradient family it belongs to: bench.py's roofline_detail)
                entries = list(pending_reduce)
                net._timed("conv_wgrad_pl", "reduce_multi", 0.0, lambda: P.wgrad_reduce_multi(entries))
            if pending_sums:
                P.channel_sum_multi(list(pending_sums), cs_ws)
            for fn in pending_post:
                fn()
            del pending_reduce[:], pending_sums[:], pending_wgrad[:], pending_post[:]
        group = net.group_wgrad
        defer = pending_reduce if net.defer_wgrad_reduce else None

        def gbuf(name):
            if name not in grads:
                c, h, w = shapes[name]
                grads[name] = PlaneTensor(n, c, h, w, dev, st.pool, st.slot(name, True))
            return grads[name]

        def mask_args(idx, op, width):
            key = src_key(op)
            if last_writer.get(key) == idx and key[0] in tscale and key[0] != "data":
                masked.setdefault(key[0], []).append((key[1], key[1] + width))
                return PSlice(acts[key[0]], key[1], width), tscale[key[0]][key[1]:key[1] + width]
            return None, None

        def is_masked(name, c0, c):
            covered = sum(max(0, min(hi, c0 + c) - max(lo, c0)) for lo, hi in masked.get(name, []))
            assert covered in (0, c), "partially finalised gradient slice %s[%d:%d]" % (name, c0, c0 + c)
            return covered == c

        def run_op(idx):
            nonlocal pending_end
            op = plan[idx]
            if op["kind"] == "gap":
                c = op["c"]
                key = (op["src"], 0)
                assert key not in inited
                fuse = op["src"] in tscale
                P.gap_bwd(dfeat, PSlice(gbuf(op["src"]), 0, c), mask=PSlice(acts[op["src"]], 0, c) if fuse else None,
                          mask_scale=tscale[op["src"]][0:c] if fuse else None)
                if fuse:
                    masked.setdefault(op["src"], []).append((0, c))
                inited.add(key)
            elif op["kind"] == "pool" and op["pool"] == "avg":
                # plain average pool (stride 1, zero padding counted): its adjoint is the same stencil on the gradient
                c = op["c"]
                key = (op["src"], 0)
                assert key not in inited, "average pool %s is not the first writer of its input's gradient" % op["lid"]
                P.avgpool_affine(PSlice(grads[op["dst"]], op["dst_c0"], c), PSlice(gbuf(op["src"]), 0, c), None, None, False,
                                 op["k"], op["p"])
                inited.add(key)
            elif op["kind"] == "bn_train":
                # batch-norm backward of the layer(s): the slice's gradient arrives untouched (NaN scales, see _fold_bn)
                off = 0
                bws = torch.empty(P.bn_train_workspace_bytes(op["c"]) // 4, device=dev, dtype=torch.float32)
                cop = conv_of[op["src"]]
                stem32 = bool(cop.get("s2d")) and "data_s2d_f32" in acts      # (its weight gradient runs on the fp32-layout kernel)
                for lid, c in zip(op["lids"], op["couts"]):
                    bn = getattr(net, lid + "_bn")
                    mean, invstd = bnstat[lid]
                    dgamma = torch.empty(c, device=dev, dtype=torch.float32)
                    dbeta = torch.empty(c, device=dev, dtype=torch.float32)
                    if stem32:
                        _, h_, w_ = shapes[op["src"]]
                        dz = grads["__stem_f32__"] = K.attach_amax(K.guarded_empty((n, c, h_, w_), dev))
                    else:
                        dz = PSlice(gbuf(op["src"]), off, c)
                    P.bn_train_bwd(PSlice(grads[op["dst"]], op["dst_c0"] + off, c), PSlice(acts[op["dst"]], op["dst_c0"] + off, c),
                                   PSlice(acts[op["src"]], off, c), mean, invstd, bn.weight.detach(), dgamma, dbeta, dz, bws, True)
                    bn_grads[lid] = (dgamma, dbeta)
                    off += c
                inited.add((op["src"], 0))
            elif op["kind"] == "pool":
                c = op["c"]
                key = (op["src"], 0)
                my, ms = mask_args(idx, op, c)
                # a 3x3 / stride-2 pool that is the only writer of its input's gradient: the fused ReLU decision is read from the
                # POOLED activation (the window's maximum IS the element the gradient goes to) -- a quarter of the mask bytes
                pooled = (my is not None and key not in inited and (op["k"], op["s"]) == (3, 2) and op["p"] in (0, 1)
                          and net.pooled_mask)
                if pooled:
                    my = PSlice(acts[op["dst"]], op["dst_c0"], c)
                if op["src"] == stem_out:
                    # the stem's output gradient has ONE consumer, the weight gradient of a 3- (12-) channel layer, which the
                    # planes kernels would pad to 32 input channels: it goes to the fp32-layout split kernel, in its layout
                    assert key not in inited
                    _, h_, w_ = shapes[op["src"]]
                    grads["__stem_f32__"] = K.attach_amax(K.guarded_empty((n, c, h_, w_), dev))
                    P.maxpool_bwd(PSlice(grads[op["dst"]], op["dst_c0"], c), argmax[op["lid"]], grads["__stem_f32__"], op["k"],
                                  op["s"], op["p"], mask=my, mask_scale=ms, mask_pooled=pooled)
                else:
                    P.maxpool_bwd(PSlice(grads[op["dst"]], op["dst_c0"], c), argmax[op["lid"]], PSlice(gbuf(op["src"]), 0, c),
                                  op["k"], op["s"], op["p"], accumulate=key in inited, mask=my, mask_scale=ms, mask_pooled=pooled)
                inited.add(key)
            elif op["kind"] == "pool_aff":
                c = op["c"]
                g = PSlice(grads[op["dst"]], op["dst_c0"], c)
                if not is_masked(op["dst"], op["dst_c0"], c):
                    P.relu_bn_bwd(g, PSlice(acts[op["dst"]], op["dst_c0"], c), tscale[op["dst"]][op["dst_c0"]:op["dst_c0"] + c])
                    masked.setdefault(op["dst"], []).append((op["dst_c0"], op["dst_c0"] + c))
                P.avgpool_affine(g, PSlice(gbuf(op["src"]), op.get("src_c0", 0), c), None, None, False, op["k"], op["p"])
                inited.add((op["src"], op.get("src_c0", 0)))
            else:
                cout, cin, s = op["cout"], op["cin"], op["s"]
                lids = op["lids"]
                raw = bool(op.get("raw"))
                c_aff, gap, split = op.get("raw_from", cout), op.get("row_gap", 0), op.get("row_split", cout)
                ranges = [(0, min(split, c_aff))] + ([(split + gap, c_aff - split)] if c_aff > split else [])
                if "row_gap" in op:
                    offs = [sum(op["couts"][:q]) for q in range(len(lids))]
                    ranges = [(o_ + (gap if o_ >= split else 0), c_) for o_, c_ in zip(offs, op["couts"]) if o_ < c_aff]
                for r0, rc in ([] if raw else ranges):
                    if not is_masked(op["dst"], op["dst_c0"] + r0, rc):
                        P.relu_bn_bwd(PSlice(grads[op["dst"]], op["dst_c0"] + r0, rc), PSlice(acts[op["dst"]], op["dst_c0"] + r0, rc),
                                      tscale[op["dst"]][op["dst_c0"] + r0:op["dst_c0"] + r0 + rc])
                wo, wn, bo, bn = lay[lids[0]]
                for extra in lids[1:]:
                    wo2, wn2, bo2, bn2 = lay[extra]
                    assert wo2 == wo + wn and bo2 == bo + bn
                    wn, bn = wn + wn2, bn + bn2
                kh, kw, ph, pw = _conv_taps(op)
                dw = flat[wo:wo + wn].view(cout, cin, kh, kw)
                db = flat[bo:bo + bn]
                wcfg = net._pl_tile("wgrad", op, n, shapes)
                flops = 2.0 * n * shapes[op["dst"]][1] * shapes[op["dst"]][2] * cout * cin * kh * kw
                gs = None if (op.get("s2d") and "__stem_f32__" in grads) else PSlice(grads[op["dst"]], op["dst_c0"], cout)
                if op.get("s2d") and "__stem_f32__" in grads:
                    g32, xs32 = grads["__stem_f32__"], acts["data_s2d_f32"]
                    from .bninception import tuned_tile
                    ocfg = tuned_tile("wgrad6s2d", n, cin, cout, op["k"], op["s"], shapes[op["src"]][1])

                    def run_wgrad(ws=ws_of(op)):
                        dw2 = torch.empty((cout, 4 * cin, 4, 4), device=dev, dtype=torch.float32)
                        K.conv_wgrad_x6(K.full(g32), K.full(xs32), dw2, db, 4, 2, ws, ocfg)
                        K.s2d_weights_bwd(dw2, dw)
                elif op.get("s2d"):
                    xs = acts["data_s2d"]

                    def run_wgrad(ws=ws_of(op)):
                        dw2 = torch.empty((cout, 4 * cin, 4, 4), device=dev, dtype=torch.float32)
                        if group:      # a problem of the grouped launch (the stem body: 16 taps as 8 tap pairs on 16-channel sub-blocks)
                            pending_wgrad.append((P.WgradJob(gs, PSlice(xs, 0, xs.g * 8), dw2, db, 4, 4, 1, 2, 2, cin=4 * cin), flops))
                            pending_post.append(lambda: K.s2d_weights_bwd(dw2, dw))
                            return
                        P.conv_wgrad(gs, PSlice(xs, 0, xs.g * 8), dw2, db, 4, 4, 1, 2, 2, ws, wcfg, cin=4 * cin)
                        K.s2d_weights_bwd(dw2, dw)
                else:
                    xin = PSlice(acts[op["src"]], op["src_c0"], cin) if op["src"] != "data" else PSlice(acts["data"], 0, acts["data"].g * 8)
                    grouped = group
                    im2col = grouped and op["src"] == "data" and cin < 8 and kh * kw > 1 and net.first_conv_im2col

                    def run_wgrad(ws=ws_of(op)):
                        if im2col:
                            # a first convolution on the caller's frames (3 channels, k x k taps): as a 1x1 problem on the im2col of the
                            # frames (C k k channels, planes copied as they are) -- on the one-tap body every tap re-reads the whole
                            # output gradient for a 3-channel operand (Inception-v3's 3 -> 32 layer at 299 x 299: 1.9 ms, 3 TF)
                            _, ho_, wo_ = shapes[op["dst"]]
                            xc = P.im2col(PSlice(acts["data"], 0, cin), kh, kw, s, ph, pw, ho_, wo_)
                            pending_wgrad.append((P.WgradJob(gs, P.pfull(xc), dw.view(cout, cin * kh * kw, 1, 1), db, 1, 1, 1, 0, 0), flops))
                            return
                        if grouped:      # recorded; launched with the other weight gradients of the pass at the next flush
                            pending_wgrad.append((P.WgradJob(gs, xin, dw, db, kh, kw, s, ph, pw, cin=cin, g_row_split=op.get("row_split", 0),
                                                             g_row_gap=op.get("row_gap", 0), hint=net._pl_tile("wgradg", op, n, shapes)),
                                                  flops))
                            return
                        P.conv_wgrad(gs, xin, dw, db, kh, kw, s, ph, pw, ws, wcfg, cin=cin, g_row_split=op.get("row_split", 0),
                                     g_row_gap=op.get("row_gap", 0), defer=defer)
                def run_wgrad_and_bias(run_wgrad=run_wgrad, op=op, db=db, cout=cout):
                    run_wgrad()
                    if (raw and not op.get("bn_train")) or "raw_from" in op:
                        # the (projection's) bias sits behind the pool: its gradient is the sum of the gradient BEFORE the pool's
                        # backward (after the weight gradient's reduction, which wrote the sum of the pooled gradient there)
                        fin = op["proj_final"] if "raw_from" in op else op["final"]
                        cp = cout - op.get("raw_from", 0)

                        entry = (PSlice(grads[fin[0]], fin[1], cp), db[op.get("raw_from", 0):])
                        if defer is not None or group:
                            pending_sums.append(entry)
                        else:
                            P.channel_sum_multi([entry], cs_ws)
                if group and not (op.get("s2d") and "__stem_f32__" in grads):
                    run_wgrad_and_bias()         # (only records: the group is timed as a whole at its flush)
                else:
                    net._timed("conv_wgrad_pl", lids[0], flops, run_wgrad_and_bias)
                if op["src"] != "data":
                    wt = packed_dg[lids[0]]
                    key = src_key(op)
                    acc_flag = key in inited
                    my, ms = mask_args(idx, op, cin)
                    dx = PSlice(gbuf(op["src"]), op["src_c0"], cin)
                    tcfg = net._pl_tile("dgrad", op, n, shapes)
                    if dg_s2[lids[0]]:
                        net._timed("conv_dgrad_pl", lids[0], flops,
                                   lambda: P.conv_dgrad_s2(gs, wt, dx, op["p"], acc_flag, tcfg, mask=my, mask_scale=ms))
                    else:
                        # (a fused block-input launch reads its rows behind the split k_gap channels further up dy's tensor)
                        net._timed("conv_dgrad_pl", lids[0], flops, lambda: P.conv_dgrad(
                            gs, wt, dx, kh, kw, ph, pw, acc_flag, tcfg, mask=my, mask_scale=ms, k_split=op.get("row_split", 0),
                            k_gap=op.get("row_gap", 0), taps_reversed=op["rect"]))
                    inited.add(key)
                closes_block = (lids[0].endswith("_1x1") or lids[0] == first_conv
                                or lids[0] in ("inception_3c_3x3_reduce", "inception_4e_3x3_reduce"))
                if fire_hook and net.grad_ready_hook is not None and closes_block:
                    # grouped weight gradients: a flush per block would undo the grouping (ten small groups: measured +1 ms per step
                    # in `--collectives overlapped`); the recorded problems are flushed -- and their range handed to the reducer --
                    # once they hold a third of the pass's weight-gradient work, i.e. three all-reduce rounds per backward
                    if not (group and pending_wgrad) or sum(f for _, f in pending_wgrad) >= wgrad_flops_total / 3.0 or wo == 0:
                        flush()                               # the range's gradients must be final before their all-reduce
                        net.grad_ready_hook.range_ready(flat, wo, pending_end)
                        pending_end = wo

        side_ops, joins = _branch_lanes(plan) if (net.branch_lanes and dfeat.is_cuda) else ({}, set())
        side = net._side_stream(dev) if side_ops else None
        main = torch.cuda.current_stream(dev) if side_ops else None
        # (backward = the forward's lanes in reverse: the short chain's LAST op -- the pool, three ops behind its "fork" -- forks
        # behind the next block's input gradient, the block-input launch joins)
        forks_bwd = {i + 3 for i, kind in side_ops.items() if kind == "fork"}
        joins_bwd = {i - 1 for i, kind in side_ops.items() if kind == "fork"}
        for idx in range(len(plan) - 1, -1, -1):
            if idx in joins_bwd:
                main.wait_stream(side)
            if idx in side_ops:
                gbuf(plan[idx]["src"])                  # (allocations on the caller's stream)
                if idx in forks_bwd:
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    run_op(idx)
            else:
                run_op(idx)
        if side_ops:
            main.wait_stream(side)
        flush()
        if fire_hook and net.grad_ready_hook is not None:
            if pending_end > 0:
                net.grad_ready_hook.range_ready(flat, 0, pending_end)
            net.grad_ready_hook.finish()

    grads = {}
    hook_obj = net.grad_ready_hook if hook else None
    # a reducer that overlaps its all-reduces with the backward needs the block-by-block ready ranges DURING a pass; a deferred
    # one (parallel.GradReducer(deferred=True)) and the chunked caller only need to be told once, at the end
    overlapping = hook_obj is not None and not getattr(hook_obj, "deferred", False)
    capturing = _capturing(dfeat)
    polling = net.scale_guard == "sync" and not capturing
    refire = False

    def silent_pass():
        launch_all(grads, False)

    st.nonfinite = False
    if not st.bwd_calibrated:
        # no history: every gradient tensor starts from the magnitude of the incoming feature gradient spread over the 7x7 pool
        # (2^11 at that magnitude: f16 then covers 2^-25 .. 2^5 of it), then passes until no scale moves
        if capturing:
            raise RuntimeError("planes executor: the first backward of a state calibrates its scales with host syncs and cannot "
                               "be captured into a hipGraph; run one eager step first")
        amax0 = float(dfeat.abs().max().item())
        if not math.isfinite(amax0):
            amax0 = 0.0              # (a non-finite feature gradient: the pass below records it and settle() gives up on it)
        hw = 1
        for op in plan:
            if op["kind"] == "gap":
                hw = shapes[op["src"]][1] * shapes[op["src"]][2]
        s0 = 2.0 ** math.floor(math.log2(2048.0 / max(amax0 / hw, 1e-30))) if amax0 > 0 else 1.0
        st.pool.scale[1::2] = s0
        silent_pass()
        st.calibration_passes[1] = 1 + st.settle(silent_pass, "gradient")
        st.bwd_calibrated = True
        refire = overlapping        # the last pass stands numerically; an overlapping reducer gets one more, with its ranges
    else:
        # (the update at the head of this step's forward already derived the gradient scales from the last backward)
        launch_all(grads, overlapping)
        if net.scale_guard != "off":
            st.check()              # range guard (see run_forward); captured: the owner of the graph polls the word
            local = bool(st.fault()) if polling else False
            fault = local
            if polling and overlapping and hasattr(hook_obj, "agree"):
                # this pass has already issued collectives: whether it is repeated must be ONE decision of all ranks
                fault = hook_obj.agree(local)
            if local:
                st.describe_fault("backward")
            if fault:
                repeats = st.settle(silent_pass, "gradient")     # (local: a rank that had no fault of its own repeats nothing here)
                st.recalibrations[1] += (1 if local else 0) + repeats
                refire = overlapping                             # ... and every rank issues the pass with its collectives again
    if hook_obj is not None:
        if refire:
            launch_all(grads, True)
        elif not overlapping:
            hook_obj.range_ready(flat, 0, total)
            hook_obj.finish()
    if st.nonfinite or extras.get("nonfinite"):
        # (see run_forward: what fp32 storage would have produced -- the backward of a forward that held inf / NaN is non-finite whatever
        # the incoming gradient; the fault word is raised)
        st.pool.flag[0] = 1
        flat.fill_(float("nan"))
        for dg, db_ in bn_grads.values():
            dg.fill_(float("nan"))
            db_.fill_(float("nan"))
    out = []
    for lid in net._conv_ids:
        conv = getattr(net, lid)
        wo, wn, bo, bn = lay[lid]
        out.append(flat[wo:wo + wn].view_as(conv.weight) if conv.weight.requires_grad else None)
        out.append(flat[bo:bo + bn] if conv.bias.requires_grad else None)
    for lid in net._train_bn_ids():
        bnm = getattr(net, lid + "_bn")
        dgamma, dbeta = bn_grads[lid]
        out.append(dgamma if bnm.weight.requires_grad else None)
        out.append(dbeta if bnm.bias.requires_grad else None)
    return out, flat


def export_decisions(net, saved):
    """bninception.BNInception.export_decisions for a planes forward: the ReLU decision is the sign of the HIGH plane (what the
    fused backward epilogues read), the pool decision the stored window-local argmax."""
    plan, shapes, acts, argmax, _tscale, _packed, _st = saved

    def hi_nchw(name):
        t = acts[name]
        return t.data[0].permute(0, 1, 3, 2).reshape(t.n, t.g * 8, t.h, t.w)

    relu, pool = {}, {}
    for op in plan:
        if op["kind"] == "conv":
            off = 0
            for lid, c in zip(op["lids"], op["couts"]):
                if "raw_from" in op and off >= op["raw_from"]:
                    name, c0 = op["proj_final"]
                elif op.get("raw"):
                    name, c0 = op["final"]
                else:
                    name, c0 = op["dst"], op["dst_c0"] + off + (op["row_gap"] if off >= op.get("row_split", 1 << 30) else 0)
                relu[lid] = hi_nchw(name)[:, c0:c0 + c] > 0
                off += c
        elif op["kind"] == "pool" and op["pool"] == "max":      # (a plain average pool -- plans with training-mode BatchNorm -- decides nothing)
            am = argmax[op["lid"]]
            _, ho, wo = shapes[op["dst"]]
            pool[op["lid"]] = am.permute(0, 1, 3, 2).reshape(am.shape[0], am.shape[1] * 8, ho, wo).long()
    return relu, pool
