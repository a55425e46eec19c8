"""BN-Inception backbone executor: the stand-in for ``model_zoo.BNInception``.

The reference builds its backbone with ``getattr(model_zoo, 'BNInception')()``
(/root/reference/ssn_models.py:121-124) and runs it through torch/cuDNN.  Here the module keeps
the same parameter surface (``<layer>.weight/.bias``, ``<layer>_bn.weight/.bias/.running_*``,
``fc``) so checkpoints, ``get_optim_policies`` and the first-conv surgery
of the reference keep working, but ``forward`` is ONE autograd node that walks the manifest of
``bninception_spec`` and launches the gfx950 kernels of include/ssn_hip.h:

  forward : bn_fold -> conv_bn_relu_fwd per conv (branches write straight into their channel
            slice of the block output: no concat), pool_fwd, global_avgpool_fwd
  backward: per conv  relu_bn_bwd (in place on the output gradient) -> conv_wgrad (+bias grad)
            -> conv_dgrad (accumulating into the input gradient when the input feeds several
            branches); pool_bwd / global_avgpool_bwd for the pools.

Gradients of all conv parameters are produced into one flat buffer (forward layer order), so a
data-parallel wrapper can all-reduce finished tail ranges while earlier layers are still in
backward (see parallel.py).
"""
import json
import os

import torch
from torch import nn

from . import kernels as K
from .bninception_spec import FEATURE_DIM, build_manifest
from .kernels import ChanSlice, full


def _load_tuned():
    """Per-shape tile choices measured on the MI355X by tools/autotune.py (optional file)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_tiles.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get("tiles", {}), int(d.get("n_images", 0)), d.get("ms", {})
    except (OSError, ValueError):
        return {}, 0, {}


_TUNED, _TUNED_N, _TUNED_MS = _load_tuned()


def _load_tuned_pl():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_tiles_pl.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


_TUNED_PL = _load_tuned_pl()


def tuned_tile(kind, n, cin, cout, k, s, hin):
    """Tile config for a conv launch: the autotuned entry when the batch is comparable, else -1 (heuristic)."""
    if not _TUNED or n * 2 < _TUNED_N:
        return -1
    return _TUNED.get("%s|%d|%d|%d|%d|%d" % (kind, cin, cout, k, s, hin), -1)


def x6_wins(kind, cin, cout, k, s, hin):
    """Split-operand ("x6": 2 x f16, 3 products) kernel or the exact-f32 MFMA kernel for this layer?  x6 unless f32 measured faster."""
    key = "|%d|%d|%d|%d|%d" % (cin, cout, k, s, hin)
    t6, t32 = _TUNED_MS.get(kind + "6" + key), _TUNED_MS.get(kind + key)
    return t6 is None or t32 is None or t6 <= t32


def conv_taps(op):
    """(kh, kw, pad_h, pad_w) of a plan convolution: square layers carry k / p only, the rectangular ones of Inception-v3
    (5x5, 1x7, 7x1, 1x3, 3x1; csrc/conv_x6_rect.hip) carry the per-axis values."""
    return op.get("kh", op["k"]), op.get("kw", op["k"]), op.get("ph", op["p"]), op.get("pw", op["p"])


def is_rect(op):
    kh, kw, _, _ = conv_taps(op)
    return kh != kw or kh not in (1, 3, 7)


def _recalibrate_after_load(module, _incompatible_keys):
    module.recalibrate()


class _BackboneFn(torch.autograd.Function):
    """The whole backbone as one autograd node.  A batch whose largest activation tensor would exceed what one kernel operand
    can address (32-bit buffer offsets: 2 GiB, 668 frames of 224 x 224) runs as consecutive sub-batches -- forward and backward
    per sub-batch, the flat conv-gradient buffers summed (ssn_add_inplace) -- with the same results as one pass (frozen
    BatchNorm: the images are independent)."""

    @staticmethod
    def forward(ctx, x, net, grad_mode, *params):
        # (needs_input_grad is set from requires_grad alone, and grad mode is always off in here -- the caller passes it: inside
        # torch.no_grad() -- validation, dense testing -- nothing will ever call backward, so nothing is kept: no argmax tensors, no
        # fp32 copy of the stem input, and the inference cache applies)
        need_grad = (grad_mode and any(ctx.needs_input_grad)) or net.debug_keep_saved
        bounds = net._chunk_bounds(x)
        if len(bounds) > 2 and net._train_bn_ids():
            raise NotImplementedError("a batch of %d frames needs chunked execution (2 GiB per kernel operand), which would change "
                                      "the batch statistics of the training-mode BatchNorm layers; use a smaller batch" % x.shape[0])
        feats, saved = [], []
        for i0, i1 in zip(bounds[:-1], bounds[1:]):
            f, s = net._run_forward(x[i0:i1], keep=need_grad)
            feats.append(f)
            saved.append(s)
        if net.debug_keep_saved:      # tests: lets the caller export the forward's ReLU / max-pool decisions (export_decisions)
            net._last_saved = saved
        ctx.net = net
        ctx.saved = saved if need_grad else None
        ctx.bounds = bounds
        ctx.n_params = len(params)
        return feats[0] if len(feats) == 1 else torch.cat(feats)

    @staticmethod
    def backward(ctx, dfeat):
        net, saved, bounds = ctx.net, ctx.saved, ctx.bounds
        ctx.saved = None
        if saved is None:
            raise RuntimeError("backbone forward ran without grad bookkeeping")
        dfeat = dfeat.contiguous()
        if len(saved) == 1:
            grads, _ = net._run_backward(dfeat, saved[0])
            return (None, None, None) + tuple(grads)
        # sub-batches: per-chunk backward without the gradient-ready hook, one summed flat buffer, then the hook once
        total = None
        for k, (i0, i1) in enumerate(zip(bounds[:-1], bounds[1:])):
            grads, flat = net._run_backward(dfeat[i0:i1], saved[k], hook=False)
            saved[k] = None
            if total is None:
                total, out = flat, grads          # the returned views alias the first chunk's buffer, which takes the sum
            else:
                K.add_(total, flat)
        if net.grad_ready_hook is not None:
            net.grad_ready_hook.range_ready(total, 0, total.numel())
            net.grad_ready_hook.finish()
        return (None, None, None) + tuple(out)


class BNInception(nn.Module):
    """Drop-in for ``model_zoo.BNInception`` (ctor signature of the upstream zoo: num_classes)."""

    def __init__(self, num_classes=1000, in_channels=3, input_size=224):
        super().__init__()
        self.in_channels = in_channels
        self.input_size_hint = input_size
        ops, _ = build_manifest(in_channels, input_size)
        self._conv_ids = []
        for op in ops:
            if op[0] == "conv":
                _, lid, _, _, _, cin, cout, k, s, p = op
                setattr(self, lid, nn.Conv2d(cin, cout, k, s, p, bias=True))
                setattr(self, lid + "_bn", nn.BatchNorm2d(cout, eps=1e-5))
                self._conv_ids.append(lid)
        self.fc = nn.Linear(FEATURE_DIM, num_classes)
        self._init_executor()

    def _init_executor(self):
        self.grad_ready_hook = None   # object with range_ready(flat, start, end) / finish() (parallel.GradReducer)
        self._ws = None
        self._side = {}               # device -> side HIP stream for the weight-gradient chain
        self._lanes = {}              # device -> two more streams for the forward branches of a block
        # run wgrad launches on a second stream, concurrently with the dgrad chain
        self.overlap_wgrad = os.environ.get("SSN_OVERLAP_WGRAD", "1") != "0"
        # forward: the branches of an Inception block run on three HIP streams
        self.branch_streams = os.environ.get("SSN_BRANCH_STREAMS", "1") != "0"
        self.profiler = None          # list; when set, every conv launch is bracketed by HIP events
        # "split": 1x1/3x3 convolutions (forward, dgrad, wgrad) multiply on the f16 matrix cores with every fp32 operand
        # scaled per tensor and split into two f16 terms, three products per multiply (fp32-class accuracy,
        # csrc/conv_x6.hip); "f32": exact-f32 MFMA
        self.conv_precision = "split"
        self.wgrad_x6 = True          # weight gradients on the split kernel too (False: exact-f32 MFMA wgrad; bisecting aid)
        # the pool projection of a block rides in the launch of its reduce pair (see _move_avg_pools); False = own launch
        self.merge_projection = os.environ.get("SSN_MERGE_PROJ", "1") != "0"
        # the 7x7 / stride-2 stem through its space-to-depth form on the split kernels (False: exact-f32 MFMA kernel)
        self.stem_s2d = os.environ.get("SSN_STEM_S2D", "1") != "0"
        # average-pool branches: pool BEHIND the 1x1 projection (see _move_avg_pools); False = the manifest's order
        self.pool_after_projection = os.environ.get("SSN_POOL_ORDER", "") != "manifest"
        # "planes": activations and their gradients live as two f16 planes in the channel-blocked layout of the matrix cores,
        # produced by the kernels that compute them (planes_exec.py; frozen BatchNorm only) -- "f32": fp32 NCHW tensors, every
        # consumer converts its operands (the round-1/2 executor below, and the path of training-mode BatchNorm)
        # Default: planes for both backbones (round 4: Inception-v3 too); a plan with training-mode BatchNorm falls back to the fp32
        # layout on its own (planes_exec.supported), SSN_LAYOUT=f32 forces it (the cross-check configuration).
        self.layout = os.environ.get("SSN_LAYOUT", "planes")
        # planes_exec: every weight gradient keeps its split-K slabs in its own workspace region and ONE launch reduces them all at the
        # end of the pass (at every gradient-ready range with an overlapping reducer) instead of one launch per layer
        self.defer_wgrad_reduce = os.environ.get("SSN_DEFER_WGRAD_REDUCE", "1") != "0"
        # planes_exec: ALL weight gradients of a backward pass as one grouped call (<= 5 launches over a device-resident problem table
        # + one reduction, csrc/wgrad_pl.hip: ssn_conv_wgrad_pl_group) at the end of the pass -- or, with an overlapping gradient
        # reducer, at every gradient-ready range -- instead of one launch (+ reduction) per layer; 0: per-layer launches
        self.group_wgrad = os.environ.get("SSN_GROUP_WGRAD", "1") != "0"
        # planes_exec: the stem's weight gradient as a problem of the grouped launch on planes operands (wgrad_stem_body); 0: the
        # fp32-layout kernel of rounds 2 - 4 (fp32 space-to-depth copy of the frames + fp32 output gradient from the pool's backward)
        self.stem_planes = os.environ.get("SSN_STEM_PLANES", "1") != "0"
        # planes_exec: weight gradient of a first convolution that is NOT in space-to-depth form (Inception-v3's 3x3 / 2) as a 1x1
        # problem on the im2col of the frames (ssn_pl_im2col)
        self.first_conv_im2col = os.environ.get("SSN_FIRST_CONV_IM2COL", "1") != "0"
        # planes_exec (training passes with the stem on planes): the frames are stored with last pass's scale, like every activation
        self.delayed_input_scale = os.environ.get("SSN_DELAYED_INPUT_SCALE", "1") != "0"
        if os.environ.get("SSN_GROUP_TUNING"):      # tooling: planner constants "fixed9,fixed1,min9,min1" (ssn_conv_wgrad_pl_group_tuning)
            from . import _lib
            f9, f1, m9, m1 = (os.environ["SSN_GROUP_TUNING"].split(",") + ["0"] * 4)[:4]
            _lib.get_lib().cdll.ssn_conv_wgrad_pl_group_tuning(float(f9), float(f1), int(m9), int(m1))
        # planes_exec: the two independent branch chains of an Inception block (3x3 + pool | double 3x3) on two streams -- parallel
        # branches of the captured step; the tails and ramps of the 200 - 900-workgroup launches at 14 x 14 / 7 x 7 overlap.
        # Measured (profiles/r5_branch_lanes_ab.txt, three alternating pairs on one box): 16.20 -> 15.96 ms per step (-1.4 %).
        # Results are unchanged bit for bit: the lanes write disjoint slices, amax slots are raised by atomic max (order-free),
        # scales are the previous step's.
        self.branch_lanes = os.environ.get("SSN_BRANCH_LANES", "1") != "0"
        self._side_streams = {}
        self.infer_cache = os.environ.get("SSN_INFER_CACHE", "1") != "0"   # planes_exec: packed weights / folded BN reused across no-grad forwards
        self.pooled_mask = os.environ.get("SSN_POOLED_MASK", "1") != "0"   # planes_exec: stem pools' backward reads the pooled sign
        # planes_exec: all weight operands of a pass packed in three launches through a device-resident plan (kernels.PackBatch)
        self.batch_packing = os.environ.get("SSN_BATCH_PACKING", "1") != "0"
        # planes_exec: training-mode BatchNorm layers (bn_mode 'partial' / 'full') on the planes kernels of csrc/planes_bn.hip;
        # 0: plans with such layers run on the fp32-layout executor below (csrc/bn_train.hip)
        self.planes_train_bn = os.environ.get("SSN_PLANES_TRAIN_BN", "1") != "0"
        self.debug_keep_saved = False
        self._last_saved = None
        self._planes_states = {}
        self._planes_flags = {}       # device -> int32[2]: the range guard's fault word (+ the calibration loops' move counter)
        # range guard of the planes path's delayed scales (planes_exec.py): "sync" = every eager pass polls the fault word and
        # repeats itself with fresh scales before anything reads its result (one 8-byte read per pass); "deferred" = the check is
        # only launched -- the caller polls scale_fault() / calls recalibrate() (what a hipGraph capture gets in any case); "off"
        self.scale_guard = os.environ.get("SSN_SCALE_GUARD", "sync")
        if self.scale_guard not in ("sync", "deferred", "off"):
            raise ValueError("SSN_SCALE_GUARD must be sync, deferred or off")
        # weights loaded after the first forward change every activation's magnitude: calibrate again
        self.register_load_state_dict_post_hook(_recalibrate_after_load)
        self.pl_tiles = {}            # (kind, cin, cout, kh, kw, s, hin) -> tile config of the planes kernels (autotuner)

    # ------------------------------------------------------------------ range guard of the planes path (planes_exec.py)
    def planes_flag(self, device):
        """int32[2] on `device`: [0] = fault word of the delayed scales (bit 0: a tensor was clamped, bit 1: one fell far below its
        scale; sticky until cleared), shared by all executor states of this backbone on the device.  Hand it to
        ``SSNSGD.step(skip_flag=)`` so that a flagged step's update is skipped on the device."""
        device = torch.device(device)
        f = self._planes_flags.get(device)
        if f is None:
            f = self._planes_flags[device] = torch.zeros(2, device=device, dtype=torch.int32)
        return f

    def scale_fault(self):
        """True if a pass since the last clear left the range of its scales (host sync).  Eager passes under
        ``scale_guard == "sync"`` repair themselves and never leave this set; a graph replay may."""
        return any(int(f[0].item()) != 0 for f in self._planes_flags.values())

    def recalibrate(self):
        """Clear the fault word and make the next forward / backward of every state calibrate from scratch (eagerly)."""
        self.__dict__.pop("_infer_cache", None)
        for st in self._planes_states.values():
            st.fwd_calibrated = st.bwd_calibrated = False
        for f in self._planes_flags.values():
            f.zero_()

    def guard_stats(self):
        """{"fwd": passes repeated by the range guard in forwards, "bwd": ... in backwards} over all states."""
        return {"fwd": sum(st.recalibrations[0] for st in self._planes_states.values()),
                "bwd": sum(st.recalibrations[1] for st in self._planes_states.values())}

    def _timed(self, family, lid, flops, fn):
        """Run one conv launch; with a profiler attached, bracket it with events on the current stream."""
        if self.profiler is None:
            fn()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.profiler.append((family, lid, flops, s, e))

    # ------------------------------------------------------------------ parameter plumbing
    def _param_list(self):
        ps = []
        for lid in self._conv_ids:
            conv = getattr(self, lid)
            ps.append(conv.weight)
            ps.append(conv.bias)
        for lid in self._train_bn_ids():       # bn_mode 'partial' / 'full': gamma / beta of the BatchNorms in training mode
            bn = getattr(self, lid + "_bn")
            ps.append(bn.weight)
            ps.append(bn.bias)
        return ps

    def _train_bn_ids(self):
        """Layers whose BatchNorm2d is in training mode (SSN.train() leaves the first one -- bn_mode 'partial' -- or all of
        them -- 'full' -- there, /root/reference/ssn_models.py:95-105,156-174); forward order."""
        return [lid for lid in self._conv_ids if getattr(self, lid + "_bn").training]

    def flat_grad_layout(self, plan=None):
        """[(layer id, weight offset, weight numel, bias offset, bias numel)], total.

        Forward (plan) order; the convolutions of ONE fused launch -- the two 1x1 "reduce" convolutions of a block,
        plus the block's pool projection when it rides along -- sit as [w_a | w_b | (w_p) | b_a | b_b | (b_p)], so
        the fused wgrad writes all weight gradients (and all bias gradients) as one contiguous matrix.
        """
        if plan is None:
            plan, _ = self._plan(torch.zeros(1, getattr(self, self._conv_ids[0]).in_channels, self.input_size_hint,
                                             self.input_size_hint))
        off, lay = 0, []
        for op in plan:
            if op["kind"] != "conv":
                continue
            convs = [getattr(self, lid) for lid in op["lids"]]
            wo = off
            for lid, conv in zip(op["lids"], convs):
                lay.append([lid, wo, conv.weight.numel(), None, conv.bias.numel()])
                wo += conv.weight.numel()
            for ent, conv in zip(lay[-len(convs):], convs):
                ent[3] = wo
                wo += conv.bias.numel()
            off = wo
        return [tuple(e) for e in lay], off

    max_operand_bytes = (1 << 31) - (1 << 20)     # what a raw buffer descriptor addresses, less the guard floats

    def _chunk_bounds(self, x):
        """[0, n1, ..., N]: sub-batches whose largest activation tensor stays below max_operand_bytes (one chunk = all of
        it for the batches of the reference's configurations)."""
        _, shapes = self._plan(x)       # (memoised; its shapes are a superset of the manifest's)
        per_image = 4 * max(v[0] * v[1] * v[2] for v in dict(shapes).values() if isinstance(v, tuple))
        per_image = max(per_image, 4 * x.shape[1] * x.shape[2] * x.shape[3])
        n, cap = x.shape[0], max(1, self.max_operand_bytes // per_image)
        parts = (n + cap - 1) // cap
        return [round(i * n / parts) for i in range(parts + 1)] if n else [0, 0]

    def features(self, x):
        if x.dim() != 4:
            raise ValueError("expected NCHW input")
        return _BackboneFn.apply(x.contiguous(), self, torch.is_grad_enabled(), *self._param_list())

    def forward(self, x):
        return self.fc(self.features(x))

    # ------------------------------------------------------------------ execution plan
    def _manifest(self, x):
        cin = getattr(self, self._conv_ids[0]).in_channels  # may differ after flow surgery
        if x.shape[1] != cin:
            raise ValueError("input has %d channels, first conv expects %d" % (x.shape[1], cin))
        if x.shape[2] != x.shape[3]:
            raise ValueError("square inputs only")
        return build_manifest(cin, x.shape[2])

    def _plan(self, x):
        """The launch plan for inputs of x's shape, memoised: building it costs 1 - 2 ms of Python per call, which an eager
        forward that ends in the range guard's poll (planes_exec) exposes on every call -- 10 % of a dense-test video."""
        key = (tuple(x.shape[1:]), tuple(self._train_bn_ids()), self.pool_after_projection, self.merge_projection,
               self.conv_precision, getattr(self, "fuse_block_inputs", None), getattr(self, self._conv_ids[0]).in_channels)
        cache = self.__dict__.setdefault("_plan_cache", {})
        hit = cache.get(key)
        if hit is None:
            hit = cache[key] = self._build_plan(x)
        return hit

    def _build_plan(self, x):
        """Manifest -> launch plan.

        Plan ops are dicts.  The two 1x1 reduce convolutions of an Inception block read the same input;
        they are planned as ONE convolution with concatenated output channels writing a shared
        "<block>_reduce" tensor, whose channel slices the following 3x3 convolutions read.  That halves the
        launches on the block input in all three passes and, in dgrad, replaces two read-modify-write
        sweeps over the input gradient by one with twice the K.
        """
        ops, shapes = self._manifest(x)
        shapes = dict(shapes)
        plan = []
        alias = {}      # manifest tensor name -> (plan tensor name, channel offset)
        skip = set()
        for i, op in enumerate(ops):
            if i in skip:
                continue
            if op[0] == "conv":
                _, lid, src, dst, c0, cin, cout, k, s, p = op
                psrc, sc0 = alias.get(src, (src, 0))
                mate_id = lid.replace("_3x3_reduce", "_double_3x3_reduce")
                mate = None
                if lid.startswith("inception_") and lid.endswith("_3x3_reduce") and "double" not in lid:
                    for j in range(i + 1, min(i + 4, len(ops))):
                        if ops[j][0] == "conv" and ops[j][1] == mate_id and ops[j][2] == src:
                            mate = (j, ops[j])
                if mate is not None:
                    j, mop = mate
                    skip.add(j)
                    red = lid[:-len("3x3_reduce")] + "reduce"
                    ca, cb = cout, mop[6]
                    shapes[red] = (ca + cb, shapes[dst][1], shapes[dst][2])
                    alias[dst] = (red, 0)
                    alias[mop[3]] = (red, ca)
                    plan.append(dict(kind="conv", lids=[lid, mate_id], src=psrc, src_c0=sc0, cin=cin, dst=red,
                                     dst_c0=0, cout=ca + cb, couts=[ca, cb], k=k, s=s, p=p))
                else:
                    plan.append(dict(kind="conv", lids=[lid], src=psrc, src_c0=sc0, cin=cin, dst=dst, dst_c0=c0,
                                     cout=cout, couts=[cout], k=k, s=s, p=p))
            elif op[0] == "pool":
                _, lid, kind, src, dst, c0, k, s, p, _ceil = op
                assert src not in alias
                plan.append(dict(kind="pool", lid=lid, pool=kind, src=src, dst=dst, dst_c0=c0, c=shapes[src][0],
                                 k=k, s=s, p=p))
            else:
                _, lid, src, dst = op
                plan.append(dict(kind="gap", lid=lid, src=src, dst=dst, c=shapes[src][0]))
        train_bn = set(self._train_bn_ids())
        # (bn_mode 'partial': only the stem's BatchNorm takes batch statistics -- the block rewrites below never touch that layer)
        if self.pool_after_projection and not (train_bn - {self._conv_ids[0]}):
            merge = self.conv_precision == "split" and self.merge_projection
            plan = self._move_avg_pools(plan, shapes, merge=merge)
            if merge:
                plan = self._merge_block_heads(plan, shapes)
        if train_bn:
            plan = self._split_train_bn(plan, shapes, train_bn)
        return plan, shapes

    @staticmethod
    def _merge_block_heads(plan, shapes):
        """ONE launch on an Inception block input: the block's 1x1 branch joins its fused reduce (+ projection) launch.  The
        1x1 branch must land at the head of the block-output tensor, the reduce rows in a tensor of their own -- so the reduce
        tensor moves INTO the block-output allocation, behind the block's own channels: output rows >= row_split are stored
        row_gap channels further up (conv_epilogue.h), the dgrad / wgrad of the launch read the gradient tensor with the same
        displacement (k_split / k_gap, g_row_split / g_row_gap).  The block input is then read once per pass and its gradient
        has a single writer (no read-modify-write, the producer's ReLU/BN backward fused into that one store)."""
        out, moved = [], {}       # moved: reduce tensor -> (block-output tensor, channel offset)
        for op in plan:
            if (op["kind"] == "conv" and len(op["lids"]) == 1 and op["lids"][0].endswith("_1x1") and op["k"] == 1
                    and op["dst_c0"] == 0 and not op.get("raw")):
                pair = next((q for q in plan if q["kind"] == "conv" and len(q["lids"]) >= 2 and q["src"] == op["src"]
                             and q.get("src_c0", 0) == op.get("src_c0", 0) and q["k"] == 1 and q["dst_c0"] == 0), None)
                if pair is not None and op["cout"] % 32 == 0 and shapes[op["dst"]][0] % 32 == 0:
                    blk, cblk, c1 = op["dst"], shapes[op["dst"]][0], op["cout"]
                    moved[pair["dst"]] = (blk, cblk)
                    merged = dict(pair, lids=op["lids"] + pair["lids"], couts=[c1] + pair["couts"], cout=c1 + pair["cout"],
                                  dst=blk, dst_c0=0, row_split=c1, row_gap=cblk - c1, block_channels=cblk, merged_from=pair)
                    if "raw_from" in pair:
                        merged["raw_from"] = c1 + pair["raw_from"]
                    shapes[blk] = (cblk + pair["cout"],) + tuple(shapes[blk][1:])
                    shapes.setdefault("__ext__", {})[blk] = cblk      # first channel of the rows behind the block's own
                    out.append(merged)
                    continue
            out.append(op)
        res = []
        for op in out:
            if any(op is m.get("merged_from") for m in out if "merged_from" in m):
                continue          # the reduce launch now lives in its block's head launch
            if op.get("src") in moved:
                blk, c0 = moved[op["src"]]
                op = dict(op, src=blk, src_c0=op.get("src_c0", 0) + c0)
            res.append(op)
        for m in res:
            m.pop("merged_from", None)
        for name in moved:
            shapes.pop(name, None)
        return res

    @staticmethod
    def _split_train_bn(plan, shapes, train_bn):
        """A convolution whose BatchNorm2d runs in training mode becomes: the bare convolution (no bias -- a per-channel
        constant cancels in z - mean(z) --, no affine, no ReLU) into its own tensor z, then a "bn_train" op: batch
        statistics of z, running-statistics update, y = relu(gamma * xhat + beta) into the layer's destination slice."""
        out = []
        for op in plan:
            if op["kind"] == "conv" and any(lid in train_bn for lid in op["lids"]):
                assert all(lid in train_bn for lid in op["lids"]), "fused pair with mixed BatchNorm modes: %s" % op["lids"]
                z = op["lids"][0] + "_zbn"
                shapes[z] = (op["cout"],) + tuple(shapes[op["dst"]][1:])
                out.append(dict(op, dst=z, dst_c0=0, raw=True, bn_train=True, final=(op["dst"], op["dst_c0"])))
                out.append(dict(kind="bn_train", lid=op["lids"][0], lids=op["lids"], couts=op["couts"], src=z,
                                dst=op["dst"], dst_c0=op["dst_c0"], c=op["cout"]))
            else:
                out.append(op)
        return out

    @staticmethod
    def _move_avg_pools(plan, shapes, merge=False):
        """<block>_pool (3x3 / s1 / p1 average, count_include_pad) -> <block>_pool_proj (1x1) + BN + ReLU is evaluated
        as  relu(scale * avgpool(conv1x1_nobias(x)) + shift): a 1x1 convolution commutes with a zero-padded average
        pool (both linear, the bias / folded BN shift is added after the pool either way).  The pool then runs on the
        projection's 32-128 output channels instead of the block's 192-1056 input channels (4-8x less HBM traffic in
        forward and backward), applies the affine + ReLU itself, and the projection joins the other 1x1 convolutions
        that read the block input directly.  With `merge` it becomes part of the block's fused reduce launch: output
        channels [ca + cb, ca + cb + cp) of the "<block>_reduce" tensor, marked raw (no affine, no ReLU) -- the block input is
        then read (forward, wgrad) and its gradient written (dgrad) by two launches instead of three."""
        out, i = [], 0
        while i < len(plan):
            op = plan[i]
            nxt = plan[i + 1] if i + 1 < len(plan) else None
            if (op["kind"] == "pool" and op["pool"] == "avg" and (op["k"], op["s"], op["p"]) == (3, 1, 1)
                    and nxt is not None and nxt["kind"] == "conv" and nxt["src"] == op["dst"] and nxt["k"] == 1
                    and len(nxt["lids"]) == 1
                    and sum(1 for q in plan if q["src"] == op["dst"]) == 1):
                z = nxt["lids"][0] + "_z"
                cout = nxt["cout"]
                pair = next((q for q in out if q["kind"] == "conv" and len(q["lids"]) == 2 and q["src"] == op["src"]
                             and q.get("src_c0", 0) == 0 and q["k"] == 1 and not q.get("raw")), None) if merge else None
                if pair is not None:
                    c0 = pair["cout"]
                    pair.update(lids=pair["lids"] + nxt["lids"], couts=pair["couts"] + [cout], cout=c0 + cout,
                                raw_from=c0, proj_final=(nxt["dst"], nxt["dst_c0"]))
                    shapes[pair["dst"]] = (c0 + cout,) + tuple(shapes[pair["dst"]][1:])
                    out.append(dict(kind="pool_aff", lid=op["lid"], conv=nxt["lids"][0], src=pair["dst"], src_c0=c0,
                                    dst=nxt["dst"], dst_c0=nxt["dst_c0"], c=cout, k=3, s=1, p=1))
                    i += 2
                    continue
                shapes[z] = (cout, shapes[op["src"]][1], shapes[op["src"]][2])
                out.append(dict(nxt, src=op["src"], src_c0=0, dst=z, dst_c0=0, raw=True,
                                final=(nxt["dst"], nxt["dst_c0"])))
                out.append(dict(kind="pool_aff", lid=op["lid"], conv=nxt["lids"][0], src=z, src_c0=0, dst=nxt["dst"],
                                dst_c0=nxt["dst_c0"], c=cout, k=3, s=1, p=1))
                i += 2
                continue
            out.append(op)
            i += 1
        return out

    def _pl_tile(self, kind, op, n, shapes):
        """Tile config of a planes-kernel launch: the autotuned table (tools/autotune_pl.py -> tuned_tiles_pl.json), else -1."""
        kh, kw = op.get("kh", op["k"]), op.get("kw", op["k"])
        key = "%s|%d|%d|%d|%d|%d|%d" % (kind, op["cin"], op["cout"], kh, kw, op["s"], shapes[op["src"]][1])
        tab = _TUNED_PL.get("tiles", {})
        if not tab or n * 2 < _TUNED_PL.get("n_images", 0):
            return -1
        return tab.get(key, -1)

    def export_decisions(self):
        """The discrete decisions of the last forward (debug_keep_saved = True; one chunk): ({layer id: bool [N, C, H, W] = "the
        ReLU behind this layer passed the element" as the BACKWARD of this executor sees it}, {pool id: int64 [N, C, Ho, Wo] =
        window-local index dr * k + ds of the maximum}).  tests/test_model_gpu.py runs the float64 oracle with exactly these
        decisions forced, which makes the loss smooth in the comparison: gradients must then agree to rounding."""
        saved = self._last_saved[0]
        if len(saved) == 7:
            from . import planes_exec
            return planes_exec.export_decisions(self, saved)
        plan, shapes, acts, argmax, _tscale, _bn = saved
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
                    relu[lid] = acts[name][:, c0:c0 + c] > 0
                    off += c
            elif op["kind"] == "pool" and op["pool"] == "max":
                pool[op["lid"]] = argmax[op["lid"]].long()
        return relu, pool

    def _use_planes(self, plan):
        if self.layout != "planes" or self.conv_precision != "split":
            return False
        from . import planes_exec
        return planes_exec.supported(self, plan)

    # ------------------------------------------------------------------ forward executor
    def _run_forward(self, x, keep):
        plan, shapes = self._plan(x)
        if self._use_planes(plan):
            from . import planes_exec
            return planes_exec.run_forward(self, x, keep)
        n, dev = x.shape[0], x.device
        acts = {"data": x}
        argmax, tscale, wcat = {}, {}, {}
        bnstat = {}      # training-mode BatchNorm layers: layer id -> (batch mean of z, 1 / sqrt(var + eps))
        last_use = {}
        for i, op in enumerate(plan):
            last_use[op["src"]] = i

        # one amax slot per activation tensor (kernels.py: "amax slots"): the kernels that write a tensor raise its
        # slot, the split convolutions that read it take their operand scale from it
        ext_base = shapes.get("__ext__", {})      # tensors with a second region (the reduce rows behind a block's output)
        slot_pool = torch.zeros(2 * len(shapes) + 2, device=dev, dtype=torch.float32)
        slot_of = {}
        # readable floats in front of every activation: lets the x6 kernels use 16-byte loads; the weight gradient of a
        # rectangular-tap layer reaches (pad_h * W + pad_w) floats in front of its input
        guard = 64
        for op in plan:
            if op["kind"] == "conv":
                op["rect"] = is_rect(op)
                if op["rect"]:
                    if self.conv_precision != "split":
                        raise NotImplementedError("%dx%d taps only exist on the split-precision kernels" % conv_taps(op)[:2])
                    guard = max(guard, K.wgrad_x6_rect_guard_floats(conv_taps(op)[2], conv_taps(op)[3], shapes[op["src"]][2]))
                elif op["k"] == 3 and op["s"] == 1 and op["p"] == 1 and shapes[op["src"]][2] < 256:
                    # (rows wider than 63 pixels: the tap in front of a pixel reaches further than the default 256 bytes)
                    guard = max(guard, K.wgrad_x6_rect_guard_floats(1, 1, shapes[op["src"]][2]))

        def get(name):
            if name not in acts:
                c, h, w = shapes[name]
                i = slot_of.setdefault(name, len(slot_of))
                acts[name] = K.attach_amax(K.guarded_empty((n, c, h, w), dev, guard), slot_pool[2 * i:2 * i + 1])
                if name in ext_base:
                    K.attach_amax_ext(acts[name], slot_pool[2 * i + 1:2 * i + 2], ext_base[name])
            return acts[name]

        def scale_slice(name, c0, c):
            # per tensor: folded-BN scale of every channel (NaN: channel is not a conv+ReLU output; out of band, so that a
            # negative folded scale -- gamma < 0 occurs in trained checkpoints -- keeps its sign)
            if name not in tscale:
                tscale[name] = torch.full((shapes[name][0],), float("nan"), device=dev, dtype=torch.float32)
            return tscale[name][c0:c0 + c]

        # frozen-BN folding of all 69 layers in two launches (scale into the per-tensor vectors, shift into one
        # flat buffer the conv epilogues read slices of)
        shift_flat = torch.empty(sum(op["cout"] for op in plan if op["kind"] == "conv"), device=dev,
                                 dtype=torch.float32)
        fold = ([], [], [], [], [], [], [], [])
        shift_of = {}
        soff = 0
        for op in plan:
            if op["kind"] != "conv":
                continue
            shift_of[op["lids"][0]] = shift_flat[soff:soff + op["cout"]]
            for lid_, o_ in zip(op["lids"], [sum(op["couts"][:q]) for q in range(len(op["lids"]))]):
                shift_of.setdefault(lid_, shift_flat[soff + o_:soff + o_ + getattr(self, lid_).out_channels])
            off = 0
            # (a projection whose pool runs behind it: its BN affine belongs to the slice the POOL writes)
            aff_dst, aff_c0 = op.get("final", (op["dst"], op["dst_c0"]))
            if op.get("bn_train"):
                # batch-statistics layer: nothing to fold; NaN scales = "not a frozen ReLU/BN output", so that the fused
                # backward epilogues of the consumers leave this slice's gradient alone (the bn_train op owns its backward)
                scale_slice(aff_dst, aff_c0, op["cout"]).fill_(float("nan"))
                soff += op["cout"]
                continue
            for lid, c in zip(op["lids"], op["couts"]):
                conv, bn = getattr(self, lid), getattr(self, lid + "_bn")
                if "raw_from" in op and off >= op["raw_from"]:
                    # the projection riding in the reduce launch: its BN affine belongs to the slice its POOL writes
                    sdst = scale_slice(op["proj_final"][0], op["proj_final"][1], c)
                else:     # (rows behind the split of a block-head launch live row_gap channels further up the tensor)
                    sdst = scale_slice(aff_dst, aff_c0 + off + (op["row_gap"] if off >= op.get("row_split", 1 << 30) else 0), c)
                for lst, v in zip(fold, (conv.bias.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                         bn.running_var, bn.eps, sdst, shift_flat[soff + off:soff + off + c])):
                    lst.append(v)
                off += c
            soff += op["cout"]
        K.bn_fold_multi(*fold)
        # all forward weight operands in two launches (fused pairs read both sources directly: no concatenation)
        conv_ops = [op for op in plan if op["kind"] == "conv"]
        for op in conv_ops:      # the stem: 7x7 / stride 2 / pad 3 on the caller's frames, even size
            op["s2d"] = (self.conv_precision == "split" and self.stem_s2d and op["src"] == "data" and len(op["lids"]) == 1
                         and (op["k"], op["s"], op["p"]) == (7, 2, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                         and not op.get("raw"))
        for op in conv_ops:      # which matrix path each layer takes (f16 2-way split, or exact f32 MFMA)
            op["x6"] = (self.conv_precision == "split" and op["k"] in (1, 3) and not op["rect"] and op["cin"] >= 16
                        and ("raw_from" in op or "row_gap" in op      # (raw rows / displaced rows: split kernel only)
                             or x6_wins("fwd", op["cin"], op["cout"], op["k"], op["s"], shapes[op["src"]][1])))
        packed_fwd = {}
        for op in conv_ops:
            if op["s2d"]:
                acts["data_s2d"] = K.space_to_depth2(x)
                packed_fwd[op["lids"][0]] = K.pack_weights_rect(K.s2d_weights(getattr(self, op["lids"][0]).weight.detach()))
        rect_ops = [op for op in conv_ops if op["rect"]]
        packed_fwd.update(zip((op["lids"][0] for op in rect_ops),
                              K.pack_rect_multi([getattr(self, op["lids"][0]).weight.detach() for op in rect_ops])))
        for x6 in (False, True):
            ops = [op for op in conv_ops if op["x6"] == x6 and not op["s2d"] and not op["rect"]]
            packed_fwd.update(zip((op["lids"][0] for op in ops), K.pack_weights_multi(
                [([getattr(self, lid).weight.detach() for lid in op["lids"]], 0) for op in ops], x6=x6)))

        # Branch-level concurrency: most launches of the 14x14 / 7x7 stages fill the 256 CUs less than twice over
        # (a 7x7 layer is 110-330 workgroups for 512 slots) and every launch ends with a partial round, so the three
        # independent chains of an Inception block -- [1x1, reduce pair, 3x3], [double 3x3 a, b] and [pool, projection]
        # -- go to three streams (fork at the block input, the double-3x3 chain additionally waits for the reduce pair,
        # join at the block output).  Inside a captured step these become parallel branches of the hipGraph.
        lanes = None
        if self.branch_streams and x.is_cuda:
            main = torch.cuda.current_stream(dev)
            sides = self._lanes.get(dev)
            if sides is None:
                sides = self._lanes[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            lanes = (main,) + sides

        def block_of(op):
            lid = op["lids"][0] if op["kind"] == "conv" else op.get("lid", "")
            return lid[:12] if lid.startswith("inception_") else None      # "inception_3a"

        def lane_of(op):
            lid = op["lids"][0] if op["kind"] == "conv" else op["lid"]
            if op["kind"] == "conv" and op.get("raw") and not op.get("bn_train"):
                return 2          # a projection in front of its pool (the convolution of a bn_train pair stays with its op)
            if "_double_3x3_1" in lid or "_double_3x3_2" in lid:
                return 1
            if lid.endswith("_pool") or lid.endswith("_pool_proj"):
                return 2
            return 0

        feat = None
        cur_block = None
        for i, op in enumerate(plan):
            lane_ctx = None
            if lanes is not None:
                blk = block_of(op)
                if blk != cur_block:
                    if cur_block is not None:         # join: the block output is complete
                        lanes[0].wait_stream(lanes[1])
                        lanes[0].wait_stream(lanes[2])
                    if blk is not None:               # fork: the block input is complete
                        lanes[1].wait_stream(lanes[0])
                        lanes[2].wait_stream(lanes[0])
                    cur_block = blk
                if blk is not None:
                    get(op["dst"])                    # allocate on the caller's stream, launch on the lane
                    if op["kind"] == "pool" and op["pool"] == "max" and keep:
                        _, ho, wo = shapes[op["dst"]]
                        argmax[op["lid"]] = torch.empty((n, op["c"], ho, wo), device=dev, dtype=torch.uint8)
                    lane_ctx = torch.cuda.stream(lanes[lane_of(op)])
                    lane_ctx.__enter__()
            if op["kind"] == "conv":
                cout, cin, k, s, p = op["cout"], op["cin"], op["k"], op["s"], op["p"]
                raw = bool(op.get("raw"))      # no affine / ReLU here: the pool behind this projection applies them
                shift = None if raw else shift_of[op["lids"][0]]
                wp = packed_fwd[op["lids"][0]]
                # (per-channel vector of the destination tensor from the slice's first channel; a block-head launch reaches
                #  row_gap channels further up for its rows behind the split, like its stores)
                scale = None if raw else scale_slice(op["dst"], op["dst_c0"], cout + op.get("row_gap", 0))
                ho, wo = shapes[op["dst"]][1], shapes[op["dst"]][2]
                hin = shapes[op["src"]][1]
                kh, kw, ph, pw = conv_taps(op)
                flops = 2.0 * n * ho * wo * cout * cin * kh * kw
                src_slice = ChanSlice(acts[op["src"]], op["src_c0"], cin)
                dst_slice = ChanSlice(get(op["dst"]), op["dst_c0"], cout)
                if op["rect"]:
                    self._timed("conv_fwd_x6", op["lids"][0], flops,
                                lambda: K.conv_x6_fwd_rect(src_slice, wp, scale, shift, dst_slice, kh, kw, ph, pw, not raw,
                                                           tuned_tile("fwd6r%dx%d" % (kh, kw), n, cin, cout, 0, 1, hin)))
                elif op["s2d"]:
                    xs = acts["data_s2d"]
                    self._timed("conv_fwd_x6", op["lids"][0], flops,
                                lambda: K.conv_x6_fwd_rect(full(xs), wp, scale, shift, dst_slice, 4, 4, 2, 2, True,
                                                           tuned_tile("fwd6s2d", n, cin, cout, k, s, hin)))
                elif op["x6"]:
                    self._timed("conv_fwd_x6", op["lids"][0], flops,
                                lambda: K.conv_x6_fwd(src_slice, wp, scale, shift, dst_slice, k, s, p,
                                                      not raw, tuned_tile("fwd6", n, cin, cout, k, s, hin),
                                                      raw_from=op.get("raw_from", 0), row_split=op.get("row_split", 0),
                                                      row_gap=op.get("row_gap", 0)))
                else:
                    self._timed("conv_fwd_f32", op["lids"][0], flops,
                                lambda: K.conv_fwd(src_slice, wp, scale, shift, dst_slice, k, s, p,
                                                   not raw, tuned_tile("fwd", n, cin, cout, k, s, hin)))
            elif op["kind"] == "pool":
                c = op["c"]
                out = ChanSlice(get(op["dst"]), op["dst_c0"], c)
                am = argmax.get(op["lid"])
                if am is None and op["pool"] == "max" and keep:
                    _, ho, wo = shapes[op["dst"]]
                    am = torch.empty((n, c, ho, wo), device=dev, dtype=torch.uint8)
                    argmax[op["lid"]] = am
                K.pool_fwd(op["pool"], ChanSlice(acts[op["src"]], 0, c), out, am, op["k"], op["s"], op["p"])
            elif op["kind"] == "pool_aff":
                c = op["c"]
                K.avgpool_affine_fwd(ChanSlice(acts[op["src"]], op.get("src_c0", 0), c),
                                     ChanSlice(get(op["dst"]), op["dst_c0"], c),
                                     scale_slice(op["dst"], op["dst_c0"], c), shift_of[op["conv"]], True,
                                     op["k"], op["s"], op["p"])
            elif op["kind"] == "bn_train":
                off = 0
                bws = torch.empty(K.bn_train_workspace_floats(n, op["c"]), device=dev, dtype=torch.float32)
                for lid, c in zip(op["lids"], op["couts"]):
                    conv, bn = getattr(self, lid), getattr(self, lid + "_bn")
                    mean = torch.empty(c, device=dev, dtype=torch.float32)
                    invstd = torch.empty(c, device=dev, dtype=torch.float32)
                    zs = ChanSlice(acts[op["src"]], off, c)
                    track = bn.track_running_stats and bn.running_mean is not None
                    if track and bn.num_batches_tracked is not None:
                        bn.num_batches_tracked.add_(1)      # torch increments BEFORE it updates the running statistics
                    mom = bn.momentum
                    if mom is None:
                        # torch: momentum=None is a CUMULATIVE average with factor 1 / num_batches_tracked (counted first): one host
                        # read per layer and call, as torch's own module does
                        mom = 1.0 / float(bn.num_batches_tracked.item()) if (track and bn.num_batches_tracked is not None) else 0.0
                    K.bn_train_stats(zs, conv.bias.detach(), mean, invstd, bn.running_mean if track else None,
                                     bn.running_var if track else None, bn.eps, mom, bws)
                    K.bn_train_apply(zs, ChanSlice(get(op["dst"]), op["dst_c0"] + off, c), mean, invstd,
                                     bn.weight.detach(), bn.bias.detach(), True)
                    bnstat[lid] = (mean, invstd)
                    off += c
            else:
                feat = torch.empty((n, op["c"]), device=dev, dtype=torch.float32)
                K.gap_fwd(ChanSlice(acts[op["src"]], 0, op["c"]), feat)
            if lane_ctx is not None:
                lane_ctx.__exit__(None, None, None)
                # the reduce pair feeds the double-3x3 chain (with a training-mode BatchNorm its bn_train op completes it)
                if (op["kind"] == "conv" and len(op["lids"]) >= 2 and not op.get("bn_train")) or \
                        (op["kind"] == "bn_train" and len(op["lids"]) >= 2):
                    ev = torch.cuda.Event()
                    ev.record(lanes[0])
                    lanes[1].wait_event(ev)
                    if "raw_from" in op:            # ... and, with the projection on board, the pool behind it
                        lanes[2].wait_event(ev)
            if not keep and lanes is None:
                # inference: drop activations as soon as their last consumer has been launched
                for name in [nm for nm, last in last_use.items() if last == i and nm != "data"]:
                    acts.pop(name, None)
        saved = (plan, shapes, acts, argmax, tscale, bnstat) if keep else None
        return feat, saved

    # ------------------------------------------------------------------ backward executor
    def _workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != dev:
            self._ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        return self._ws

    def _side_stream(self, dev):
        st = self._side_streams.get(dev)
        if st is None:
            st = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        return st

    def _wgrad_group_buffers(self, jobs, dev):
        """(workspace, table) of a grouped weight-gradient call: persistent buffers, grown on demand (planes_exec.run_backward)."""
        from . import planes as P
        ws_bytes, tb_bytes, _ = P.wgrad_group_plan(jobs)
        bufs = self.__dict__.setdefault("_wg_group_bufs", {})
        ws, tb = bufs.get(dev, (None, None))
        if ws is None or ws.numel() * 4 < ws_bytes + 16:
            ws = torch.empty(ws_bytes // 4 + 4, device=dev, dtype=torch.float32)
        if tb is None or tb.numel() < tb_bytes:
            tb = torch.empty(max(tb_bytes, 1 << 15), device=dev, dtype=torch.uint8)
        bufs[dev] = (ws, tb)
        return ws, tb

    def _run_backward(self, dfeat, saved, hook=True):
        if len(saved) == 7:      # a forward of the planes executor
            from . import planes_exec
            return planes_exec.run_backward(self, dfeat, saved, hook)
        plan, shapes, acts, argmax, tscale, bnstat = saved
        bn_grads = {}    # training-mode BatchNorm layers: layer id -> (dgamma, dbeta)
        n, dev = dfeat.shape[0], dfeat.device
        layout, total = self.flat_grad_layout(plan)
        lay = {lid: (wo, wn, bo, bn) for lid, wo, wn, bo, bn in layout}
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        grads = {}
        inited = set()      # (tensor, c0) gradient slices that already hold a contribution

        ext_base = shapes.get("__ext__", {})
        gslot_pool = torch.zeros(2 * len(shapes) + 2, device=dev, dtype=torch.float32)   # amax slots of the gradient tensors
        gslot_of = {}

        def gbuf(name):
            if name not in grads:
                c, h, w = shapes[name]
                i = gslot_of.setdefault(name, len(gslot_of))
                grads[name] = K.attach_amax(K.guarded_empty((n, c, h, w), dev), gslot_pool[2 * i:2 * i + 1])
                if name in ext_base:
                    K.attach_amax_ext(grads[name], gslot_pool[2 * i + 1:2 * i + 2], ext_base[name])
            return grads[name]

        ws_bytes = 0
        wg_x6 = {}
        for op in plan:
            if op["kind"] == "conv":
                hin, win = shapes[op["src"]][1], shapes[op["src"]][2]
                if op.get("s2d"):
                    ws_bytes = max(ws_bytes, K.wgrad_x6_workspace_bytes(
                        n, 4 * op["cin"], op["cout"], hin // 2, win // 2, 4,
                        tuned_tile("wgrad6s2d", n, op["cin"], op["cout"], op["k"], op["s"], hin)))
                kh, kw, ph, pw = conv_taps(op)
                if op["rect"]:      # stride-1 same-size layers with rectangular taps: runtime-tap instantiation of the split kernel
                    wg_x6[op["lids"][0]] = True
                    ws_bytes = max(ws_bytes, K.wgrad_x6_rect_workspace_bytes(
                        n, op["cin"], op["cout"], hin, win, kh, kw,
                        tuned_tile("wgrad6r%dx%d" % (kh, kw), n, op["cin"], op["cout"], 0, 1, hin)))
                    continue
                # an unpadded stride-1 3x3 layer: same-grid problem once its output gradient is laid into planes of the input's size
                op["wg_embed"] = (self.conv_precision == "split" and self.wgrad_x6 and op["src"] != "data" and len(op["lids"]) == 1
                                  and (op["k"], op["s"], op["p"]) == (3, 1, 0) and op["cin"] >= 16)
                if op["wg_embed"]:
                    wg_x6[op["lids"][0]] = True
                    ws_bytes = max(ws_bytes, K.wgrad_x6_rect_workspace_bytes(n, op["cin"], op["cout"], hin, win, 3, 3))
                    continue
                x6 = (self.conv_precision == "split" and op["src"] != "data"
                      and K.wgrad_x6_supported(op["k"], op["s"], op["p"], hin, win, K.guard_bytes(ChanSlice(acts[op["src"]], op["src_c0"], op["cin"])))
                      and ("row_gap" in op or (self.wgrad_x6 and x6_wins("wgrad", op["cin"], op["cout"], op["k"], op["s"], hin))))
                wg_x6[op["lids"][0]] = x6
                if x6:
                    ws_bytes = max(ws_bytes, K.wgrad_x6_workspace_bytes(
                        n, op["cin"], op["cout"], hin, win, op["k"],
                        tuned_tile("wgrad6", n, op["cin"], op["cout"], op["k"], op["s"], hin)))
                else:
                    ws_bytes = max(ws_bytes, K.wgrad_workspace_bytes(
                        n, op["cin"], op["cout"], shapes[op["dst"]][1], shapes[op["dst"]][2], op["k"],
                        tuned_tile("wgrad", n, op["cin"], op["cout"], op["k"], op["s"], hin)))
                if op.get("raw") or "raw_from" in op:
                    ws_bytes = max(ws_bytes, K.channel_sum_workspace_bytes(n, op["cout"]))
        ws = self._workspace(ws_bytes, dev)
        # all dgrad weight operands in two launches
        dg_ops = [op for op in plan if op["kind"] == "conv" and op["src"] != "data"]
        dg_layout = {op["lids"][0]: 1 if op["rect"] else K.dgrad_layout(op["k"], op["s"], op["p"], shapes[op["src"]][1],
                                                                        shapes[op["src"]][2]) for op in dg_ops}
        dg_x6 = {op["lids"][0]: (self.conv_precision == "split" and op["k"] in (1, 3) and op["s"] == 1 and not op["rect"]
                                 and ("raw_from" in op or "row_gap" in op
                                      or x6_wins("dgrad", op["cin"], op["cout"], op["k"], op["s"], shapes[op["src"]][1])))
                 for op in dg_ops}
        # 3x3 / stride-2 layers: four parity-class stride-1 launches on the x6 kernel (no tap that does not contribute)
        # (pad 1 on an even input: BN-Inception; pad 0: the "valid" stride-2 layers of Inception-v3)
        dg_s2 = {op["lids"][0]: (self.conv_precision == "split" and len(op["lids"]) == 1 and not op["rect"] and op["k"] == 3
                                 and op["s"] == 2 and (dg_layout[op["lids"][0]] == 2 or op["p"] == 0))
                 for op in dg_ops}
        packed_dg = {}
        for op in dg_ops:
            if dg_s2[op["lids"][0]]:
                packed_dg[op["lids"][0]] = K.pack_dgrad_s2(getattr(self, op["lids"][0]).weight.detach())
        rect_ops = [op for op in dg_ops if op["rect"]]
        packed_dg.update(zip((op["lids"][0] for op in rect_ops),
                             K.pack_rect_multi([getattr(self, op["lids"][0]).weight.detach() for op in rect_ops], dgrad=True)))
        for x6 in (False, True):
            ops = [op for op in dg_ops if dg_x6[op["lids"][0]] == x6 and not dg_s2[op["lids"][0]] and not op["rect"]]
            packed_dg.update(zip((op["lids"][0] for op in ops), K.pack_weights_multi(
                [([getattr(self, lid).weight.detach() for lid in op["lids"]], 1 if x6 else dg_layout[op["lids"][0]])
                 for op in ops], x6=x6)))

        # The backward of ReLU + frozen BN (dy <- dy * (y > 0) * scale) is fused into the store of whichever
        # launch writes a gradient slice LAST (conv dgrad or pool backward); only slices whose last writer
        # cannot do it (the global-pool backward) take the separate ssn_relu_bn_bwd pass.
        def src_key(op):
            return (op["src"], op.get("src_c0", 0))
        last_writer = {}
        for idx in range(len(plan) - 1, -1, -1):
            last_writer[src_key(plan[idx])] = idx      # reverse walk: the smallest op index writes last
        masked = {}     # tensor -> disjoint channel intervals whose ReLU/BN backward is already applied

        def mask_args(idx, op, width):
            key = src_key(op)
            if last_writer.get(key) == idx and key[0] in tscale and key[0] != "data":
                masked.setdefault(key[0], []).append((key[1], key[1] + width))
                return ChanSlice(acts[key[0]], key[1], width), tscale[key[0]][key[1]:key[1] + width]
            return None, None

        def is_masked(name, c0, c):
            covered = sum(max(0, min(hi, c0 + c) - max(lo, c0)) for lo, hi in masked.get(name, []))
            assert covered in (0, c), "partially finalised gradient slice %s[%d:%d]" % (name, c0, c0 + c)
            return covered == c

        # Two HIP streams: the data-gradient chain (dgrad / pool backward, each layer depends on the previous
        # one) stays on the caller's stream; every weight gradient only needs its layer's finished output
        # gradient and goes to a side stream, so the two kernel families fill each other's tails and
        # low-occupancy phases.  All wgrads share one stream (and therefore the split-K workspace) in order.
        use_side = self.overlap_wgrad and dfeat.is_cuda
        main = torch.cuda.current_stream(dev) if use_side else None
        side = None
        if use_side:
            side = self._side.get(dev)
            if side is None:
                side = self._side[dev] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)

        first_conv = self._conv_ids[0]
        pending_end = total
        for idx in range(len(plan) - 1, -1, -1):
            op = plan[idx]
            if op["kind"] == "gap":
                key = (op["src"], 0)
                K.gap_bwd(dfeat, ChanSlice(gbuf(op["src"]), 0, op["c"]), accumulate=key in inited)
                inited.add(key)
                # The global pool cannot fuse the ReLU / frozen-BN backward of the block it reads, so it is applied here, once,
                # for all of the block's channels -- not slice by slice when each branch gets its turn: the tensor (and its amax
                # slot) is then final before the first weight gradient on the side stream reads it.
                if op["src"] in tscale and not self._train_bn_ids():
                    c = op["c"]
                    K.relu_bn_bwd(ChanSlice(grads[op["src"]], 0, c), ChanSlice(acts[op["src"]], 0, c), tscale[op["src"]][0:c])
                    masked.setdefault(op["src"], []).append((0, c))
            elif op["kind"] == "pool":
                c = op["c"]
                key = (op["src"], 0)
                my, ms = mask_args(idx, op, c)
                K.pool_bwd(op["pool"], ChanSlice(grads[op["dst"]], op["dst_c0"], c), argmax.get(op["lid"]),
                           ChanSlice(gbuf(op["src"]), 0, c), op["k"], op["s"], op["p"], accumulate=key in inited,
                           mask_y=my, mask_scale=ms)
                inited.add(key)
            elif op["kind"] == "pool_aff":
                # y = relu(scale * avgpool(z) + shift): finish the slice's ReLU/BN backward if no later launch did, then
                # the pool's backward (for 3x3 / s1 / p1 with count_include_pad: the same stencil on the gradient)
                c = op["c"]
                g = ChanSlice(grads[op["dst"]], op["dst_c0"], c)
                if not is_masked(op["dst"], op["dst_c0"], c):
                    K.relu_bn_bwd(g, ChanSlice(acts[op["dst"]], op["dst_c0"], c),
                                  tscale[op["dst"]][op["dst_c0"]:op["dst_c0"] + c])
                    masked.setdefault(op["dst"], []).append((op["dst_c0"], op["dst_c0"] + c))
                K.pool_bwd("avg", g, None, ChanSlice(gbuf(op["src"]), op.get("src_c0", 0), c), op["k"], op["s"], op["p"],
                           accumulate=False)
                inited.add((op["src"], op.get("src_c0", 0)))
            elif op["kind"] == "bn_train":
                # batch-norm backward of the layer(s): the slice's gradient arrives untouched (NaN scales, see forward).
                # Its reductions get their own scratch: `ws` belongs to the weight-gradient chain on the side stream.
                off = 0
                bws = torch.empty(K.bn_train_workspace_floats(n, op["c"]), device=dev, dtype=torch.float32)
                for lid, c in zip(op["lids"], op["couts"]):
                    bn = getattr(self, lid + "_bn")
                    mean, invstd = bnstat[lid]
                    dgamma = torch.empty(c, device=dev, dtype=torch.float32)
                    dbeta = torch.empty(c, device=dev, dtype=torch.float32)
                    K.bn_train_bwd(ChanSlice(grads[op["dst"]], op["dst_c0"] + off, c),
                                   ChanSlice(acts[op["dst"]], op["dst_c0"] + off, c), ChanSlice(acts[op["src"]], off, c),
                                   mean, invstd, bn.weight.detach(), dgamma, dbeta, ChanSlice(gbuf(op["src"]), off, c), bws,
                                   True)
                    bn_grads[lid] = (dgamma, dbeta)
                    off += c
                inited.add((op["src"], 0))
            else:
                cout, cin, k, s, p = op["cout"], op["cin"], op["k"], op["s"], op["p"]
                lids = op["lids"]
                g = ChanSlice(grads[op["dst"]], op["dst_c0"], cout)
                raw = bool(op.get("raw"))       # bias-free, affine-free projection in front of a pool: nothing to undo
                # channel ranges of the destination tensor whose ReLU / frozen-BN backward is still due (the raw rows of a
                # projection riding along have none; a block-head launch has its rows behind the split row_gap channels up)
                c_aff, gap, split = op.get("raw_from", cout), op.get("row_gap", 0), op.get("row_split", cout)
                ranges = [(0, min(split, c_aff))] + ([(split + gap, c_aff - split)] if c_aff > split else [])
                if "row_gap" in op:      # (finer: the reduce rows are finalised per consuming convolution)
                    offs = [sum(op["couts"][:q]) for q in range(len(lids))]
                    ranges = [(o_ + (gap if o_ >= split else 0), c_) for o_, c_ in zip(offs, op["couts"]) if o_ < c_aff]
                for r0, rc in ([] if raw else ranges):
                    if not is_masked(op["dst"], op["dst_c0"] + r0, rc):
                        K.relu_bn_bwd(ChanSlice(grads[op["dst"]], op["dst_c0"] + r0, rc),
                                      ChanSlice(acts[op["dst"]], op["dst_c0"] + r0, rc),
                                      tscale[op["dst"]][op["dst_c0"] + r0:op["dst_c0"] + r0 + rc])
                wo, wn, bo, bn = lay[lids[0]]
                for extra in lids[1:]:   # fused launch: [wA | wB | (wP)] and [bA | bB | (bP)] are contiguous in the flat layout
                    wo2, wn2, bo2, bn2 = lay[extra]
                    assert wo2 == wo + wn and bo2 == bo + bn
                    wn, bn = wn + wn2, bn + bn2
                kh, kw, ph, pw = conv_taps(op)
                dw = flat[wo:wo + wn].view(cout, cin, kh, kw)
                db = flat[bo:bo + bn]
                ho = shapes[op["dst"]][1]
                hin = shapes[op["src"]][1]
                flops = 2.0 * n * ho * shapes[op["dst"]][2] * cout * cin * kh * kw
                xin = ChanSlice(acts[op["src"]], op["src_c0"], cin)
                if op["rect"]:
                    wcfg = tuned_tile("wgrad6r%dx%d" % (kh, kw), n, cin, cout, 0, 1, hin)
                    run_wgrad = lambda: K.conv_wgrad_x6_rect(g, xin, dw, db, kh, kw, ph, pw, ws, wcfg)   # noqa: E731
                elif op.get("wg_embed"):
                    def run_wgrad():
                        gp = torch.empty((n, cout, hin, shapes[op["src"]][2]), device=dev, dtype=torch.float32)
                        K.embed_planes(g, gp)
                        K.conv_wgrad_x6_rect(full(gp), xin, dw, db, 3, 3, 0, 0, ws)
                elif op.get("s2d"):
                    # the stem in its space-to-depth form: 4x4-tap weight gradient on the split kernel, gathered back into
                    # the 7x7 layout of the parameter (the bias gradient comes with it)
                    xs2 = full(acts["data_s2d"])
                    dw2 = torch.empty((cout, 4 * cin, 4, 4), device=dev, dtype=torch.float32)
                    wcfg = tuned_tile("wgrad6s2d", n, cin, cout, k, s, hin)

                    def run_wgrad():
                        K.conv_wgrad_x6(g, xs2, dw2, db, 4, 2, ws, wcfg)
                        K.s2d_weights_bwd(dw2, dw)
                elif wg_x6[lids[0]]:
                    wcfg = tuned_tile("wgrad6", n, cin, cout, k, s, hin)
                    run_wgrad = lambda: K.conv_wgrad_x6(g, xin, dw, db, k, p, ws, wcfg,   # noqa: E731
                                                        g_row_split=op.get("row_split", 0), g_row_gap=op.get("row_gap", 0))
                else:
                    wcfg = tuned_tile("wgrad", n, cin, cout, k, s, hin)
                    run_wgrad = lambda: K.conv_wgrad(g, xin, dw, db, k, s, p, ws, wcfg)   # noqa: E731
                if (raw and not op.get("bn_train")) or "raw_from" in op:
                    # the (projection's) bias sits behind the pool: its gradient is the sum of the gradient BEFORE the
                    # pool's backward (the wgrad kernel's bias column sums the pooled gradient, which differs at the border)
                    fin = op["proj_final"] if "raw_from" in op else op["final"]
                    cp = cout - op.get("raw_from", 0)
                    g_pre = ChanSlice(grads[fin[0]], fin[1], cp)
                    db_proj = db[op.get("raw_from", 0):]
                    inner_wgrad = run_wgrad

                    def run_wgrad():
                        inner_wgrad()
                        K.channel_sum(g_pre, db_proj, ws)
                wfam = "conv_wgrad_x6" if (wg_x6[lids[0]] or op.get("s2d")) else "conv_wgrad_f32"
                if use_side:
                    ready = torch.cuda.Event()
                    ready.record(main)            # the output gradient of this layer is final here
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        self._timed(wfam, lids[0], flops, run_wgrad)
                else:
                    self._timed(wfam, lids[0], flops, run_wgrad)
                if op["src"] != "data":
                    layout = dg_layout[lids[0]]
                    wt = packed_dg[lids[0]]
                    key = src_key(op)
                    acc_flag = key in inited
                    my, ms = mask_args(idx, op, cin)
                    dx = ChanSlice(gbuf(op["src"]), op["src_c0"], cin)
                    if op["rect"]:
                        self._timed("conv_dgrad_x6", lids[0], flops,
                                    lambda: K.conv_x6_dgrad_rect(g, wt, dx, kh, kw, ph, pw, acc_flag,
                                                                 tuned_tile("dgrad6r%dx%d" % (kh, kw), n, cin, cout, 0, 1, hin),
                                                                 mask_y=my, mask_scale=ms))
                    elif dg_s2[lids[0]]:
                        self._timed("conv_dgrad_x6", lids[0], flops,
                                    lambda: K.conv_x6_dgrad_s2(g, wt, dx, acc_flag,
                                                               tuned_tile("dgrad6s2", n, cin, cout, k, s, hin),
                                                               mask_y=my, mask_scale=ms, pad=p))
                    elif dg_x6[lids[0]]:
                        self._timed("conv_dgrad_x6", lids[0], flops,
                                    lambda: K.conv_x6_dgrad(g, wt, dx, k, p, acc_flag,
                                                            tuned_tile("dgrad6", n, cin, cout, k, s, hin),
                                                            mask_y=my, mask_scale=ms, k_split=op.get("row_split", 0),
                                                            k_gap=op.get("row_gap", 0)))
                    else:
                        self._timed("conv_dgrad_f32", lids[0], flops,
                                    lambda: K.conv_dgrad(g, wt, dx, k, s, p, accumulate=acc_flag,
                                                         tile_cfg=tuned_tile("dgrad", n, cin, cout, k, s, hin),
                                                         mask_y=my, mask_scale=ms, wt_layout=layout))
                    inited.add(key)
                closes_block = (lids[0].endswith("_1x1") or lids[0] == first_conv
                                or lids[0] in ("inception_3c_3x3_reduce", "inception_4e_3x3_reduce"))
                if hook and self.grad_ready_hook is not None and closes_block:
                    # the first conv of a block (forward order) closes that block's contiguous range
                    if use_side:
                        main.wait_stream(side)    # the block's wgrads must have landed before the all-reduce
                    self.grad_ready_hook.range_ready(flat, wo, pending_end)
                    pending_end = wo
        if use_side:
            main.wait_stream(side)
        if hook and self.grad_ready_hook is not None:
            if pending_end > 0:
                self.grad_ready_hook.range_ready(flat, 0, pending_end)
            self.grad_ready_hook.finish()
        out = []
        for lid in self._conv_ids:
            conv = getattr(self, lid)
            wo, wn, bo, bn = lay[lid]
            out.append(flat[wo:wo + wn].view_as(conv.weight) if conv.weight.requires_grad else None)
            out.append(flat[bo:bo + bn] if conv.bias.requires_grad else None)
        for lid in self._train_bn_ids():
            bnm = getattr(self, lid + "_bn")
            dgamma, dbeta = bn_grads[lid]
            out.append(dgamma if bnm.weight.requires_grad else None)
            out.append(dbeta if bnm.bias.requires_grad else None)
        return out, flat
