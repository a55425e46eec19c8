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
        return d.get("tiles", {}), int(d.get("n_images", 0))
    except (OSError, ValueError):
        return {}, 0


_TUNED, _TUNED_N = _load_tuned()


def tuned_tile(kind, n, cin, cout, k, s, hin):
    """Tile config for a conv launch: the autotuned entry when the batch is comparable, else -1 (heuristic)."""
    if not _TUNED or n * 2 < _TUNED_N:
        return -1
    return _TUNED.get("%s|%d|%d|%d|%d|%d" % (kind, cin, cout, k, s, hin), -1)


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, *params):
        need_grad = any(ctx.needs_input_grad)
        feat, saved = net._run_forward(x, keep=need_grad)
        ctx.net = net
        ctx.saved = saved
        ctx.n_params = len(params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        net, saved = ctx.net, ctx.saved
        ctx.saved = None
        if saved is None:
            raise RuntimeError("backbone forward ran without grad bookkeeping")
        grads = net._run_backward(dfeat.contiguous(), saved)
        return (None, None) + tuple(grads)


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
        self.grad_ready_hook = None   # object with range_ready(flat, start, end) / finish() (parallel.GradReducer)
        self._ws = None
        self._side = {}               # device -> side HIP stream for the weight-gradient chain
        self.overlap_wgrad = True     # run wgrad launches on a second stream, concurrently with the dgrad chain
        self.profiler = None          # list; when set, every conv launch is bracketed by HIP events

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
        return ps

    def flat_grad_layout(self):
        """[(layer id, weight offset, weight numel, bias offset, bias numel)], total -- forward order."""
        off, lay = 0, []
        for lid in self._conv_ids:
            conv = getattr(self, lid)
            wn, bn = conv.weight.numel(), conv.bias.numel()
            lay.append((lid, off, wn, off + wn, bn))
            off += wn + bn
        return lay, off

    def features(self, x):
        if x.dim() != 4:
            raise ValueError("expected NCHW input")
        for lid in self._conv_ids:
            if getattr(self, lid + "_bn").training:
                raise NotImplementedError(
                    "bn_mode 'partial'/'full' (training-mode BatchNorm) is not built yet; "
                    "SSN's default bn_mode='frozen' keeps every BatchNorm2d in eval mode")
        return _BackboneFn.apply(x.contiguous(), self, *self._param_list())

    def forward(self, x):
        return self.fc(self.features(x))

    # ------------------------------------------------------------------ forward executor
    def _manifest(self, x):
        cin = getattr(self, self._conv_ids[0]).in_channels  # may differ after flow surgery
        if x.shape[1] != cin:
            raise ValueError("input has %d channels, first conv expects %d" % (x.shape[1], cin))
        if x.shape[2] != x.shape[3]:
            raise ValueError("square inputs only")
        return build_manifest(cin, x.shape[2])

    def _run_forward(self, x, keep):
        ops, shapes = self._manifest(x)
        n, dev = x.shape[0], x.device
        acts = {"data": x}
        argmax = {}
        folds = {}
        last_use = {}
        for i, op in enumerate(ops):
            src = op[2] if op[0] != "pool" else op[3]
            last_use[src] = i

        def get(name):
            if name not in acts:
                c, h, w = shapes[name]
                acts[name] = torch.empty((n, c, h, w), device=dev, dtype=torch.float32)
            return acts[name]

        tscale = {}   # per tensor: folded-BN scale of every channel (-1: channel is not a conv+ReLU output)

        def scale_slice(name, c0, c):
            if name not in tscale:
                tscale[name] = torch.full((shapes[name][0],), -1.0, device=dev, dtype=torch.float32)
            return tscale[name][c0:c0 + c]

        feat = None
        for i, op in enumerate(ops):
            if op[0] == "conv":
                _, lid, src, dst, c0, cin, cout, k, s, p = op
                conv, bn = getattr(self, lid), getattr(self, lid + "_bn")
                scale = scale_slice(dst, c0, cout)
                shift = torch.empty(cout, device=dev, dtype=torch.float32)
                K.bn_fold(conv.bias.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                          bn.running_var, bn.eps, scale, shift)
                ho = shapes[dst][1]
                flops = 2.0 * n * ho * ho * cout * cin * k * k
                self._timed("conv_fwd", lid, flops,
                            lambda: K.conv_fwd(full(acts[src]), K.pack_weights(conv.weight.detach(), False), scale, shift,
                                               ChanSlice(get(dst), c0, cout), k, s, p, True,
                                               tuned_tile("fwd", n, cin, cout, k, s, shapes[src][1])))
                folds[lid] = scale
            elif op[0] == "pool":
                _, lid, kind, src, dst, c0, k, s, p, _ceil = op
                c = shapes[src][0]
                out = ChanSlice(get(dst), c0, c)
                am = None
                if kind == "max" and keep:
                    _, ho, wo = shapes[dst]
                    am = torch.empty((n, c, ho, wo), device=dev, dtype=torch.uint8)
                    argmax[lid] = am
                K.pool_fwd(kind, full(acts[src]), out, am, k, s, p)
            else:
                _, lid, src, dst = op
                feat = torch.empty((n, shapes[src][0]), device=dev, dtype=torch.float32)
                K.gap_fwd(full(acts[src]), feat)
            if not keep:
                # inference: drop activations as soon as their last consumer has been launched
                for name in [nm for nm, last in last_use.items() if last == i and nm != "data"]:
                    acts.pop(name, None)
        saved = (ops, shapes, acts, argmax, folds, tscale) if keep else None
        return feat, saved

    # ------------------------------------------------------------------ backward executor
    def _workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != dev:
            self._ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        return self._ws

    def _run_backward(self, dfeat, saved):
        ops, shapes, acts, argmax, folds, tscale = saved
        n, dev = dfeat.shape[0], dfeat.device
        layout, total = self.flat_grad_layout()
        lay = {lid: (wo, wn, bo, bn) for lid, wo, wn, bo, bn in layout}
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        grads = {}
        inited = set()

        def gbuf(name):
            if name not in grads:
                c, h, w = shapes[name]
                grads[name] = torch.empty((n, c, h, w), device=dev, dtype=torch.float32)
            return grads[name]

        ws_bytes = 0
        for op in ops:
            if op[0] == "conv":
                _, lid, src, dst, c0, cin, cout, k, s, p = op
                ws_bytes = max(ws_bytes, K.wgrad_workspace_bytes(
                    n, cin, cout, shapes[dst][1], shapes[dst][2], k,
                    tuned_tile("wgrad", n, cin, cout, k, s, shapes[src][1])))
        ws = self._workspace(ws_bytes, dev)

        # The backward of ReLU + frozen BN (dy <- dy * (y > 0) * scale) is fused into the store of whichever
        # launch writes a gradient tensor LAST (conv dgrad or max-pool backward); only tensors whose last
        # writer cannot do it (the global-pool backward) take the separate ssn_relu_bn_bwd pass.
        last_writer = {}
        for idx in range(len(ops) - 1, -1, -1):
            op = ops[idx]
            src = op[3] if op[0] == "pool" else op[2]
            last_writer[src] = idx          # reverse walk: the smallest op index writes last
        masked = set()

        def mask_args(idx, src, fusable):
            if fusable and last_writer.get(src) == idx and src in tscale and src != "data":
                masked.add(src)
                return full(acts[src]), tscale[src]
            return None, None

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

        pending_end = total
        for idx in range(len(ops) - 1, -1, -1):
            op = ops[idx]
            if op[0] == "gap":
                _, lid, src, dst = op
                K.gap_bwd(dfeat, full(gbuf(src)), accumulate=src in inited)
                inited.add(src)
            elif op[0] == "pool":
                _, lid, kind, src, dst, c0, k, s, p, _ceil = op
                c = shapes[src][0]
                my, ms = mask_args(idx, src, True)
                K.pool_bwd(kind, ChanSlice(grads[dst], c0, c), argmax.get(lid), full(gbuf(src)), k, s, p,
                           accumulate=src in inited, mask_y=my, mask_scale=ms)
                inited.add(src)
            else:
                _, lid, src, dst, c0, cin, cout, k, s, p = op
                conv = getattr(self, lid)
                g = ChanSlice(grads[dst], c0, cout)
                if dst not in masked:
                    K.relu_bn_bwd(g, ChanSlice(acts[dst], c0, cout), folds[lid])
                wo, wn, bo, bn = lay[lid]
                dw = flat[wo:wo + wn].view_as(conv.weight)
                db = flat[bo:bo + bn]
                ho = shapes[dst][1]
                flops = 2.0 * n * ho * ho * cout * cin * k * k
                hin = shapes[src][1]
                wcfg = tuned_tile("wgrad", n, cin, cout, k, s, hin)
                if use_side:
                    ready = torch.cuda.Event()
                    ready.record(main)            # the output gradient of this layer is final here
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        self._timed("conv_wgrad", lid, flops,
                                    lambda: K.conv_wgrad(g, full(acts[src]), dw, db, k, s, p, ws, wcfg))
                else:
                    self._timed("conv_wgrad", lid, flops,
                                lambda: K.conv_wgrad(g, full(acts[src]), dw, db, k, s, p, ws, wcfg))
                if src != "data":
                    wt = K.pack_weights(conv.weight.detach(), True)
                    acc_flag = src in inited
                    my, ms = mask_args(idx, src, True)
                    self._timed("conv_dgrad", lid, flops,
                                lambda: K.conv_dgrad(g, wt, full(gbuf(src)), k, s, p, accumulate=acc_flag,
                                                     tile_cfg=tuned_tile("dgrad", n, cin, cout, k, s, hin),
                                                     mask_y=my, mask_scale=ms))
                    inited.add(src)
                if self.grad_ready_hook is not None and (lid.endswith("_1x1") or lid == self._conv_ids[0]
                                                         or lid == "inception_3c_3x3_reduce"
                                                         or lid == "inception_4e_3x3_reduce"):
                    # the first conv of a block (forward order) closes that block's contiguous range
                    if use_side:
                        main.wait_stream(side)    # the block's wgrads must have landed before the all-reduce
                    self.grad_ready_hook.range_ready(flat, wo, pending_end)
                    pending_end = wo
        if use_side:
            main.wait_stream(side)
        if self.grad_ready_hook is not None:
            if pending_end > 0:
                self.grad_ready_hook.range_ready(flat, 0, pending_end)
            self.grad_ready_hook.finish()
        out = []
        for lid in self._conv_ids:
            conv = getattr(self, lid)
            wo, wn, bo, bn = lay[lid]
            out.append(flat[wo:wo + wn].view_as(conv.weight) if conv.weight.requires_grad else None)
            out.append(flat[bo:bo + bn] if conv.bias.requires_grad else None)
        return out
