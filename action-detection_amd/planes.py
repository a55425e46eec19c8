"""Planes tensors (csrc/planes.h) on the host side: allocation, channel slices, scale / amax slots, and thin wrappers over
the C-ABI entry points that read or write them.

A planes tensor holds an fp32 activation as two f16 terms of ``x * scale`` in the channel-blocked NC8HW8 layout the
matrix cores consume directly.  ``PlaneTensor`` owns the two planes (one ``torch.float16`` tensor ``[2, N, G, H*W, 8]``)
and the association with its two slots: ``scale`` (a power of two, fixed while a step runs) and ``amax`` (the largest
magnitude the step's producers stored).  ``SlotPool`` owns the slots of an executor and the update between steps.
"""
import torch

from . import _lib
from . import kernels as K
from .kernels import _p, _stream


class SlotPool:
    """amax / scale slots of a set of planes tensors, plus the two status words the update / range-check kernels maintain:
    flag[0] = range fault of a pass (bit 0: a tensor outgrew its scale and was clamped; bit 1: a tensor fell far below it),
    flag[1] = number of scales changed since the last reset.  Several pools may share one flag tensor (all executor states of
    a backbone on a device do: the optimizer's skip word, ``kernels.sgd_step_multi(skip_flag=)``)."""

    def __init__(self, n, device, flag=None):
        self.device = device
        self.amax = torch.zeros(n, device=device, dtype=torch.float32)
        self.scale = torch.ones(n, device=device, dtype=torch.float32)
        self.flag = torch.zeros(2, device=device, dtype=torch.int32) if flag is None else flag
        self.used = 0
        self.odd_extra_bits = 0      # extra head-room of the odd slots (planes_exec.PlanesState keeps gradient tensors there)

    def take(self):
        i = self.used
        if i >= self.amax.numel():
            raise RuntimeError("slot pool exhausted")
        self.used += 1
        return i

    def update(self, exact=False, first=0, count=None):
        """scale <- f(amax) for slots [first, first + count), amax <- 0."""
        lib = _lib.get_lib()
        count = self.used - first if count is None else count
        if count <= 0:
            return
        lib.call("ssn_pl_scales_update", self.amax.data_ptr() + 4 * first, self.scale.data_ptr() + 4 * first,
                 _p(self.flag), int(count), int(bool(exact)), int(first), int(self.odd_extra_bits), _stream(lib, self.amax))

    def range_check(self):
        """flag[0] |= fault bits of the used slots (no slot is modified; no host sync: hipGraph-capturable)."""
        if self.used <= 0:
            return
        lib = _lib.get_lib()
        lib.call("ssn_pl_range_check", _p(self.amax), _p(self.scale), _p(self.flag), int(self.used), int(self.odd_extra_bits),
                 _stream(lib, self.amax))


class PlaneTensor:
    """[N, C, H, W] activation as hi / lo f16 planes (C padded to a multiple of 8)."""

    def __init__(self, n, c, h, w, device, pool=None, slot=None, scale_store=None):
        self.n, self.c, self.h, self.w = int(n), int(c), int(h), int(w)
        self.g = (self.c + 7) // 8
        self.data = torch.empty((2, self.n, self.g, self.h * self.w, 8), device=device, dtype=torch.float16)
        if pool is None:
            pool = SlotPool(1, device)
        self.pool = pool
        self.slot = pool.take() if slot is None else slot
        # where this tensor's scale is read from: the pool's live array, or a snapshot of it taken when the pass that wrote
        # the tensor started (the pool's entry moves on with the next pass; data already stored must keep its scale)
        self.scale_store = pool.scale if scale_store is None else scale_store

    @property
    def device(self):
        return self.data.device

    @property
    def scale(self):
        return self.scale_store[self.slot:self.slot + 1]

    @property
    def amax(self):
        return self.pool.amax[self.slot:self.slot + 1]

    @property
    def scale_ptr(self):
        return self.scale_store.data_ptr() + 4 * self.slot

    @property
    def amax_ptr(self):
        return self.pool.amax.data_ptr() + 4 * self.slot

    def plane_ptr(self, plane, c0=0):
        assert c0 % 8 == 0
        return self.data[plane].data_ptr() + (c0 // 8) * self.h * self.w * 16

    def zero_(self):
        self.data.zero_()
        return self


class PSlice:
    """Channels [c0, c0 + c) of a PlaneTensor (c0 a multiple of 8)."""

    def __init__(self, t, c0=0, c=None):
        self.t, self.c0 = t, int(c0)
        self.c = t.c - self.c0 if c is None else int(c)
        assert self.c0 % 8 == 0 and self.c0 + self.c <= t.g * 8

    @property
    def hi(self):
        return self.t.plane_ptr(0, self.c0)

    @property
    def lo(self):
        return self.t.plane_ptr(1, self.c0)

    @property
    def n(self):
        return self.t.n

    @property
    def hw(self):
        return self.t.h, self.t.w

    @property
    def groups(self):
        return self.t.g


def pfull(t):
    return PSlice(t, 0, t.c)


def _lib_for(t):
    lib = _lib.get_lib()
    if not lib.is_emulator and not t.data.is_cuda:
        raise RuntimeError("SSN HIP ops need HIP (cuda) tensors; there is no CPU fallback")
    return lib


def _st(lib, t):
    return _stream(lib, t.data)


def from_f32(x, dst=None, pool=None, s2d=False, exact=True, amax=None):
    """fp32 NCHW tensor (or ChanSlice-like [N, C, H, W] contiguous tensor) -> planes.  exact: measure max |x| first and
    derive the scale from it (two passes: what the caller's frames and test inputs get); otherwise the destination's
    current scale is used and its amax slot raised (the delayed protocol).  amax (with exact): a one-element tensor that already
    holds max |x| -- another kernel that read x left it there -- instead of the measuring pass."""
    from . import kernels as K
    n, c, h, w = x.shape
    if dst is None:
        dst = PlaneTensor(n, 4 * c if s2d else c, h // 2 if s2d else h, w // 2 if s2d else w, x.device, pool)
    t = dst.t if isinstance(dst, PSlice) else dst
    sl = dst if isinstance(dst, PSlice) else pfull(dst)
    lib = _lib_for(t)
    if exact:
        if amax is not None:
            t.amax.copy_(amax.reshape(t.amax.shape))
        else:
            t.amax.zero_()
            K.tensor_amax(x, t.amax)
        t.pool.update(exact=True, first=t.slot, count=1)
        if t.scale_store is not t.pool.scale:
            t.scale_store[t.slot:t.slot + 1].copy_(t.pool.scale[t.slot:t.slot + 1])
    lib.call("ssn_pl_from_f32", _p(x), c * h * w, sl.hi, sl.lo, n, c, h, w, t.g, int(bool(s2d)), t.scale_ptr,
             None if exact else t.amax_ptr, _st(lib, t))
    return dst


def to_f32(src, out=None):
    """planes (PlaneTensor / PSlice) -> fp32 NCHW tensor."""
    sl = src if isinstance(src, PSlice) else pfull(src)
    t = sl.t
    lib = _lib_for(t)
    if out is None:
        out = torch.empty((t.n, sl.c, t.h, t.w), device=t.device, dtype=torch.float32)
    lib.call("ssn_pl_to_f32", sl.hi, sl.lo, t.g, _p(out), sl.c * t.h * t.w, t.n, sl.c, t.h * t.w, t.scale_ptr, _st(lib, t))
    return out


def im2col(x, kh, kw, stride, pad_h, pad_w, ho, wo):
    """PSlice x [N, C, H, W] with few channels -> PlaneTensor [N, C kh kw, ho, wo] of its window elements (channel c kh kw + r kw + s),
    sharing x's scale: the operand that turns the weight gradient of a first convolution into a 1x1 problem."""
    t = x.t
    lib = _lib_for(t)
    h, w = x.hw
    y = PlaneTensor(x.n, x.c * kh * kw, ho, wo, t.device, t.pool, t.slot, t.scale_store)
    lib.call("ssn_pl_im2col", x.hi, x.lo, x.groups, y.plane_ptr(0), y.plane_ptr(1), y.g, x.n, x.c, h, w, ho, wo, kh, kw, stride, pad_h, pad_w,
             _st(lib, t))
    return y


def conv_fwd(x, w_packed, scale, shift, y, kh, kw, stride, pad_h, pad_w, relu=True, tile_cfg=-1, raw_from=0, row_split=0,
             row_gap=0):
    """y <- relu?(scale * conv(x) + shift) on planes slices.  w_packed: kernels.pack_weights_multi(x6=True) /
    pack_weights_rect / pack_rect_multi forward operand (the packed image is shared with the fp32-layout split kernels)."""
    lib = _lib_for(x.t)
    h, w = x.hw
    ho, wo = y.hw
    lib.call("ssn_conv_pl_fwd", x.hi, x.lo, _p(w_packed), _p(scale), _p(shift), y.hi, y.lo, x.n, x.c, h, w, x.groups, y.c, ho,
             wo, y.groups, kh, kw, stride, pad_h, pad_w, int(bool(relu)), tile_cfg, x.t.scale_ptr, y.t.scale_ptr, y.t.amax_ptr,
             int(raw_from), int(row_split), int(row_gap), _st(lib, x.t))


def conv_dgrad(dy, wt_packed, dx, kh, kw, pad_h, pad_w, accumulate=False, tile_cfg=-1, mask=None, mask_scale=None, k_split=0,
               k_gap=0, taps_reversed=False):
    """dx (+)= conv_transpose(dy) on planes slices (stride 1).  mask: PSlice of the forward activation at dx's channels +
    mask_scale [dx.c]: fuse the ReLU / frozen-BN backward into the store.  taps_reversed: wt_packed is the transposed,
    tap-reversed operand (kernels.pack_dgrad_rect / pack_rect_multi(dgrad=True)); otherwise pack mode 1."""
    lib = _lib_for(dy.t)
    ho, wo = dy.hw
    h, w = dx.hw
    lib.call("ssn_conv_pl_dgrad", dy.hi, dy.lo, _p(wt_packed), dx.hi, dx.lo, dy.n, dy.c, ho, wo, dy.groups, dx.c, h, w,
             dx.groups, kh, kw, pad_h, pad_w, int(bool(accumulate)), mask.hi if mask is not None else None,
             mask.groups if mask is not None else 0, _p(mask_scale), tile_cfg, dy.t.scale_ptr, dx.t.scale_ptr, dx.t.amax_ptr,
             int(k_split), int(k_gap), int(bool(taps_reversed)), _st(lib, dy.t))


def wgrad_workspace_bytes(n, cin, cout, ho, wo, kh, kw, tile_cfg=-1):
    return int(_lib.get_lib().cdll.ssn_conv_wgrad_pl_workspace_bytes(n, cin, cout, ho, wo, kh, kw, tile_cfg))


def conv_wgrad(g, x, dw, db, kh, kw, stride, pad_h, pad_w, workspace, tile_cfg=-1, cin=None, g_row_split=0, g_row_gap=0, defer=None):
    """dw [Cout, Cin, kh, kw] (fp32), db [Cout] or None <- weight / bias gradient from the planes slices g (output gradient, final)
    and x (the layer's input).  cin: real input channels when x's slice is zero-padded (the 12-channel stem).
    defer: a list -- the split-K slabs then STAY in `workspace` (which must not be reused before the flush) and an entry for
    ``wgrad_reduce_multi`` is appended instead of launching this layer's reduction."""
    import ctypes
    lib = _lib_for(g.t)
    h, w = x.hw
    ho, wo = g.hw
    cin = x.c if cin is None else cin
    info = (ctypes.c_int * 2)() if defer is not None else None
    lib.call("ssn_conv_wgrad_pl", g.hi, g.lo, x.hi, x.lo, _p(dw), _p(db), x.n, cin, h, w, x.groups, g.c, ho, wo, g.groups, kh, kw,
             stride, pad_h, pad_w, _p(workspace), workspace.numel() * workspace.element_size(), tile_cfg, g.t.scale_ptr,
             x.t.scale_ptr, int(g_row_split), int(g_row_gap), ctypes.addressof(info) if info is not None else None, _st(lib, g.t))
    if defer is not None:
        defer.append((workspace, dw, db, g.c, cin * kh * kw, int(info[0]), int(info[1])))


def wgrad_reduce_multi(entries):
    """Reduce the deferred split-K slabs of many layers (entries of ``conv_wgrad(defer=)``) in one launch per 48 layers."""
    import ctypes
    if not entries:
        return
    lib = _lib.get_lib()
    n = len(entries)
    parts = (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in entries])
    dws = (ctypes.c_void_p * n)(*[e[1].data_ptr() for e in entries])
    dbs = (ctypes.c_void_p * n)(*[(e[2].data_ptr() if e[2] is not None else None) for e in entries])
    arr = [(ctypes.c_int * n)(*[int(e[k]) for e in entries]) for k in (3, 4, 5, 6)]
    lib.call("ssn_wgrad_reduce_multi", n, ctypes.addressof(parts), ctypes.addressof(dws), ctypes.addressof(dbs),
             ctypes.addressof(arr[0]), ctypes.addressof(arr[1]), ctypes.addressof(arr[2]), ctypes.addressof(arr[3]),
             _stream(lib, entries[0][0]))


class WgradJob:
    """One problem of a grouped weight-gradient launch: the arguments of ``conv_wgrad`` (g: final output gradient slice, x: the layer's
    input slice, dw / db: fp32 destinations) + an optional tile hint (-1: the library chooses)."""

    __slots__ = ("g", "x", "dw", "db", "kh", "kw", "stride", "pad_h", "pad_w", "cin", "g_row_split", "g_row_gap", "hint")

    def __init__(self, g, x, dw, db, kh, kw, stride, pad_h, pad_w, cin=None, g_row_split=0, g_row_gap=0, hint=-1):
        self.g, self.x, self.dw, self.db = g, x, dw, db
        self.kh, self.kw, self.stride, self.pad_h, self.pad_w = int(kh), int(kw), int(stride), int(pad_h), int(pad_w)
        self.cin = x.c if cin is None else int(cin)
        self.g_row_split, self.g_row_gap, self.hint = int(g_row_split), int(g_row_gap), int(hint)

    def shape16(self):
        h, w = self.x.hw
        ho, wo = self.g.hw
        return [self.x.n, self.cin, h, w, self.g.c, ho, wo, self.kh, self.kw, self.stride, self.pad_h, self.pad_w,
                self.g_row_split, self.g_row_gap, self.hint, 0]


def _wgrad_group_arrays(jobs):
    import ctypes
    n = len(jobs)
    shape = (ctypes.c_int * (16 * n))(*[v for j in jobs for v in j.shape16()])
    groups = (ctypes.c_long * (2 * n))(*[v for j in jobs for v in (j.x.groups, j.g.groups)])
    return shape, groups


def wgrad_group_plan(jobs):
    """(workspace bytes, table bytes, [(family, variant, splits, units per split)]) of a grouped launch -- host-side planning only."""
    import ctypes
    lib = _lib.get_lib()
    n = len(jobs)
    if n == 0:
        return 0, 0, []
    shape, groups = _wgrad_group_arrays(jobs)
    plan = (ctypes.c_int * (4 * n))()
    ws = int(lib.cdll.ssn_conv_wgrad_pl_group_workspace_bytes(n, ctypes.addressof(shape), ctypes.addressof(groups), ctypes.addressof(plan)))
    if ws < 0:
        raise RuntimeError("ssn_conv_wgrad_pl_group_workspace_bytes failed: %s" % lib.cdll.ssn_last_error().decode())
    return ws, int(lib.cdll.ssn_conv_wgrad_pl_group_table_bytes(n)), [tuple(plan[4 * i:4 * i + 4]) for i in range(n)]


def conv_wgrad_group(jobs, workspace=None, table=None):
    """Every weight + bias gradient of ``jobs`` (WgradJob) in at most five launches + one reduction (ssn_conv_wgrad_pl_group).
    workspace (fp32 tensor) / table (uint8 tensor): buffers of at least ``wgrad_group_plan(jobs)`` bytes, allocated when None."""
    import ctypes
    if not jobs:
        return
    first = jobs[0].g
    lib = _lib_for(first.t)
    n = len(jobs)
    dev = first.t.device
    assert all(j.g.t.device == dev and j.x.t.device == dev for j in jobs)
    shape, groups = _wgrad_group_arrays(jobs)
    if workspace is None or table is None:
        ws_bytes, tb_bytes, _ = wgrad_group_plan(jobs)
        if workspace is None:
            workspace = torch.empty(ws_bytes // 4 + 4, device=dev, dtype=torch.float32)
        if table is None:
            table = torch.empty(tb_bytes, device=dev, dtype=torch.uint8)
    vp = ctypes.c_void_p * n
    g_hi, g_lo = vp(*[j.g.hi for j in jobs]), vp(*[j.g.lo for j in jobs])
    x_hi, x_lo = vp(*[j.x.hi for j in jobs]), vp(*[j.x.lo for j in jobs])
    dws = vp(*[_p(j.dw) for j in jobs])
    dbs = vp(*[_p(j.db) for j in jobs])
    gs, xs = vp(*[j.g.t.scale_ptr for j in jobs]), vp(*[j.x.t.scale_ptr for j in jobs])
    lib.call("ssn_conv_wgrad_pl_group", n, ctypes.addressof(g_hi), ctypes.addressof(g_lo), ctypes.addressof(x_hi),
             ctypes.addressof(x_lo), ctypes.addressof(dws), ctypes.addressof(dbs), ctypes.addressof(shape), ctypes.addressof(groups),
             ctypes.addressof(gs), ctypes.addressof(xs), _p(workspace), workspace.numel() * workspace.element_size(), _p(table),
             table.numel() * table.element_size(), _st(lib, first.t))


def conv_dgrad_s2(dy, wt_packed, dx, pad, accumulate=False, tile_cfg=-1, mask=None, mask_scale=None):
    """dgrad of a 3x3 / stride-2 conv on planes slices (pad 1 on an even input, or pad 0).  wt_packed: kernels.pack_dgrad_s2(w)."""
    lib = _lib_for(dy.t)
    ho, wo = dy.hw
    h, w = dx.hw
    lib.call("ssn_conv_pl_dgrad_s2", dy.hi, dy.lo, _p(wt_packed), dx.hi, dx.lo, dy.n, dy.c, ho, wo, dy.groups, dx.c, h, w, dx.groups,
             int(pad), int(bool(accumulate)), mask.hi if mask is not None else None, mask.groups if mask is not None else 0,
             _p(mask_scale), tile_cfg, dy.t.scale_ptr, dx.t.scale_ptr, dx.t.amax_ptr, _st(lib, dy.t))


def maxpool_fwd(x, y, argmax, k, s, pad):
    """x, y: PSlice with equal channel counts; argmax: uint8 tensor [N, C/8, Ho*Wo, 8] or None."""
    lib = _lib_for(x.t)
    h, w = x.hw
    ho, wo = y.hw
    lib.call("ssn_pl_maxpool_fwd", x.hi, x.lo, x.groups, y.hi, y.lo, y.groups, _p(argmax), x.n, x.c, h, w, ho, wo, k, s, pad,
             x.t.scale_ptr, y.t.scale_ptr, y.t.amax_ptr, _st(lib, x.t))


def maxpool_bwd(dy, argmax, dx, k, s, pad, accumulate=False, mask=None, mask_scale=None, mask_pooled=False):
    """dx: PSlice, or an fp32 NCHW tensor [N, C, H, W] with an amax slot attached (kernels.attach_amax): the gradient is then
    written in the fp32 layout (for a consumer on the fp32-layout kernels).  mask: the forward activation the pool READ (its ReLU /
    frozen-BN backward is fused) -- or, with mask_pooled, the activation the pool WROTE (3x3 / stride 2, no accumulation: the
    decision of a window's maximum is the sign of the pooled value; a quarter of the bytes)."""
    lib = _lib_for(dy.t)
    ho, wo = dy.hw
    if isinstance(dx, PSlice):
        h, w = dx.hw
        lib.call("ssn_pl_maxpool_bwd", dy.hi, dy.lo, dy.groups, _p(argmax), dx.hi, dx.lo, dx.groups, dx.n, dx.c, h, w, ho, wo, k,
                 s, pad, int(bool(accumulate)), mask.hi if mask is not None else None, mask.groups if mask is not None else 0,
                 _p(mask_scale), int(bool(mask_pooled)), dy.t.scale_ptr, dx.t.scale_ptr, dx.t.amax_ptr, None, 0, _st(lib, dy.t))
        return
    n, c, h, w = dx.shape
    assert not accumulate and dx.is_contiguous() and dx.dtype == torch.float32
    one = _ones(dx.device)
    lib.call("ssn_pl_maxpool_bwd", dy.hi, dy.lo, dy.groups, _p(argmax), None, None, 0, n, c, h, w, ho, wo, k, s, pad, 0,
             mask.hi if mask is not None else None, mask.groups if mask is not None else 0, _p(mask_scale),
             int(bool(mask_pooled)), dy.t.scale_ptr, _p(one), _p(getattr(dx, "_ssn_amax", None)), _p(dx), c * h * w, _st(lib, dy.t))


_ONES = {}


def _ones(device):
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones(1, device=device, dtype=torch.float32)
    return t


def avgpool_affine(x, y, scale, shift, relu, k, pad):
    """y = relu?(scale * avgpool_kxk(x) + shift), stride 1, count_include_pad (scale / shift None: the plain average -- also the
    backward of that pool when x is the output gradient)."""
    lib = _lib_for(x.t)
    h, w = x.hw
    lib.call("ssn_pl_avgpool_affine", x.hi, x.lo, x.groups, y.hi, y.lo, y.groups, _p(scale), _p(shift), int(bool(relu)), x.n, x.c,
             h, w, k, pad, x.t.scale_ptr, y.t.scale_ptr, y.t.amax_ptr, _st(lib, x.t))


def relu_bn_bwd(g, y, scale):
    """in place on the planes slice g: g <- g * (y > 0) * scale[c] (NaN scale: pass through)."""
    lib = _lib_for(g.t)
    h, w = g.hw
    lib.call("ssn_pl_relu_bn_bwd", g.hi, g.lo, g.groups, y.hi, y.groups, _p(scale), g.n, g.c, h * w, g.t.scale_ptr, g.t.amax_ptr,
             _st(lib, g.t))


def gap_fwd(x, out):
    """planes slice [N, C, H, W] -> fp32 [N, C] mean over the pixels."""
    lib = _lib_for(x.t)
    h, w = x.hw
    lib.call("ssn_pl_gap_fwd", x.hi, x.lo, x.groups, _p(out), x.n, x.c, h * w, x.t.scale_ptr, _st(lib, x.t))


def gap_bwd(dy, dx, mask=None, mask_scale=None):
    """fp32 [N, C] -> planes slice dx = dy / HW per pixel (times the fused ReLU / frozen-BN backward when mask is given)."""
    lib = _lib_for(dx.t)
    h, w = dx.hw
    lib.call("ssn_pl_gap_bwd", _p(dy), dx.hi, dx.lo, dx.groups, dx.n, dx.c, h * w, mask.hi if mask is not None else None,
             mask.groups if mask is not None else 0, _p(mask_scale), dx.t.scale_ptr, dx.t.amax_ptr, _st(lib, dx.t))


def channel_sum_multi(entries, workspace):
    """entries: [(g: PSlice, out: fp32 [C])] of one pass -> every out = per-channel sums of its slice, in one pair of launches."""
    import ctypes
    if not entries:
        return
    first = entries[0][0]
    lib = _lib_for(first.t)
    n = len(entries)
    vp = ctypes.c_void_p * n
    hi, lo = vp(*[g.hi for g, _ in entries]), vp(*[g.lo for g, _ in entries])
    outs, scales = vp(*[_p(o) for _, o in entries]), vp(*[g.t.scale_ptr for g, _ in entries])
    groups = (ctypes.c_long * n)(*[g.groups for g, _ in entries])
    cs = (ctypes.c_int * n)(*[g.c for g, _ in entries])
    hws = (ctypes.c_int * n)(*[g.hw[0] * g.hw[1] for g, _ in entries])
    assert all(g.n == first.n for g, _ in entries)
    lib.call("ssn_pl_channel_sum_multi", n, ctypes.addressof(hi), ctypes.addressof(lo), ctypes.addressof(groups),
             ctypes.addressof(outs), first.n, ctypes.addressof(cs), ctypes.addressof(hws), ctypes.addressof(scales), _p(workspace),
             workspace.numel() * workspace.element_size(), _st(lib, first.t))


def bn_train_workspace_bytes(c):
    return int(_lib.get_lib().cdll.ssn_pl_bn_train_workspace_bytes(int(c)))


def bn_train_stats(z, conv_bias, mean, invstd, running_mean, running_var, eps, momentum, workspace):
    """Batch statistics of the planes slice z (the convolution WITHOUT its bias): mean / invstd (real units) and the running
    statistics update of torch.nn.BatchNorm2d (running_mean / running_var None: not tracked)."""
    if running_mean is not None:
        K.PARAM_EPOCH += 1      # (written through raw pointers: invisible to torch's version counters; see planes_exec's inference cache)
    lib = _lib_for(z.t)
    h, w = z.hw
    lib.call("ssn_pl_bn_train_stats", z.hi, z.lo, z.groups, z.t.scale_ptr, _p(conv_bias), _p(mean), _p(invstd), _p(running_mean),
             _p(running_var), z.n, z.c, h * w, float(eps), float(momentum), _p(workspace),
             workspace.numel() * workspace.element_size(), _st(lib, z.t))


def bn_train_apply(z, y, mean, invstd, gamma, beta, relu=True):
    """y = relu?(gamma * (z - mean) * invstd + beta), planes slice -> planes slice."""
    lib = _lib_for(z.t)
    h, w = z.hw
    lib.call("ssn_pl_bn_train_apply", z.hi, z.lo, z.groups, z.t.scale_ptr, y.hi, y.lo, y.groups, y.t.scale_ptr, y.t.amax_ptr,
             _p(mean), _p(invstd), _p(gamma), _p(beta), int(bool(relu)), z.n, z.c, h * w, _st(lib, z.t))


def bn_train_bwd(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, workspace, relu=True):
    """Backward of bn_train_apply (+ the statistics): dgamma, dbeta, and dz -- a PSlice, or an fp32 NCHW tensor with an amax slot
    attached (kernels.attach_amax: the stem, whose weight gradient runs on the fp32-layout kernel).  y: the layer's output slice
    (only the sign of its high plane is read)."""
    lib = _lib_for(dy.t)
    h, w = z.hw
    ws_bytes = workspace.numel() * workspace.element_size()
    if isinstance(dz, PSlice):
        lib.call("ssn_pl_bn_train_bwd", dy.hi, dy.lo, dy.groups, dy.t.scale_ptr, y.hi if relu else None, y.groups, z.hi, z.lo,
                 z.groups, z.t.scale_ptr, _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), dz.hi, dz.lo, dz.groups,
                 dz.t.scale_ptr, dz.t.amax_ptr, None, 0, z.n, z.c, h * w, _p(workspace), ws_bytes, _st(lib, dy.t))
        return
    n, c, hh, ww = dz.shape
    assert (n, c, hh * ww) == (z.n, z.c, h * w) and dz.is_contiguous() and dz.dtype == torch.float32
    lib.call("ssn_pl_bn_train_bwd", dy.hi, dy.lo, dy.groups, dy.t.scale_ptr, y.hi if relu else None, y.groups, z.hi, z.lo,
             z.groups, z.t.scale_ptr, _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), None, None, 0, None,
             _p(getattr(dz, "_ssn_amax", None)), _p(dz), c * hh * ww, z.n, z.c, h * w, _p(workspace), ws_bytes, _st(lib, dy.t))


def channel_sum_workspace_bytes(c):
    return int(_lib.get_lib().cdll.ssn_pl_channel_sum_workspace_bytes(int(c)))


def channel_sum(g, out, workspace):
    lib = _lib_for(g.t)
    h, w = g.hw
    lib.call("ssn_pl_channel_sum", g.hi, g.lo, g.groups, _p(out), g.n, g.c, h * w, g.t.scale_ptr, _p(workspace),
             workspace.numel() * workspace.element_size(), _st(lib, g.t))
