"""Thin tensor-level wrappers over the C ABI (include/ssn_hip.h).

Every function launches on the caller's current HIP stream and writes into caller-provided
tensors (PyTorch owns all device memory).  ``ChanSlice`` addresses a channel range of an NCHW
tensor, which is how the concat-free Inception block outputs are written and read.
"""
import ctypes
from collections import namedtuple

import torch

from . import _lib
from ._lib import SsnStppTable, STPP_MAX_PARTS


class ChanSlice(namedtuple("ChanSlice", "t c0 c")):
    """Channels [c0, c0+c) of a contiguous NCHW tensor ``t``."""

    @property
    def ptr(self):
        _, _, h, w = self.t.shape
        return self.t.data_ptr() + 4 * self.c0 * h * w

    @property
    def img_stride(self):
        _, ct, h, w = self.t.shape
        return ct * h * w

    @property
    def n(self):
        return self.t.shape[0]

    @property
    def hw(self):
        return self.t.shape[2], self.t.shape[3]


def full(t):
    return ChanSlice(t, 0, t.shape[1])


def _check(*tensors):
    lib = _lib.get_lib()
    for t in tensors:
        if t is None:
            continue
        tt = t.t if isinstance(t, ChanSlice) else t
        if not tt.is_contiguous():
            raise ValueError("SSN HIP ops need contiguous tensors")
        if not lib.is_emulator and not tt.is_cuda:
            raise RuntimeError("SSN HIP ops need HIP (cuda) tensors; there is no CPU fallback")
    return lib


def _stream(lib, t):
    if lib.is_emulator:
        return None
    tt = t.t if isinstance(t, ChanSlice) else t
    return torch.cuda.current_stream(tt.device).cuda_stream


def _p(t):
    if t is None:
        return None
    return t.ptr if isinstance(t, ChanSlice) else t.data_ptr()


# ------------------------------------------------------------------------------------ amax slots
# The split-operand ("x6") kernels scale every operand tensor by a power of two derived from an upper bound of its
# largest magnitude (include/ssn_hip.h: "amax slots").  The association tensor -> slot lives on the torch tensor
# object (attribute `_ssn_amax`, a 1-element fp32 view): producers hand the slot of their OUTPUT tensor to the kernel,
# which raises it; consumers hand the slot of their INPUT tensor.  Executors attach zeroed slots to the tensors they
# allocate (attach_amax); a tensor without a slot (test inputs, the caller's frames) is measured on the spot.
def attach_amax(t, slot=None):
    """Give tensor `t` an amax slot (a zeroed 1-element tensor, or the given view of a zeroed pool)."""
    t._ssn_amax = torch.zeros(1, device=t.device, dtype=torch.float32) if slot is None else slot
    return t


def _tensor_of(x):
    return x.t if isinstance(x, ChanSlice) else x


def attach_amax_ext(t, slot, first_channel):
    """Second amax slot for the channels >= first_channel of `t` (a block-output tensor that also holds the block's reduce
    rows): the two regions are written and read by concurrent launches, so each has its own slot -- a slot is then always
    complete before anything reads it (results stay bit-deterministic)."""
    t._ssn_amax_ext = (slot, int(first_channel))
    return t


def _slot_of(x):
    """The amax slot that covers ChanSlice / tensor x (None: not tracked)."""
    t = _tensor_of(x)
    ext = getattr(t, "_ssn_amax_ext", None)
    if ext is not None and isinstance(x, ChanSlice) and x.c0 >= ext[1]:
        return ext[0]
    return getattr(t, "_ssn_amax", None)


def _slot_ext(x):
    ext = getattr(_tensor_of(x), "_ssn_amax_ext", None)
    return None if ext is None else ext[0]


def _amax_out(x):
    """Slot pointer of the tensor (region) a kernel writes (None: not tracked)."""
    return _p(_slot_of(x))


def tensor_amax(t, slot=None):
    """Measure max |t| into `slot` (a fresh zeroed one by default) and return the slot."""
    lib = _check(t)
    if slot is None:
        slot = torch.zeros(1, device=t.device, dtype=torch.float32)
    lib.call("ssn_tensor_amax", _p(t), t.numel(), _p(slot), _stream(lib, t))
    return slot


def _amax_in(x):
    """Slot (tensor, kept alive by the caller) of the tensor a split kernel reads; untracked tensors are measured now."""
    slot = _slot_of(x)
    return slot if slot is not None else tensor_amax(_tensor_of(x))


# Measurement hook (bench.py): when a list is installed here, the wrappers of the HBM-bound kernels of the path (STPP,
# heads, row selection, losses) bracket their launch with HIP events on the launch stream and append
# (kernel name, algorithmic bytes = operands read + results written, start event, end event).
HBM_PROFILER = None
HBM_PROFILER_REPEATS = 16


def _hbm_timed(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        prof = HBM_PROFILER
        if prof is None:
            return fn(*args, **kw)
        flat = [b for a in args for b in (a if isinstance(a, (list, tuple)) else (a,))]
        tens = [a for a in flat if torch.is_tensor(a)]
        slices = [a for a in args if isinstance(a, ChanSlice)]
        first = tens[0] if tens else (slices[0].t if slices else None)
        if first is None or not first.is_cuda:
            return fn(*args, **kw)
        nbytes = sum(t.numel() * t.element_size() for t in tens)
        nbytes += sum(c.n * c.c * c.hw[0] * c.hw[1] * c.t.element_size() for c in slices)
        # an event pair around ONE launch of a 2-5 us kernel reads 15-25 us (the pair's own granularity): the launch is
        # repeated back to back -- these kernels are pure functions of their operands (no atomics, nothing accumulated unless
        # asked) -- and the pair divided by the count
        reps = HBM_PROFILER_REPEATS
        if kw.get("accumulate") or kw.get("accumulate_dx") or (fn.__name__ == "gap_bwd" and len(args) > 2 and args[2]) or \
                (fn.__name__ == "linear_bwd" and len(args) > 6 and args[6]):
            reps = 1
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*args, **kw)
        for _ in range(reps - 1):
            fn(*args, **kw)
        e.record()
        prof.append((fn.__name__, nbytes, s, e, reps))
        return out
    return wrapper


# ------------------------------------------------------------------------------------ backbone
def bn_fold(conv_bias, gamma, beta, mean, var, eps, scale, shift):
    lib = _check(conv_bias, gamma, beta, mean, var, scale, shift)
    lib.call("ssn_bn_fold", _p(conv_bias), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), _p(scale),
             _p(shift), gamma.numel(), _stream(lib, gamma))


def packed_floats(cout, cin, ksize, transposed, x6=False):
    fn = _lib.get_lib().cdll.ssn_conv_x6_packed_floats if x6 else _lib.get_lib().cdll.ssn_conv_packed_floats
    return int(fn(cout, cin, ksize, int(bool(transposed))))


def dgrad_layout(ksize, stride, pad, h, w):
    """Packed-weight layout ssn_conv_dgrad wants for this conv (2 = parity-ordered stride-2 path, else 1)."""
    return int(_lib.get_lib().cdll.ssn_conv_dgrad_layout(ksize, stride, pad, h, w))


def pack_weights(w, transposed, out=None):
    """Re-lay a torch-layout weight [Cout, Cin, k, k] into the slab/lane-half order the conv kernels read.

    transposed=False/0 -> forward operand, True/1 -> dgrad operand, 2 -> parity-ordered stride-2 dgrad operand
    (use dgrad_layout() to choose between 1 and 2).
    """
    lib = _check(w, out)
    cout, cin, k, _ = w.shape
    n = packed_floats(cout, cin, k, transposed)
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    assert out.numel() >= n
    lib.call("ssn_conv_pack_weights", _p(w), _p(out), cout, cin, k, int(transposed), _stream(lib, w))
    return out


_PACK_BATCH = None      # the open PackBatch, if any


def _pack_alloc(nfloats, device):
    """Operand buffer of a packing call: fresh, or -- inside a PackBatch -- the persistent one of this position in the batch."""
    if _PACK_BATCH is not None:
        return _PACK_BATCH.take(int(nfloats), device)
    return torch.empty(int(nfloats), device=device, dtype=torch.float32)


class PackBatch:
    """``with PackBatch(state, device):`` -- the x6 / planes weight-packing calls inside (pack_weights_multi(x6=True), pack_rect_multi,
    pack_weights_rect, pack_dgrad_s2, pack_dgrad_rect, s2d_weights' output) are only RECORDED; leaving the block issues all of them
    in three launches through a device-resident plan (ssn_conv_x6_pack_batch_end).  `state` is a dict the caller keeps (one per
    backbone and kind of pass): it holds the operand buffers -- the k-th allocation of a batch gets the same buffer every time, so
    the recorded entries repeat from step to step and the plan is written once -- and the plan buffer.  The returned operands are
    valid after the block; they are overwritten by the next batch on the same state (same weights -> same contents)."""

    def __init__(self, state, device):
        self.state, self.device, self.k = state, device, 0

    def take(self, nfloats, device):
        bufs = self.state.setdefault("bufs", [])
        if self.k < len(bufs) and bufs[self.k].numel() == nfloats and bufs[self.k].device == device:
            t = bufs[self.k]
        else:
            t = torch.empty(nfloats, device=device, dtype=torch.float32)
            del bufs[self.k:]
            bufs.append(t)
        self.k += 1
        return t

    def __enter__(self):
        global _PACK_BATCH
        assert _PACK_BATCH is None, "PackBatch blocks do not nest"
        self.lib = _lib.get_lib()
        self.lib.call("ssn_conv_x6_pack_batch_begin")
        _PACK_BATCH = self
        return self

    def __exit__(self, et, ev, tb):
        global _PACK_BATCH
        _PACK_BATCH = None
        if et is not None:
            self.lib.cdll.ssn_conv_x6_pack_batch_abort()
            return False
        need = int(self.lib.cdll.ssn_conv_x6_pack_batch_entries()) * int(self.lib.cdll.ssn_conv_x6_pack_entry_bytes())
        plan, fresh = self.state.get("plan"), 0
        if need and (plan is None or plan.numel() < need or plan.device != self.device):
            plan = self.state["plan"] = torch.empty(max(need, 1 << 16), device=self.device, dtype=torch.uint8)
            fresh = 1
        self.lib.call("ssn_conv_x6_pack_batch_end", _p(plan), plan.numel() if plan is not None else 0, fresh,
                      _stream(self.lib, plan) if plan is not None else None)
        return False


def pack_weights_multi(entries, x6=False):
    """entries: list of (weights, mode) with weights = [w], [wA, wB] (fused pair) or -- x6 only -- [wA, wB, wC]
    ... [wA, wB, wC, wD] (concatenated output channels).

    Returns one packed tensor per entry (views of a single flat buffer); ceil(len / 40) launches in total.
    x6=True: scale + split every weight into two f16 planes for the conv_x6 kernels (modes 0/1, ksize 1/3 only).
    """
    if not entries:
        return []
    n = len(entries)
    lib = _check(*[w for ws, _ in entries for w in ws])
    first = entries[0][0][0]
    couts = [sum(w.shape[0] for w in ws) for ws, _ in entries]
    cins = [ws[0].shape[1] for ws, _ in entries]
    kss = [ws[0].shape[2] for ws, _ in entries]
    sizes = [packed_floats(co, ci, k, m, x6) for co, ci, k, (_, m) in zip(couts, cins, kss, entries)]
    flat = _pack_alloc(sum(sizes), first.device) if x6 else torch.empty(sum(sizes), device=first.device, dtype=torch.float32)
    outs, off = [], 0
    for sz in sizes:
        outs.append(flat[off:off + sz])
        off += sz
    w0 = _ptr_array([ws[0] for ws, _ in entries])
    w1 = _ptr_array([ws[1] if len(ws) > 1 else None for ws, _ in entries])
    po = _ptr_array(outs)
    ia = lambda vals: (ctypes.c_int * n)(*[int(v) for v in vals])   # noqa: E731
    a_cout, a_cin, a_ks = ia(couts), ia(cins), ia(kss)
    a_mode = ia([m for _, m in entries])
    a_split = ia([ws[0].shape[0] if len(ws) > 1 else co for (ws, _), co in zip(entries, couts)])
    if x6:
        assert all(len(ws) <= 4 for ws, _ in entries)
        w2 = _ptr_array([ws[2] if len(ws) > 2 else None for ws, _ in entries])
        w3 = _ptr_array([ws[3] if len(ws) > 3 else None for ws, _ in entries])
        cum = lambda ws, k: sum(w.shape[0] for w in ws[:k])   # noqa: E731
        a_split2 = ia([cum(ws, 2) if len(ws) > 2 else co for (ws, _), co in zip(entries, couts)])
        a_split3 = ia([cum(ws, 3) if len(ws) > 3 else co for (ws, _), co in zip(entries, couts)])
        lib.call("ssn_conv_x6_pack_weights_multi", n, ctypes.addressof(w0), ctypes.addressof(w1), ctypes.addressof(w2),
                 ctypes.addressof(w3), ctypes.addressof(po), ctypes.addressof(a_cout), ctypes.addressof(a_cin),
                 ctypes.addressof(a_ks), ctypes.addressof(a_mode), ctypes.addressof(a_split), ctypes.addressof(a_split2),
                 ctypes.addressof(a_split3), _stream(lib, first))
    else:
        assert all(len(ws) <= 2 for ws, _ in entries)
        lib.call("ssn_conv_pack_weights_multi", n, ctypes.addressof(w0), ctypes.addressof(w1), ctypes.addressof(po),
                 ctypes.addressof(a_cout), ctypes.addressof(a_cin), ctypes.addressof(a_ks), ctypes.addressof(a_mode),
                 ctypes.addressof(a_split), _stream(lib, first))
    return outs


def conv_fwd(x, w_packed, scale, shift, y, ksize, stride, pad, relu=True, tile_cfg=-1):
    """x, y: ChanSlice.  w_packed: pack_weights(w, transposed=False)."""
    lib = _check(x, w_packed, scale, shift, y)
    h, wd = x.hw
    ho, wo = y.hw
    assert w_packed.numel() >= packed_floats(y.c, x.c, ksize, False)
    lib.call("ssn_conv_bn_relu_fwd", _p(x), _p(w_packed), _p(scale), _p(shift), _p(y), x.n, x.c, h, wd,
             x.img_stride, y.c, ho, wo, y.img_stride, ksize, stride, pad, int(relu), tile_cfg, _amax_out(y),
             _stream(lib, w_packed))


def guard_bytes(cs):
    """Readable bytes in front of a ChanSlice: the channels below it, plus the pad of tensors from `guarded_empty`."""
    base = cs.t.untyped_storage().data_ptr()
    return int(min(cs.ptr - base, 1 << 20)) if base else 0


def guarded_empty(shape, device, guard_floats=64):
    """torch.empty with `guard_floats` readable floats in front of element 0 (see ssn_conv_x6_fwd: x_guard_bytes)."""
    n = 1
    for d in shape:
        n *= int(d)
    flat = torch.empty(n + guard_floats, device=device, dtype=torch.float32)
    return flat[guard_floats:].view(*shape)


def conv_x6_fwd(x, w_packed, scale, shift, y, ksize, stride, pad, relu=True, tile_cfg=-1, raw_from=0, row_split=0, row_gap=0):
    """conv_fwd on the f16 matrix cores (2-way split, fp32-class accuracy).  w_packed: pack_weights_multi(x6=True).
    raw_from > 0: output channels >= raw_from take neither the affine nor the ReLU.  row_gap > 0: output channels >= row_split
    are stored row_gap channels further up y's tensor (y: the ChanSlice of the first row_split channels' home)."""
    lib = _check(x, w_packed, scale, shift, y)
    h, wd = x.hw
    ho, wo = y.hw
    assert w_packed.numel() >= packed_floats(y.c, x.c, ksize, False, True)
    xa = _amax_in(x)
    lib.call("ssn_conv_x6_fwd", _p(x), _p(w_packed), _p(scale), _p(shift), _p(y), x.n, x.c, h, wd,
             x.img_stride, y.c, ho, wo, y.img_stride, ksize, stride, pad, int(relu), guard_bytes(x), tile_cfg,
             _p(xa), _amax_out(y), int(raw_from), int(row_split), int(row_gap),
             _p(_slot_ext(y)) if row_gap else None, _stream(lib, w_packed))


def pack_weights_rect(w):
    """Split + pack one [cout, cin, kh, kw] forward weight for conv_x6_fwd_rect."""
    lib = _check(w)
    cout, cin, kh, kw = w.shape
    out = _pack_alloc(int(lib.cdll.ssn_conv_x6_packed_floats_rect(cout, cin, kh, kw)), w.device)
    lib.call("ssn_conv_x6_pack_weights_rect", _p(w.contiguous()), _p(out), cout, cin, kh, kw, _stream(lib, w))
    return out


def conv_x6_fwd_rect(x, w_packed, scale, shift, y, kh, kw, pad_h, pad_w, relu=True, tile_cfg=-1):
    """Stride-1 forward convolution with kh x kw taps (5x5, 1x7, 7x1, 1x3, 3x1) on the split kernel."""
    lib = _check(x, w_packed, scale, shift, y)
    h, wd = x.hw
    ho, wo = y.hw
    xa = _amax_in(x)
    lib.call("ssn_conv_x6_fwd_rect", _p(x), _p(w_packed), _p(scale), _p(shift), _p(y), x.n, x.c, h, wd,
             x.img_stride, y.c, ho, wo, y.img_stride, kh, kw, pad_h, pad_w, int(relu), guard_bytes(x), tile_cfg,
             _p(xa), _amax_out(y), _stream(lib, w_packed))


def conv_x6_dgrad(dy, wt, dx, ksize, pad, accumulate, tile_cfg=-1, mask_y=None, mask_scale=None, k_split=0, k_gap=0):
    """Stride-1 conv_dgrad on the f16 matrix cores.  wt: pack_weights_multi([... mode 1], x6=True).
    k_gap > 0: channels >= k_split of dy sit k_gap channels further up its tensor (dy.c counts the channels read)."""
    lib = _check(dy, wt, dx, mask_y, mask_scale)
    assert wt.numel() >= packed_floats(dy.c, dx.c, ksize, True, True)
    ho, wo = dy.hw
    h, w = dx.hw
    ga = _amax_in(dy)
    lib.call("ssn_conv_x6_dgrad", _p(dy), _p(wt), _p(dx), dy.n, dy.c, ho, wo, dy.img_stride, dx.c, h, w,
             dx.img_stride, ksize, pad, int(accumulate), _p(mask_y),
             mask_y.img_stride if mask_y is not None else 0, _p(mask_scale), guard_bytes(dy), tile_cfg,
             _p(ga), _amax_out(dx), int(k_split), int(k_gap), _p(_slot_ext(dy)) if k_gap else None, _stream(lib, wt))


def pack_dgrad_s2(w):
    """dgrad operand of a 3x3 / stride-2 / pad-1 conv for conv_x6_dgrad_s2 (four parity-class sections)."""
    lib = _check(w)
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3)
    out = _pack_alloc(int(lib.cdll.ssn_conv_x6_dgrad_s2_packed_floats(cout, cin)), w.device)
    lib.call("ssn_conv_x6_pack_dgrad_s2", _p(w.contiguous()), _p(out), cout, cin, _stream(lib, w))
    return out


def pack_dgrad_rect(w):
    """dgrad operand of a stride-1 layer with rectangular taps for conv_x6_dgrad_rect (transposed, taps reversed)."""
    lib = _check(w)
    cout, cin, kh, kw = w.shape
    out = _pack_alloc(int(lib.cdll.ssn_conv_x6_packed_floats_dgrad_rect(cout, cin, kh, kw)), w.device)
    lib.call("ssn_conv_x6_pack_dgrad_rect", _p(w.contiguous()), _p(out), cout, cin, kh, kw, _stream(lib, w))
    return out


def pack_rect_multi(ws, dgrad=False):
    """Rectangular-tap weights [cout, cin, kh, kw] -> their packed operands (views of one flat buffer) in ceil(len / 40) x 3
    launches: forward operands (pack_weights_rect) or, with dgrad, the transposed tap-reversed ones (pack_dgrad_rect)."""
    if not ws:
        return []
    lib = _check(*ws)
    n = len(ws)
    size_fn = lib.cdll.ssn_conv_x6_packed_floats_dgrad_rect if dgrad else lib.cdll.ssn_conv_x6_packed_floats_rect
    sizes = [int(size_fn(*w.shape)) for w in ws]
    flat = _pack_alloc(sum(sizes), ws[0].device)
    outs, off = [], 0
    for sz in sizes:
        outs.append(flat[off:off + sz])
        off += sz
    ws = [w.contiguous() for w in ws]
    ia = lambda vals: (ctypes.c_int * n)(*[int(v) for v in vals])   # noqa: E731
    pw, po = _ptr_array(ws), _ptr_array(outs)
    a_cout, a_cin, a_kh, a_kw = (ia([w.shape[d] for w in ws]) for d in range(4))
    a_mode = ia([2 if dgrad else 0] * n)
    lib.call("ssn_conv_x6_pack_rect_multi", n, ctypes.addressof(pw), ctypes.addressof(po), ctypes.addressof(a_cout),
             ctypes.addressof(a_cin), ctypes.addressof(a_kh), ctypes.addressof(a_kw), ctypes.addressof(a_mode), _stream(lib, ws[0]))
    return outs


def conv_x6_dgrad_rect(dy, wt, dx, kh, kw, pad_h, pad_w, accumulate, tile_cfg=-1, mask_y=None, mask_scale=None):
    """dgrad of a stride-1 same-size layer with kh x kw taps on the f16 matrix cores.  wt: pack_dgrad_rect(w)."""
    lib = _check(dy, wt, dx, mask_y, mask_scale)
    h, w = dx.hw
    assert dy.hw == dx.hw
    ga = _amax_in(dy)
    lib.call("ssn_conv_x6_dgrad_rect", _p(dy), _p(wt), _p(dx), dy.n, dy.c, h, w, dy.img_stride, dx.c, dx.img_stride,
             kh, kw, pad_h, pad_w, int(accumulate), _p(mask_y), mask_y.img_stride if mask_y is not None else 0,
             _p(mask_scale), guard_bytes(dy), tile_cfg, _p(ga), _amax_out(dx), _stream(lib, wt))


def conv_x6_dgrad_s2(dy, wt, dx, accumulate, tile_cfg=-1, mask_y=None, mask_scale=None, pad=1):
    """dgrad of a 3x3 / stride-2 conv on the f16 matrix cores: pad 1 (even input size) or pad 0.  wt: pack_dgrad_s2(w)."""
    lib = _check(dy, wt, dx, mask_y, mask_scale)
    ho, wo = dy.hw
    h, w = dx.hw
    ga = _amax_in(dy)
    lib.call("ssn_conv_x6_dgrad_s2", _p(dy), _p(wt), _p(dx), dy.n, dy.c, ho, wo, dy.img_stride, dx.c, h, w,
             dx.img_stride, int(accumulate), _p(mask_y), mask_y.img_stride if mask_y is not None else 0,
             _p(mask_scale), guard_bytes(dy), tile_cfg, _p(ga), _amax_out(dx), int(pad), _stream(lib, wt))


def relu_bn_bwd(dy, y, scale):
    """In place: dy <- dy * (y > 0) * scale[c].  dy, y: ChanSlice with equal channel counts."""
    lib = _check(dy, y, scale)
    h, w = y.hw
    lib.call("ssn_relu_bn_bwd", _p(dy), _p(y), _p(scale), y.n, y.c, h * w, dy.img_stride, y.img_stride,
             _amax_out(dy), _stream(lib, scale))


def conv_dgrad(dy, wt, dx, ksize, stride, pad, accumulate, tile_cfg=-1, mask_y=None, mask_scale=None, wt_layout=1):
    """dy: ChanSlice (grad of conv output), dx: ChanSlice (grad of conv input), wt: pack_weights(w, True).

    mask_y (ChanSlice like dx) + mask_scale [dx.c]: fuse the ReLU+frozen-BN backward of dx's tensor into the store.
    """
    lib = _check(dy, wt, dx, mask_y, mask_scale)
    assert wt.numel() >= packed_floats(dy.c, dx.c, ksize, True)
    ho, wo = dy.hw
    h, w = dx.hw
    lib.call("ssn_conv_dgrad", _p(dy), _p(wt), _p(dx), dy.n, dy.c, ho, wo, dy.img_stride, dx.c, h, w,
             dx.img_stride, ksize, stride, pad, int(accumulate), _p(mask_y),
             mask_y.img_stride if mask_y is not None else 0, _p(mask_scale), int(wt_layout), tile_cfg,
             _amax_out(dx), _stream(lib, wt))


def wgrad_workspace_bytes(n, cin, cout, ho, wo, ksize, tile_cfg=-1):
    return _lib.get_lib().wgrad_workspace_bytes(n, cin, cout, ho, wo, ksize, tile_cfg)


def conv_wgrad(g, x, dw, db, ksize, stride, pad, workspace, tile_cfg=-1):
    """g: ChanSlice (masked grad of conv output), x: ChanSlice (conv input); dw [Cout,Cin,k,k], db [Cout] or None."""
    lib = _check(g, x, dw, db, workspace)
    ho, wo = g.hw
    h, w = x.hw
    lib.call("ssn_conv_wgrad", _p(g), _p(x), _p(dw), _p(db), x.n, x.c, h, w, x.img_stride, g.c, ho, wo,
             g.img_stride, ksize, stride, pad, _p(workspace), workspace.numel() * workspace.element_size(),
             tile_cfg, _stream(lib, dw))


def wgrad_x6_supported(ksize, stride, pad, h, w, guard_bytes=256):
    """Layers the x6 weight-gradient kernel takes (the others stay on the exact-f32 kernel); guard_bytes: the readable bytes
    in front of the layer's input."""
    return ksize in (1, 3) and stride == 1 and 2 * pad == ksize - 1 and (pad * w + pad) * 4 <= guard_bytes


def space_to_depth2(x, guard_floats=256):
    """x [N, C, H, W] -> xs [N, 4C, H/2, W/2] (tracked: amax slot attached; `guard_floats` readable floats in front)."""
    lib = _check(x)
    n, c, h, w = x.shape
    xs = attach_amax(guarded_empty((n, 4 * c, h // 2, w // 2), x.device, guard_floats))
    lib.call("ssn_space_to_depth2", _p(x), _p(xs), n, c, h, w, _amax_out(xs), _stream(lib, x))
    return xs


def s2d_weights(w):
    """[Cout, C, k, k] (k odd, stride-2 layer) -> [Cout, 4C, (k+1)/2, (k+1)/2] for the space-to-depth input."""
    lib = _check(w)
    cout, c, k, _ = w.shape
    # (inside a PackBatch the result is an operand of a recorded packing call: it gets a persistent buffer like the packed ones)
    w2 = _pack_alloc(cout * 4 * c * ((k + 1) // 2) ** 2, w.device).view(cout, 4 * c, (k + 1) // 2, (k + 1) // 2)
    lib.call("ssn_s2d_weights", _p(w.contiguous()), _p(w2), cout, c, k, _stream(lib, w))
    return w2


def s2d_weights_bwd(dw2, dw):
    """Gradient of s2d_weights: gathers dw [Cout, C, k, k] out of dw2 [Cout, 4C, (k+1)/2, (k+1)/2]."""
    lib = _check(dw2, dw)
    cout, c, k, _ = dw.shape
    lib.call("ssn_s2d_weights_bwd", _p(dw2), _p(dw), cout, c, k, _stream(lib, dw))


def wgrad_x6_workspace_bytes(n, cin, cout, h, w, ksize, tile_cfg=-1):
    return int(_lib.get_lib().cdll.ssn_conv_wgrad_x6_workspace_bytes(n, cin, cout, h, w, ksize, tile_cfg))


def conv_wgrad_x6(g, x, dw, db, ksize, pad, workspace, tile_cfg=-1, g_row_split=0, g_row_gap=0):
    """conv_wgrad on the f16 matrix cores (stride 1, same size).  x needs >= 256 readable bytes in front of it.
    g_row_gap > 0: rows >= g_row_split of g sit g_row_gap channels further up its tensor (g.c counts the rows read)."""
    lib = _check(g, x, dw, db, workspace)
    h, w = x.hw
    assert g.hw == x.hw
    ga, xa = _amax_in(g), _amax_in(x)
    lib.call("ssn_conv_wgrad_x6", _p(g), _p(x), _p(dw), _p(db), x.n, x.c, h, w, x.img_stride, g.c, g.img_stride,
             ksize, pad, guard_bytes(x), _p(workspace), workspace.numel() * workspace.element_size(), tile_cfg,
             _p(ga), _p(xa), int(g_row_split), int(g_row_gap), _p(_slot_ext(g)) if g_row_gap else None, _stream(lib, dw))


def wgrad_x6_rect_workspace_bytes(n, cin, cout, h, w, kh, kw, tile_cfg=-1):
    return int(_lib.get_lib().cdll.ssn_conv_wgrad_x6_rect_workspace_bytes(n, cin, cout, h, w, kh, kw, tile_cfg))


def wgrad_x6_rect_guard_floats(pad_h, pad_w, w):
    """Readable floats conv_wgrad_x6_rect needs in front of x (the taps in front of a pixel reach that far)."""
    return max(64, ((pad_h * w + pad_w) * 4 + 255) // 256 * 64)


def conv_wgrad_x6_rect(g, x, dw, db, kh, kw, pad_h, pad_w, workspace, tile_cfg=-1):
    """conv_wgrad of a stride-1 same-size layer with kh x kw taps on the f16 matrix cores; dw [Cout, Cin, kh, kw]."""
    lib = _check(g, x, dw, db, workspace)
    h, w = x.hw
    assert g.hw == x.hw
    ga, xa = _amax_in(g), _amax_in(x)
    lib.call("ssn_conv_wgrad_x6_rect", _p(g), _p(x), _p(dw), _p(db), x.n, x.c, h, w, x.img_stride, g.c, g.img_stride,
             kh, kw, pad_h, pad_w, guard_bytes(x), _p(workspace), workspace.numel() * workspace.element_size(), tile_cfg,
             _p(ga), _p(xa), _stream(lib, dw))


def pool_fwd(kind, x, y, argmax, ksize, stride, pad):
    lib = _check(x, y, argmax)
    h, w = x.hw
    ho, wo = y.hw
    lib.call("ssn_pool_fwd", int(kind == "max"), _p(x), _p(y), _p(argmax), x.n, x.c, h, w, x.img_stride, ho, wo,
             y.img_stride, ksize, stride, pad, _amax_out(y), _stream(lib, x))


def pool_bwd(kind, dy, argmax, dx, ksize, stride, pad, accumulate, mask_y=None, mask_scale=None):
    lib = _check(dy, dx, argmax, mask_y, mask_scale)
    h, w = dx.hw
    ho, wo = dy.hw
    lib.call("ssn_pool_bwd", int(kind == "max"), _p(dy), _p(argmax), _p(dx), dx.n, dx.c, h, w, dx.img_stride, ho,
             wo, dy.img_stride, ksize, stride, pad, int(accumulate), _p(mask_y),
             mask_y.img_stride if mask_y is not None else 0, _p(mask_scale), _amax_out(dx), _stream(lib, dx))


def avgpool_affine_fwd(x, y, scale, shift, relu, ksize, stride, pad):
    """y = relu?(scale[c] * avgpool(x) + shift[c]) on ChanSlices (see ssn_avgpool_affine_fwd)."""
    lib = _check(x, y, scale, shift)
    h, w = x.hw
    ho, wo = y.hw
    lib.call("ssn_avgpool_affine_fwd", _p(x), _p(y), _p(scale), _p(shift), int(bool(relu)), x.n, x.c, h, w,
             x.img_stride, ho, wo, y.img_stride, ksize, stride, pad, _amax_out(y), _stream(lib, y))


def channel_sum_workspace_bytes(n, c):
    return 4 * c * int(_lib.get_lib().cdll.ssn_channel_sum_shares(n))


def channel_sum(g, out, workspace):
    """out[c] = sum over images and pixels of the ChanSlice g (two passes through `workspace`, fixed order)."""
    lib = _check(g, out, workspace)
    h, w = g.hw
    lib.call("ssn_channel_sum", _p(g), _p(out), g.n, g.c, h * w, g.img_stride, _p(workspace),
             workspace.numel() * workspace.element_size(), _stream(lib, out))


def bn_train_workspace_floats(n, c):
    fn = _lib.get_lib().cdll.ssn_bn_train_workspace_floats
    fn.restype = ctypes.c_long
    return int(fn(int(n), int(c)))


def bn_train_stats(z, conv_bias, mean, invstd, running_mean, running_var, eps, momentum, workspace):
    """Batch statistics of the ChanSlice z (the convolution WITHOUT its bias) + running-statistics update."""
    global PARAM_EPOCH
    PARAM_EPOCH += 1        # (the running statistics are written through raw pointers)
    lib = _check(z, conv_bias, mean, invstd, running_mean, running_var, workspace)
    h, w = z.hw
    lib.call("ssn_bn_train_stats", _p(z), _p(conv_bias), _p(mean), _p(invstd), _p(running_mean), _p(running_var), z.n,
             z.c, h * w, z.img_stride, float(eps), float(momentum), _p(workspace),
             workspace.numel() * workspace.element_size(), _stream(lib, mean))


def bn_train_apply(z, y, mean, invstd, gamma, beta, relu=True):
    """y = relu?(gamma * (z - mean) * invstd + beta) on ChanSlices."""
    lib = _check(z, y, mean, invstd, gamma, beta)
    h, w = z.hw
    lib.call("ssn_bn_train_apply", _p(z), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), int(bool(relu)), z.n, z.c,
             h * w, z.img_stride, y.img_stride, _amax_out(y), _stream(lib, mean))


def bn_train_bwd(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, workspace, relu=True):
    """Backward of bn_train_apply(bn_train_stats(z)): dgamma, dbeta [C] and dz (ChanSlice)."""
    lib = _check(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, workspace)
    h, w = z.hw
    lib.call("ssn_bn_train_bwd", _p(dy), _p(y), _p(z), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dz),
             int(bool(relu)), z.n, z.c, h * w, dy.img_stride, y.img_stride, z.img_stride, dz.img_stride, _p(workspace),
             workspace.numel() * workspace.element_size(), _amax_out(dz), _stream(lib, mean))


@_hbm_timed
def gap_fwd(x, y):
    lib = _check(x, y)
    h, w = x.hw
    lib.call("ssn_global_avgpool_fwd", _p(x), _p(y), x.n, x.c, h * w, x.img_stride, _stream(lib, y))


@_hbm_timed
def gap_bwd(dy, dx, accumulate=False):
    lib = _check(dy, dx)
    h, w = dx.hw
    lib.call("ssn_global_avgpool_bwd", _p(dy), _p(dx), dx.n, dx.c, h * w, dx.img_stride, int(accumulate),
             _amax_out(dx), _stream(lib, dy))


@_hbm_timed
def dropout_fwd(x, y, mask, p, seed, counter=None):
    """counter: optional int64[1] device tensor; it is mixed into the Philox key and incremented by the call."""
    lib = _check(x, y, mask, counter)
    lib.call("ssn_dropout_fwd", _p(x), _p(y), _p(mask), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
             _p(counter), _stream(lib, x))


@_hbm_timed
def dropout_bwd(dy, mask, dx, p):
    lib = _check(dy, mask, dx)
    lib.call("ssn_dropout_bwd", _p(dy), _p(mask), _p(dx), dy.numel(), float(p), _stream(lib, dy))


# ------------------------------------------------------------------------------------ STPP / heads / losses
def make_stpp_table(parts, n_seg, act_lo, act_hi):
    """parts: list of (lo, hi, norm, col)."""
    if len(parts) > STPP_MAX_PARTS:
        raise ValueError("STPP config has %d parts (max %d)" % (len(parts), STPP_MAX_PARTS))
    t = SsnStppTable()
    t.n_parts, t.n_seg, t.act_lo, t.act_hi = len(parts), n_seg, act_lo, act_hi
    for i, (lo, hi, norm, col) in enumerate(parts):
        t.lo[i], t.hi[i], t.norm[i], t.col[i] = lo, hi, norm, col
    return t


@_hbm_timed
def stpp_fwd(ft, scaling, act_ft, stpp_ft, table):
    lib = _check(ft, scaling, act_ft, stpp_ft)
    lib.call("ssn_stpp_fwd", _p(ft), _p(scaling), _p(act_ft), _p(stpp_ft), act_ft.shape[0], ft.shape[1],
             ctypes.addressof(table), _stream(lib, ft))


@_hbm_timed
def stpp_bwd(d_act, d_stpp, scaling, d_ft, table):
    lib = _check(d_act, d_stpp, scaling, d_ft)
    lib.call("ssn_stpp_bwd", _p(d_act), _p(d_stpp), _p(scaling), _p(d_ft), d_stpp.shape[0], d_ft.shape[1],
             ctypes.addressof(table), _stream(lib, d_ft))


def _heads_tables(ws, bs, pos, idx, outs):
    arr3 = ctypes.c_void_p * 3
    ci3 = ctypes.c_int * 3
    return (arr3(*[None if t is None else t.data_ptr() for t in ws]), arr3(*[None if t is None else t.data_ptr() for t in bs]),
            arr3(*[None if t is None else t.data_ptr() for t in pos]), arr3(*[None if t is None else t.data_ptr() for t in idx]),
            arr3(*[None if t is None else t.data_ptr() for t in outs]),
            ci3(*[0 if w is None else int(w.shape[0]) for w in ws]), ci3(*[0 if t is None else int(t.numel()) for t in idx]))


@_hbm_timed
def heads_fwd(ft, scaling, ws, bs, pos, idx, outs, act_ft, stpp_ft, table):
    """STPP + the three Linear heads + the prop_type row selection in one launch (ssn_heads_fwd).  ws / bs / pos / idx / outs:
    lists of 3 (activity, completeness, regression; entries of an absent regression head None)."""
    lib = _check(ft, scaling, act_ft, stpp_ft, *[t for t in list(ws) + list(bs) + list(outs) if t is not None])
    tabs = _heads_tables(ws, bs, pos, idx, outs)
    lib.call("ssn_heads_fwd", _p(ft), _p(scaling), *[ctypes.addressof(t) for t in tabs], _p(act_ft), _p(stpp_ft),
             act_ft.shape[0], ft.shape[1], ctypes.addressof(table), _stream(lib, ft))


@_hbm_timed
def heads_bwd(scaling, ws, bs, pos, idx, douts, act_ft, stpp_ft, table, d_ft, dws, dbs):
    lib = _check(scaling, act_ft, stpp_ft, d_ft, *[t for t in list(ws) + list(douts) + list(dws) + list(dbs) if t is not None])
    tabs = _heads_tables(ws, bs, pos, idx, douts)
    arr3 = ctypes.c_void_p * 3
    pdw = arr3(*[None if t is None else t.data_ptr() for t in dws])
    pdb = arr3(*[None if t is None else t.data_ptr() for t in dbs])
    lib.call("ssn_heads_bwd", None, _p(scaling), *[ctypes.addressof(t) for t in tabs], _p(act_ft), _p(stpp_ft), act_ft.shape[0],
             d_ft.shape[1], ctypes.addressof(table), _p(d_ft), ctypes.addressof(pdw), ctypes.addressof(pdb), _stream(lib, d_ft))


def stpp_reorg(scores, ranges, act_range, scaling, part_scale_col, act_len, comp_len, reg_len, out_act, out_comp,
               out_reg):
    lib = _check(ranges, act_range, scaling, part_scale_col, out_act, out_comp, out_reg)
    # scores: [T, D], or a column range of one (rows D_total apart): the kernel addresses row r at r * (row stride)
    assert scores.dim() == 2 and scores.stride(1) == 1 and scores.dtype == torch.float32
    assert scores.stride(0) >= act_len + ranges.shape[1] * (comp_len + (reg_len if out_reg is not None else 0))
    lib.call("ssn_stpp_reorg", _p(scores), scores.shape[0], scores.stride(0), _p(ranges), _p(act_range),
             _p(scaling), _p(part_scale_col), ranges.shape[0], ranges.shape[1], act_len, comp_len, reg_len,
             _p(out_act), _p(out_comp), _p(out_reg), _stream(lib, scores))


def crop_mean(x, num_crop, out):
    """x [num_crop * T, D] (crop-major) -> out [T, D] = mean over the crops (ssn_test.py:85)."""
    lib = _check(x, out)
    t = out.shape[0]
    assert x.shape[0] == num_crop * t and x.shape[1] == out.shape[1]
    lib.call("ssn_crop_mean", _p(x), _p(out), num_crop, t, out.shape[1], _stream(lib, x))


def reg_denorm(reg, mean0, std0, mean1, std1):
    """reg [..., 2] in place: reg[..., k] * std[k] + mean[k] (ssn_test.py:88-90)."""
    lib = _check(reg)
    assert reg.shape[-1] == 2 and reg.is_contiguous()
    lib.call("ssn_reg_denorm", _p(reg), reg.numel() // 2, float(mean0), float(std0), float(mean1), float(std1),
             _stream(lib, reg))


def frames_crop_normalize(src, dst, crops, roll, invert_even, mean, std):
    """src uint8 [n_img, Hs, Ws, C] -> dst fp32 [n_crops, n_img, C, ch, cw]; crops = [(off_x, off_y, flip), ...];
    mean / std: device float tensors (repeat over the stacked channels like GroupNormalize)."""
    import ctypes
    lib = _check(src, dst, mean, std)
    assert src.dtype == torch.uint8 and src.dim() == 4 and dst.dim() == 5 and src.is_contiguous() and dst.is_contiguous()
    n_img, hs, ws, c = src.shape
    nc, _, _, ch, cw = dst.shape
    assert nc == len(crops) and dst.shape[1] == n_img and dst.shape[2] == c
    arr = ctypes.c_int * nc
    ox, oy, fl = arr(*[int(t[0]) for t in crops]), arr(*[int(t[1]) for t in crops]), arr(*[int(bool(t[2])) for t in crops])
    lib.call("ssn_frames_crop_normalize", _p(src), _p(dst), n_img, hs, ws, c, ch, cw, nc,
             ctypes.cast(ox, ctypes.c_void_p), ctypes.cast(oy, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p),
             int(roll), int(invert_even), _p(mean), mean.numel(), _p(std), std.numel(), _stream(lib, src))


def frames_crop_resize_normalize(src, boxes, flips, out_hw, roll, invert_even, mean, std, dst=None):
    """The training augmentation in two launches (ssn_frames_crop_resize_normalize): src uint8 [n_img, Hs, Ws, C] decoded frames
    -> fp32 [n_img, C, out_h, out_w] = normalise(roll(flip(PIL-bilinear-resize(crop)))).  boxes: [n_img, 4] ints (x0, y0, crop_w,
    crop_h) -- host list / array (validated and uploaded here) or a device int32 tensor the caller vouches for; flips likewise."""
    lib = _check(src, mean, std)
    assert src.dtype == torch.uint8 and src.dim() == 4 and src.is_contiguous()
    n_img, hs, ws, c = src.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    if not torch.is_tensor(boxes):
        import numpy as np
        b = np.asarray(boxes, dtype=np.int64).reshape(n_img, 4)
        if (b[:, 0] < 0).any() or (b[:, 1] < 0).any() or (b[:, 2] < 1).any() or (b[:, 3] < 1).any() \
                or (b[:, 0] + b[:, 2] > ws).any() or (b[:, 1] + b[:, 3] > hs).any():
            raise ValueError("crop box outside the %dx%d frame" % (ws, hs))
        if (b[:, 2] > 3 * ow).any() or (b[:, 3] > 3 * oh).any():
            raise ValueError("crop more than 3x the output size (the resize kernel holds 7 taps per axis)")
        boxes = torch.from_numpy(b.astype(np.int32)).to(src.device)
    if not torch.is_tensor(flips):
        flips = torch.tensor([int(bool(f)) for f in flips], dtype=torch.int32).to(src.device)
    assert boxes.dtype == torch.int32 and flips.dtype == torch.int32 and boxes.numel() == 4 * n_img and flips.numel() == n_img
    if dst is None:
        dst = torch.empty((n_img, c, oh, ow), device=src.device, dtype=torch.float32)
    ws_bytes = int(lib.cdll.ssn_frames_resize_workspace_bytes(n_img, oh, ow))
    wsp = torch.empty(max(ws_bytes, 4) // 4, device=src.device, dtype=torch.int32)
    lib.call("ssn_frames_crop_resize_normalize", _p(src), _p(dst), n_img, hs, ws, c, oh, ow, _p(boxes), _p(flips), int(roll),
             int(invert_even), _p(mean), mean.numel(), _p(std), std.numel(), _p(wsp), ws_bytes, _stream(lib, src))
    return dst


def frame_diff(x, new_length, channels=3):
    """RGBDiff input (SSN._get_diff): x [..., (new_length + 1) * channels * k, H, W] stacked frames per segment ->
    [n_segments, new_length * channels, H, W] differences of consecutive frames."""
    lib = _check(x)
    h, w = x.shape[-2:]
    per_in = (new_length + 1) * channels
    assert x.numel() % (per_in * h * w) == 0
    n_seg = x.numel() // (per_in * h * w)
    out = torch.empty((n_seg, new_length * channels, h, w), device=x.device, dtype=torch.float32)
    lib.call("ssn_frame_diff", _p(x), _p(out), n_seg, int(new_length), int(channels), h * w, _stream(lib, x))
    return out


def detections(act, comp, reg, rel_prop, top_k, include_bg, nms_thresh, regress):
    """One video: -> (combined [P, C] fp32, dets [C, max_det, 5] fp64, counts [C] int32).  See ssn_detections
    (include_bg: 0 softmax over the class scores, 1 / True over all C + 1 scores, 2 no softmax)."""
    lib = _check(act, comp, reg, rel_prop)
    p, c = comp.shape
    assert act.shape == (p, c + 1) and rel_prop.shape == (p, 2) and rel_prop.dtype == torch.float64
    dev = act.device
    max_det = max(1, p)
    combined = torch.empty((p, c), device=dev, dtype=torch.float32)
    dets = torch.zeros((c, max_det, 5), device=dev, dtype=torch.float64)
    counts = torch.zeros(c, device=dev, dtype=torch.int32)
    ws_bytes = int(lib.cdll.ssn_detections_workspace_bytes(p, c))
    ws = torch.zeros((ws_bytes + 3) // 4, device=dev, dtype=torch.int32)
    lib.call("ssn_detections", _p(act), _p(comp), _p(reg), _p(rel_prop), _p(combined), _p(dets), _p(counts), _p(ws),
             ws_bytes, p, c, max_det, int(top_k), int(include_bg), float(nms_thresh), int(bool(regress)),
             _stream(lib, act))
    return combined, dets, counts


@_hbm_timed
def linear_fwd(x, w, b, out):
    lib = _check(x, w, b, out)
    lib.call("ssn_linear_fwd", _p(x), _p(w), _p(b), _p(out), x.shape[0], w.shape[0], w.shape[1], _stream(lib, x))


@_hbm_timed
def linear_bwd(dout, x, w, dx, dw, db, accumulate_dx=False):
    lib = _check(dout, x, w, dx, dw, db)
    lib.call("ssn_linear_bwd", _p(dout), _p(x), _p(w), _p(dx), _p(dw), _p(db), x.shape[0], w.shape[0], w.shape[1],
             int(accumulate_dx), _stream(lib, x))


@_hbm_timed
def row_gather(src, index, dst):
    lib = _check(src, index, dst)
    width = src[0].numel() if src.shape[0] else 0
    lib.call("ssn_row_gather", _p(src), _p(index), _p(dst), index.numel(), width, _stream(lib, src))


@_hbm_timed
def row_scatter(src, index, dst):
    lib = _check(src, index, dst)
    width = dst[0].numel()
    lib.call("ssn_row_scatter", _p(src), _p(index), _p(dst), index.numel(), dst.shape[0], width, _stream(lib, dst))


@_hbm_timed
def ce_loss_fwd(logits, target, loss, workspace):
    lib = _check(logits, target, loss, workspace)
    lib.call("ssn_ce_loss_fwd", _p(logits), _p(target), _p(loss), _p(workspace), logits.shape[0], logits.shape[1],
             _stream(lib, logits))


@_hbm_timed
def ce_loss_bwd(logits, target, lse, gout, dlogits):
    lib = _check(logits, target, lse, gout, dlogits)
    lib.call("ssn_ce_loss_bwd", _p(logits), _p(target), _p(lse), _p(gout), _p(dlogits), logits.shape[0],
             logits.shape[1], _stream(lib, logits))


@_hbm_timed
def completeness_loss_fwd(pred, labels, loss, coef, workspace, group, split, keep_pos, keep_neg, den):
    lib = _check(pred, labels, loss, coef, workspace)
    lib.call("ssn_completeness_loss_fwd", _p(pred), _p(labels), _p(loss), _p(coef), _p(workspace), pred.shape[0],
             pred.shape[1], group, split, keep_pos, keep_neg, float(den), _stream(lib, pred))


@_hbm_timed
def completeness_loss_bwd(labels, coef, gout, dpred, den):
    lib = _check(labels, coef, gout, dpred)
    lib.call("ssn_completeness_loss_bwd", _p(labels), _p(coef), _p(gout), _p(dpred), dpred.shape[0], dpred.shape[1],
             float(den), _stream(lib, dpred))


@_hbm_timed
def cw_smoothl1_fwd(pred, labels, targets, loss, diff):
    lib = _check(pred, labels, targets, loss, diff)
    lib.call("ssn_cw_smoothl1_fwd", _p(pred), _p(labels), _p(targets), _p(loss), _p(diff), pred.shape[0],
             pred.shape[1], _stream(lib, pred))


@_hbm_timed
def cw_smoothl1_bwd(labels, diff, gout, dpred):
    lib = _check(labels, diff, gout, dpred)
    lib.call("ssn_cw_smoothl1_bwd", _p(labels), _p(diff), _p(gout), _p(dpred), dpred.shape[0], dpred.shape[1],
             _stream(lib, dpred))


@_hbm_timed
def total_loss_fwd(act, act_t, comp, comp_t, reg, reg_lbl, reg_t, group, split, keep_pos, keep_neg, den, w_comp, w_reg, losses, lse, coef,
                   diff, scratch):
    """ssn_total_loss_fwd: losses[4] = activity CE, completeness, regression, act + w_comp * comp + w_reg * reg (reg may be None)."""
    lib = _check(act, act_t, comp, comp_t, reg, reg_lbl, reg_t, losses, lse, coef, diff, scratch)
    has = reg is not None
    lib.call("ssn_total_loss_fwd", _p(act), _p(act_t), act.shape[0], act.shape[1], _p(comp), _p(comp_t), comp.shape[0], comp.shape[1],
             group, split, keep_pos, keep_neg, float(den), _p(reg), _p(reg_lbl) if has else None, _p(reg_t) if has else None,
             reg.shape[0] if has else 0, reg.shape[1] if has else 0, float(w_comp), float(w_reg), _p(losses), _p(lse), _p(coef),
             _p(diff), _p(scratch), _stream(lib, act))


@_hbm_timed
def total_loss_bwd(act, act_t, comp_t, comp_shape, reg_lbl, reg_shape, den, w_comp, w_reg, lse, coef, diff, gout, d_act, d_comp, d_reg):
    lib = _check(act, act_t, comp_t, reg_lbl, lse, coef, diff, gout, d_act, d_comp, d_reg)
    has = d_reg is not None
    lib.call("ssn_total_loss_bwd", _p(act), _p(act_t), act.shape[0], act.shape[1], _p(comp_t), comp_shape[0], comp_shape[1], float(den),
             _p(reg_lbl) if has else None, reg_shape[0] if has else 0, reg_shape[1] if has else 0, float(w_comp), float(w_reg), _p(lse),
             _p(coef), _p(diff), _p(gout), _p(d_act), _p(d_comp), _p(d_reg), _stream(lib, act))


def label_select(target, reg_target, idx, outs, out_reg):
    """target[idx[k]] -> outs[k] (k = 0, 1, and 2 with reg_target[idx[2]] -> out_reg when idx[2] is not None): one launch."""
    lib = _check(target, reg_target, *[t for t in list(idx) + list(outs) + [out_reg] if t is not None])
    n = [0 if i is None else i.numel() for i in idx]
    lib.call("ssn_label_select", _p(target), _p(reg_target), _p(idx[0]), n[0], _p(idx[1]), n[1], _p(idx[2]), n[2], _p(outs[0]), _p(outs[1]),
             _p(outs[2]), _p(out_reg), _stream(lib, target))


class ParamChecksum(object):
    """Device-side fingerprint of a set of tensors (csrc/elementwise.hip: ssn_param_checksum): recorded once, checked later WITHOUT a
    host read -- a mismatch raises `bit` in a device flag word (the planes path's fault word: the pass that used the stale derivative is
    then repeated by the range guard's poll, see planes_exec.run_forward)."""

    def __init__(self, tensors, device):
        import numpy as np
        self.tensors = [t for t in tensors]      # (kept alive: the table holds raw pointers)
        tab = np.zeros((len(self.tensors), 2), dtype=np.int64)
        for i, t in enumerate(self.tensors):
            assert t.is_contiguous() and t.element_size() == 4, "checksummed tensors are contiguous 32-bit"
            tab[i] = (t.data_ptr(), t.numel())
        self.table = torch.from_numpy(tab).to(device)
        self.expected = torch.zeros(1, device=device, dtype=torch.int64)
        self.fresh = torch.zeros(1, device=device, dtype=torch.int64)
        self.ptrs = tuple(int(v) for v in tab[:, 0])
        lib = _lib.get_lib()
        lib.call("ssn_param_checksum", _p(self.table), len(self.tensors), _p(self.expected), None, None, 0, _stream(lib, self.table))

    def same_storage(self, tensors):
        return self.ptrs == tuple(t.data_ptr() for t in tensors)

    def check(self, flag, bit):
        lib = _lib.get_lib()
        lib.call("ssn_param_checksum", _p(self.table), len(self.tensors), _p(self.fresh), _p(self.expected), _p(flag), int(bit),
                 _stream(lib, self.table))


# ------------------------------------------------------------------------------------ optimiser
# Counts the in-place parameter updates issued through this module (they go through raw pointers: torch's version counters do not
# see them).  planes_exec keys its inference cache of packed weights on it.
PARAM_EPOCH = 0


def sgd_step(w, grad, buf, lr, momentum, weight_decay, grad_scale=1.0, first_step=False, skip_flag=None):
    global PARAM_EPOCH
    PARAM_EPOCH += 1
    lib = _check(w, grad, buf)
    lib.call("ssn_sgd_step", _p(w), _p(grad), _p(buf), w.numel(), float(lr), float(momentum), float(weight_decay),
             float(grad_scale), int(first_step), _p(skip_flag), _stream(lib, w))


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def sgd_step_multi(ws, grads, bufs, lrs, wds, momentum, grad_scale=1.0, first_step=False, skip_flag=None):
    """One fused launch (per 48 tensors) of the SGD update for many parameter tensors.  skip_flag: device int32 tensor; the
    update is skipped while its first word is non-zero (the range guard of the planes path, planes_exec.PlanesState)."""
    if not ws:
        return
    global PARAM_EPOCH
    PARAM_EPOCH += 1
    lib = _check(*ws, *grads, *bufs)
    n = len(ws)
    sizes = (ctypes.c_long * n)(*[w.numel() for w in ws])
    lr = (ctypes.c_float * n)(*[float(v) for v in lrs])
    wd = (ctypes.c_float * n)(*[float(v) for v in wds])
    pw, pg, pb = _ptr_array(ws), _ptr_array(grads), _ptr_array(bufs)   # keep the arrays alive across the call
    lib.call("ssn_sgd_step_multi", n, ctypes.addressof(pw), ctypes.addressof(pg), ctypes.addressof(pb),
             ctypes.addressof(sizes), ctypes.addressof(lr), ctypes.addressof(wd), float(momentum), float(grad_scale),
             int(first_step), _p(skip_flag), _stream(lib, ws[0]))


def wgrad_reduce(part, dw, db, splits, taps=1):
    """dw [M, K] (any trailing shape), db [M] or None <- sum over the `splits` partial slabs part [splits, M, K + 1] in a fixed
    order (column K = bias).  taps > 1: the slabs' columns are tap-major (column t * (K / taps) + ci holds dW[m][ci][t]), the
    layout of the nine-tap planes weight gradient."""
    lib = _check(part, dw, db)
    m = dw.shape[0]
    k = dw.numel() // m
    assert part.numel() >= splits * m * (k + 1)
    lib.call("ssn_wgrad_reduce_taps", _p(part), _p(dw), _p(db), m, k, int(splits), int(taps), _stream(lib, part))


def bn_fold_multi(biases, gammas, betas, means, variances, eps, scales, shifts):
    """ssn_bn_fold for many layers in one launch per 48 layers."""
    if not gammas:
        return
    lib = _check(*gammas, *scales, *shifts)
    n = len(gammas)
    e = (ctypes.c_float * n)(*[float(v) for v in eps])
    c = (ctypes.c_int * n)(*[g.numel() for g in gammas])
    arrs = [_ptr_array(x) for x in (biases, gammas, betas, means, variances, scales, shifts)]   # kept alive
    lib.call("ssn_bn_fold_multi", n, ctypes.addressof(arrs[0]), ctypes.addressof(arrs[1]), ctypes.addressof(arrs[2]),
             ctypes.addressof(arrs[3]), ctypes.addressof(arrs[4]), ctypes.addressof(e), ctypes.addressof(arrs[5]),
             ctypes.addressof(arrs[6]), ctypes.addressof(c), _stream(lib, gammas[0]))


def sumsq(x, out, accumulate, workspace):
    lib = _check(x, out, workspace)
    lib.call("ssn_sumsq", _p(x), x.numel(), _p(out), int(accumulate), _p(workspace), _stream(lib, x))


def embed_planes(g, out):
    """out (ChanSlice / tensor [N, C, H, W]) <- g (ChanSlice [N, C, Ho, Wo]) in the top-left corner of every plane, zero elsewhere;
    out shares g's amax slot (same values)."""
    lib = _check(g, out)
    out_s = out if isinstance(out, ChanSlice) else full(out)
    ho, wo = g.hw
    h, w = out_s.hw
    lib.call("ssn_embed_planes", _p(g), _p(out_s), g.n, g.c, ho, wo, g.img_stride, h, w, out_s.img_stride, _stream(lib, _tensor_of(out_s)))
    attach_amax(_tensor_of(out_s), _amax_in(g))
    return out


def add_(dst, src):
    """dst += src (flat fp32 buffers of equal length)."""
    lib = _check(dst, src)
    assert dst.numel() == src.numel()
    lib.call("ssn_add_inplace", _p(dst), _p(src), dst.numel(), _stream(lib, dst))


def scale_(x, coef_dev=None, coef=1.0):
    lib = _check(x, coef_dev)
    lib.call("ssn_scale", _p(x), x.numel(), _p(coef_dev), float(coef), _stream(lib, x))
