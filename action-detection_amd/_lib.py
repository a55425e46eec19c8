"""ctypes binding of libssn_hip.so (C ABI declared in include/ssn_hip.h).

The library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``) and loaded
lazily.  There is NO CPU fallback: if the shared object is missing or a tensor is not on a
HIP device the ops raise.  (The CPU-only test tier drives the same C ABI through a host
emulation build of the very same kernel sources, ``tests/emu``; it has to be installed
explicitly with :func:`use_library_for_testing` and is never picked up by the product path.)
"""
import ctypes
import os
import subprocess
import threading

# torch first: it bundles its own HIP runtime, and libssn_hip.so must bind to THAT copy (same SONAME).  If this
# library were dlopen-ed before torch, /opt/rocm's runtime would be loaded first and the process would end up with
# two HIP runtimes ("no ROCm-capable device is detected" at the first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libssn_hip.so")
SOURCES = ["conv_igemm.hip", "conv_x6.hip", "conv_x6_rect.hip", "conv_wgrad.hip", "conv_wgrad_x6.hip", "elementwise.hip", "bn_train.hip", "frames.hip", "detect.hip", "pool.hip", "stpp.hip", "heads_losses.hip", "conv_pl.hip", "planes_ops.hip", "planes_bn.hip", "wgrad_pl.hip"]

STPP_MAX_PARTS = 24


class SsnStppTable(ctypes.Structure):
    _fields_ = [("n_parts", ctypes.c_int), ("n_seg", ctypes.c_int),
                ("act_lo", ctypes.c_int), ("act_hi", ctypes.c_int),
                ("lo", ctypes.c_int * STPP_MAX_PARTS), ("hi", ctypes.c_int * STPP_MAX_PARTS),
                ("norm", ctypes.c_int * STPP_MAX_PARTS), ("col", ctypes.c_int * STPP_MAX_PARTS)]


# signature codes: p pointer, i int, l long, f float, u unsigned long long
_SIGS = {
    "ssn_conv_bn_relu_fwd": "pppppiiiiliiiliiiiipp",
    "ssn_bn_fold": "pppppfppip",
    "ssn_relu_bn_bwd": "pppiiillpp",
    "ssn_conv_dgrad": "pppiiiiliiiliiiiplpiipp",
    "ssn_conv_pack_weights": "ppiiiip",
    "ssn_conv_pack_weights_multi": "ippppppppp",
    "ssn_conv_wgrad": "ppppiiiiliiiliiiplip",
    "ssn_conv_wgrad_x6": "ppppiiiililiiiplippiipp",
    "ssn_wgrad_reduce": "pppiiip",
    "ssn_wgrad_reduce_taps": "pppiiiip",
    "ssn_conv_x6_pack_weights_multi": "ippppppppppppp",
    "ssn_conv_x6_fwd": "pppppiiiiliiiliiiiiippiiipp",
    "ssn_conv_x6_dgrad": "pppiiiiliiiliiiplpiippiipp",
    "ssn_conv_x6_fwd_rect": "pppppiiiiliiiliiiiiiippp",
    "ssn_conv_x6_pack_dgrad_s2": "ppiip",
    "ssn_conv_x6_dgrad_s2": "pppiiiiliiiliplpiippip",
    "ssn_conv_x6_pack_dgrad_rect": "ppiiiip",
    "ssn_conv_x6_pack_rect_multi": "ipppppppp",
    "ssn_conv_x6_dgrad_rect": "pppiiiililiiiiiplpiippp",
    "ssn_conv_wgrad_x6_rect": "ppppiiiililiiiiiplippp",
    "ssn_conv_x6_pack_weights_rect": "ppiiiip",
    "ssn_pool_fwd": "ipppiiiiliiliiipp",
    "ssn_pool_bwd": "ipppiiiiliiliiiiplppp",
    "ssn_avgpool_affine_fwd": "ppppiiiiiliiliiipp",
    "ssn_channel_sum": "ppiiilpup",
    "ssn_tensor_amax": "plpp",
    "ssn_bn_train_stats": "ppppppiiilffpup",
    "ssn_bn_train_apply": "ppppppiiiillpp",
    "ssn_bn_train_bwd": "pppppppppiiiillllpupp",
    "ssn_global_avgpool_fwd": "ppiiilp",
    "ssn_global_avgpool_bwd": "ppiiilipp",
    "ssn_dropout_fwd": "ppplfupp",
    "ssn_dropout_bwd": "ppplfp",
    "ssn_stpp_fwd": "ppppiipp",
    "ssn_stpp_bwd": "ppppiipp",
    "ssn_heads_fwd": "pppppppppppiipp",
    "ssn_heads_bwd": "pppppppppppiippppp",
    "ssn_stpp_reorg": "piippppiiiiipppp",
    "ssn_crop_mean": "ppiiip",
    "ssn_detections": "ppppppppuiiiiidip",
    "ssn_frames_crop_normalize": "ppiiiiiiipppiipipip",
    "ssn_frames_crop_resize_normalize": "ppiiiiiippiipipipup",
    "ssn_reg_denorm": "plffffp",
    "ssn_frame_diff": "ppliiip",
    "ssn_linear_fwd": "ppppiiip",
    "ssn_linear_bwd": "ppppppiiiip",
    "ssn_row_gather": "pppiip",
    "ssn_row_scatter": "pppiiip",
    "ssn_ce_loss_fwd": "ppppiip",
    "ssn_ce_loss_bwd": "pppppiip",
    "ssn_completeness_loss_fwd": "pppppiiiiiifp",
    "ssn_completeness_loss_bwd": "ppppiifp",
    "ssn_cw_smoothl1_fwd": "pppppiip",
    "ssn_total_loss_fwd": "ppiippiiiiiifpppiiffpppppp",
    "ssn_total_loss_bwd": "ppiipiifpiiffpppppppp",
    "ssn_label_select": "pppipipippppp",
    "ssn_param_checksum": "pipppip",
    "ssn_cw_smoothl1_bwd": "ppppiip",
    "ssn_sgd_step": "ppplffffipp",
    "ssn_sgd_step_multi": "ippppppffipp",
    "ssn_bn_fold_multi": "ipppppppppp",
    "ssn_sumsq": "plpipp",
    "ssn_scale": "plpfp",
    "ssn_add_inplace": "pplp",
    "ssn_embed_planes": "ppiiiiliilp",
    "ssn_space_to_depth2": "ppiiiipp",
    "ssn_s2d_weights": "ppiiip",
    "ssn_s2d_weights_bwd": "ppiiip",
    # planes tensors (csrc/planes.h): hi/lo f16 planes, NC8HW8
    "ssn_pl_scales_update": "pppiiiip",
    "ssn_pl_range_check": "pppiip",
    "ssn_pl_from_f32": "plppiiiilippp",
    "ssn_pl_to_f32": "pplpliiipp",
    "ssn_pl_im2col": "pplppliiiiiiiiiiip",
    "ssn_conv_pl_fwd": "pppppppiiiiliiiliiiiiiipppiiip",
    "ssn_conv_pl_dgrad": "pppppiiiiliiiliiiiiplpipppiiip",
    "ssn_conv_wgrad_pl": "ppppppiiiiliiiliiiiiplippiipp",
    "ssn_wgrad_reduce_multi": "ipppppppp",
    "ssn_conv_wgrad_pl_group": "ippppppppppplplp",
    "ssn_conv_pl_dgrad_s2": "pppppiiiiliiiliiplpipppp",
    "ssn_pl_maxpool_fwd": "pplpplpiiiiiiiiipppp",
    "ssn_pl_maxpool_bwd": "pplpppl" + "i" * 10 + "plpipppplp",
    "ssn_pl_avgpool_affine": "pplpplppiiiiiiipppp",
    "ssn_pl_relu_bn_bwd": "pplplpiiippp",
    "ssn_pl_gap_fwd": "pplpiiipp",
    "ssn_pl_gap_bwd": "pppliiiplpppp",
    "ssn_pl_channel_sum": "pplpiiipplp",
    "ssn_pl_channel_sum_multi": "ippppipppplp",
    "ssn_conv_x6_pack_batch_begin": "",
    "ssn_conv_x6_pack_batch_end": "plip",
    "ssn_pl_bn_train_stats": "pplpppppp" + "iiiffplp",
    "ssn_pl_bn_train_apply": "pplppplpp" + "ppppiiiip",
    "ssn_pl_bn_train_bwd": "pplpplpplp" + "ppppp" + "pplpp" + "pliiiplp",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "d": ctypes.c_double,
       "u": ctypes.c_ulonglong}

EXPORTS = sorted(list(_SIGS) + ["ssn_last_error", "ssn_abi_version", "ssn_conv_wgrad_workspace_bytes",
                                "ssn_conv_pick_tile", "ssn_conv_packed_floats", "ssn_conv_x6_packed_floats", "ssn_conv_x6_packed_floats_rect", "ssn_conv_x6_packed_floats_dgrad_rect", "ssn_conv_wgrad_x6_rect_workspace_bytes", "ssn_conv_x6_dgrad_s2_packed_floats", "ssn_conv_x6_debug_flags", "ssn_conv_x6_debug_trace",
                                "ssn_conv_wgrad_x6_workspace_bytes", "ssn_detections_workspace_bytes",
                                "ssn_conv_debug_flags", "ssn_channel_sum_shares", "ssn_bn_train_workspace_floats",
                                "ssn_conv_dgrad_layout", "ssn_conv_pl_tiles", "ssn_conv_pl_halo_taken", "ssn_conv_pl_debug_flags", "ssn_conv_pl_debug_trace", "ssn_conv_wgrad_pl_debug_trace", "ssn_conv_wgrad_pl_debug_flags", "ssn_conv_pl_tile_shape", "ssn_conv_wgrad_pl_tiles", "ssn_conv_wgrad_pl_workspace_bytes", "ssn_conv_wgrad_pl_group_workspace_bytes", "ssn_conv_wgrad_pl_group_table_bytes", "ssn_conv_wgrad_pl_group_tuning", "ssn_pl_channel_sum_workspace_bytes", "ssn_pl_bn_train_workspace_bytes", "ssn_conv_x6_pack_batch_entries", "ssn_conv_x6_pack_entry_bytes", "ssn_conv_x6_pack_batch_abort", "ssn_frames_resize_workspace_bytes"])


class SsnLibrary:
    """A loaded libssn_hip.so (or, in tests only, the host-emulation build of the same ABI)."""

    def __init__(self, path, is_emulator=False):
        self.path = path
        self.is_emulator = is_emulator
        self.cdll = ctypes.CDLL(path)
        self.cdll.ssn_last_error.restype = ctypes.c_char_p
        self.cdll.ssn_last_error.argtypes = []
        self.cdll.ssn_abi_version.restype = ctypes.c_int
        self.cdll.ssn_conv_wgrad_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 7
        self.cdll.ssn_conv_packed_floats.restype = ctypes.c_long
        self.cdll.ssn_conv_packed_floats.argtypes = [ctypes.c_int] * 4
        self.cdll.ssn_conv_x6_packed_floats.restype = ctypes.c_long
        self.cdll.ssn_conv_x6_packed_floats.argtypes = [ctypes.c_int] * 4
        self.cdll.ssn_conv_x6_dgrad_s2_packed_floats.restype = ctypes.c_long
        self.cdll.ssn_conv_x6_dgrad_s2_packed_floats.argtypes = [ctypes.c_int] * 2
        self.cdll.ssn_conv_x6_packed_floats_rect.restype = ctypes.c_long
        self.cdll.ssn_conv_x6_packed_floats_rect.argtypes = [ctypes.c_int] * 4
        self.cdll.ssn_conv_x6_packed_floats_dgrad_rect.restype = ctypes.c_long
        self.cdll.ssn_conv_x6_packed_floats_dgrad_rect.argtypes = [ctypes.c_int] * 4
        self.cdll.ssn_conv_wgrad_x6_rect_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_x6_rect_workspace_bytes.argtypes = [ctypes.c_int] * 8
        self.cdll.ssn_conv_wgrad_x6_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_x6_workspace_bytes.argtypes = [ctypes.c_int] * 7
        self.cdll.ssn_detections_workspace_bytes.restype = ctypes.c_size_t
        self.cdll.ssn_detections_workspace_bytes.argtypes = [ctypes.c_int] * 2
        self.cdll.ssn_conv_pick_tile.restype = ctypes.c_int
        self.cdll.ssn_conv_pick_tile.argtypes = [ctypes.c_int, ctypes.c_long]
        self.cdll.ssn_pl_channel_sum_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_pl_channel_sum_workspace_bytes.argtypes = [ctypes.c_int]
        self.cdll.ssn_conv_x6_pack_entry_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_x6_pack_batch_abort.restype = None
        self.cdll.ssn_pl_bn_train_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_pl_bn_train_workspace_bytes.argtypes = [ctypes.c_int]
        self.cdll.ssn_frames_resize_workspace_bytes.restype = ctypes.c_size_t
        self.cdll.ssn_frames_resize_workspace_bytes.argtypes = [ctypes.c_int] * 3
        self.cdll.ssn_conv_wgrad_pl_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_pl_workspace_bytes.argtypes = [ctypes.c_int] * 8
        self.cdll.ssn_conv_wgrad_pl_group_workspace_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_pl_group_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.cdll.ssn_conv_wgrad_pl_group_table_bytes.restype = ctypes.c_long
        self.cdll.ssn_conv_wgrad_pl_group_table_bytes.argtypes = [ctypes.c_int]
        self.cdll.ssn_conv_wgrad_pl_group_tuning.restype = None
        self.cdll.ssn_conv_wgrad_pl_group_tuning.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        self._fn = {}
        for name, sig in _SIGS.items():
            fn = getattr(self.cdll, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[c] for c in sig]
            self._fn[name] = fn

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, self.cdll.ssn_last_error().decode()))

    def wgrad_workspace_bytes(self, n, cin, cout, ho, wo, ksize, tile_cfg=-1):
        return int(self.cdll.ssn_conv_wgrad_workspace_bytes(n, cin, cout, ho, wo, ksize, tile_cfg))


_lock = threading.Lock()
_lib = None
_test_lib = None


def _source_stamp(deps, flags):
    """sha256 over the compile flags and the contents of every source / header the library is built from."""
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into the in-tree libssn_hip.so (cross-compiles on CPU).

    Up-to-date check by CONTENT (a stamp file next to the library holds the hash of the sources it was built from):
    file times do not survive being copied to another box.  A file lock serialises concurrent builders (the ranks of
    a multi-GPU launch all call this)."""
    import fcntl
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("ssn_common.h", "conv_epilogue.h", "conv_x6_kernel.h", "planes.h", "conv_pl_epilogue.inc")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    stamp_path = LIB_PATH + ".stamp"
    want = _source_stamp(deps, flags)

    def fresh():
        if not os.path.exists(LIB_PATH) or not os.path.exists(stamp_path):
            return False
        with open(stamp_path) as f:
            return f.read().strip() == want

    if not force and fresh():
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():      # another process built it while this one waited
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            objs = []
            procs = []
            headers = deps[len(srcs):]
            for s in srcs:
                o = os.path.join(CSRC, os.path.basename(s) + ".o")
                objs.append(o)
                # per-object stamp (source + every header + flags): an edit to one translation unit recompiles that one only
                ostamp = _source_stamp([s] + headers, flags)
                try:
                    with open(o + ".stamp") as f:
                        if not force and os.path.exists(o) and f.read().strip() == ostamp:
                            continue
                except OSError:
                    pass
                cmd = [hipcc] + flags + ["-c", s, "-o", o]
                procs.append((cmd, o, ostamp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            for cmd, o, ostamp, p in procs:
                out, _ = p.communicate()
                if verbose and out:
                    print(out.decode())
                if p.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
                with open(o + ".stamp", "w") as f:
                    f.write(ostamp + "\n")
            tmp = LIB_PATH + ".tmp.%d" % os.getpid()
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
            os.replace(tmp, LIB_PATH)      # readers never see a half-written library
            with open(stamp_path, "w") as f:
                f.write(want + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def get_lib():
    """The product library.  Raises if it has not been built -- never falls back to anything else."""
    global _lib
    if _test_lib is not None:
        return _test_lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "libssn_hip.so is missing (%s). Build it with __graft_entry__.build() / "
                        "action_detection_amd.build(); there is no CPU fallback." % LIB_PATH)
                _lib = SsnLibrary(LIB_PATH)
    return _lib


def use_library_for_testing(lib):
    """Install (or with None remove) an alternative ABI implementation.  Tests only."""
    global _test_lib
    _test_lib = lib


def emulator_active():
    return _test_lib is not None and _test_lib.is_emulator
