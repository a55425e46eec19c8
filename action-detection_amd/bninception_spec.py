"""BN-Inception layer manifest owned by this repo.

The reference imports its backbone from the un-vendored ``model_zoo`` submodule
(/root/reference/ssn_models.py:121-124, .gitmodules:1-4); the directory is empty, so
the topology is re-specified here.  Layer ids follow the
upstream Caffe-derived naming (``/`` -> ``_``) so ``state_dict`` keys line up with
the checkpoints the reference loads at /root/reference/ssn_train.py:42,49,56.

The manifest is a flat, ordered list of ops over named tensors, which is exactly
what the executor (``bninception.py``) walks; no Python control flow lives in the
topology itself.

Op tuples
---------
("conv", id, src, dst, dst_c0, cin, cout, k, stride, pad)
    conv ``id`` + BatchNorm ``id_bn`` + ReLU, reads all channels of tensor ``src``
    and writes channels [dst_c0, dst_c0+cout) of tensor ``dst`` (concat-free
    Inception outputs: each branch writes straight into its slice).
("pool", id, kind, src, dst, dst_c0, k, stride, pad, ceil)
    kind in {"max", "avg"}; avg uses count_include_pad=True (Caffe heritage).
("gap", id, src, dst)
    global average pool to [N, C].
"""

# block table: name, cin, 1x1, (3x3 reduce, 3x3), (dbl reduce, dbl_1, dbl_2), (pool kind, proj), stride
_BLOCKS = [
    ("3a", 192, 64, (64, 64), (64, 96, 96), ("avg", 32), 1),
    ("3b", 256, 64, (64, 96), (64, 96, 96), ("avg", 64), 1),
    ("3c", 320, 0, (128, 160), (64, 96, 96), ("max", 0), 2),
    ("4a", 576, 224, (64, 96), (96, 128, 128), ("avg", 128), 1),
    ("4b", 576, 192, (96, 128), (96, 128, 128), ("avg", 128), 1),
    ("4c", 576, 160, (128, 160), (128, 160, 160), ("avg", 128), 1),
    ("4d", 608, 96, (128, 192), (160, 192, 192), ("avg", 128), 1),
    ("4e", 608, 0, (128, 192), (192, 256, 256), ("max", 0), 2),
    ("5a", 1056, 352, (192, 320), (160, 224, 224), ("avg", 128), 1),
    ("5b", 1024, 352, (192, 320), (192, 224, 224), ("max", 128), 1),
]


def _pool_out(size, k, stride, pad, ceil):
    num = size + 2 * pad - k
    o = (-(-num // stride) if ceil else num // stride) + 1
    if ceil and (o - 1) * stride >= size + pad:
        o -= 1
    return o


def build_manifest(in_channels=3, input_size=224):
    """Return (ops, tensors) where tensors maps name -> (C, H, W)."""
    ops = []
    tensors = {"data": (in_channels, input_size, input_size)}

    def conv(id_, src, dst, c0, cin, cout, k, s, p):
        ops.append(("conv", id_, src, dst, c0, cin, cout, k, s, p))

    def pool(id_, kind, src, dst, c0, k, s, p, ceil):
        ops.append(("pool", id_, kind, src, dst, c0, k, s, p, ceil))

    h = (input_size + 2 * 3 - 7) // 2 + 1
    tensors["conv1"] = (64, h, h)
    conv("conv1_7x7_s2", "data", "conv1", 0, in_channels, 64, 7, 2, 3)
    h = _pool_out(h, 3, 2, 0, True)
    tensors["pool1"] = (64, h, h)
    pool("pool1_3x3_s2", "max", "conv1", "pool1", 0, 3, 2, 0, True)
    tensors["conv2_reduce"] = (64, h, h)
    conv("conv2_3x3_reduce", "pool1", "conv2_reduce", 0, 64, 64, 1, 1, 0)
    tensors["conv2"] = (192, h, h)
    conv("conv2_3x3", "conv2_reduce", "conv2", 0, 64, 192, 3, 1, 1)
    h = _pool_out(h, 3, 2, 0, True)
    tensors["pool2"] = (192, h, h)
    pool("pool2_3x3_s2", "max", "conv2", "pool2", 0, 3, 2, 0, True)

    cur = "pool2"
    for name, cin, c1, (r3, c3), (rd, cd1, cd2), (pkind, cproj), stride in _BLOCKS:
        assert tensors[cur][0] == cin, (name, tensors[cur], cin)
        pre = "inception_%s_" % name
        ho = h if stride == 1 else _pool_out(h, 3, 2, 0, True)
        cout = c1 + c3 + cd2 + (cproj if cproj else cin)
        out = pre + "output"
        tensors[out] = (cout, ho, ho)
        c0 = 0
        if c1:
            conv(pre + "1x1", cur, out, c0, cin, c1, 1, 1, 0)
            c0 += c1
        t = pre + "3x3_reduce"
        tensors[t] = (r3, h, h)
        conv(pre + "3x3_reduce", cur, t, 0, cin, r3, 1, 1, 0)
        conv(pre + "3x3", t, out, c0, r3, c3, 3, stride, 1)
        c0 += c3
        t = pre + "double_3x3_reduce"
        tensors[t] = (rd, h, h)
        conv(pre + "double_3x3_reduce", cur, t, 0, cin, rd, 1, 1, 0)
        t2 = pre + "double_3x3_1"
        tensors[t2] = (cd1, h, h)
        conv(pre + "double_3x3_1", t, t2, 0, rd, cd1, 3, 1, 1)
        conv(pre + "double_3x3_2", t2, out, c0, cd1, cd2, 3, stride, 1)
        c0 += cd2
        if cproj:
            t = pre + "pool"
            tensors[t] = (cin, h, h)
            pool(pre + "pool", pkind, cur, t, 0, 3, 1, 1, True)
            conv(pre + "pool_proj", t, out, c0, cin, cproj, 1, 1, 0)
            c0 += cproj
        else:
            # stride-2 blocks: the pooled input passes straight through into the concat
            pool(pre + "pool", pkind, cur, out, c0, 3, 2, 0, True)
            c0 += cin
        assert c0 == cout
        cur, h = out, ho

    tensors["global_pool"] = (tensors[cur][0], 1, 1)
    ops.append(("gap", "global_pool", cur, "global_pool"))
    return ops, tensors


def conv_macs(ops, tensors):
    """Direct-convolution MACs per image (the algorithmic work of SURVEY.md section 8d)."""
    total = 0
    for op in ops:
        if op[0] != "conv":
            continue
        _, _, src, dst, _, cin, cout, k, s, p = op
        ho = tensors[dst][1]
        total += ho * ho * cin * cout * k * k
    return total


FEATURE_DIM = 1024
