"""Inception-v3 layer manifest owned by this repo (BASELINE.json configs[4]: dense testing on ActivityNet).

The reference takes this backbone from the same un-vendored ``model_zoo`` submodule as BN-Inception
(/root/reference/ssn_models.py:133-139: ``getattr(model_zoo, 'InceptionV3')()``, last layer ``top_cls_fc`` with
2048 inputs, 299x299 crops, BGR mean [104, 117, 128], std 1).  The upstream YAML manifest is not in the reference
tree and cannot be fetched, so the topology is the published Inception-v3 (Szegedy et al. 2016, the layout every
public implementation shares): **unpinned against upstream** (layer ids, bias/eps conventions and the pool type of
the last block could differ from yjxiong/tensorflow-model-zoo.torch); everything that depends on those choices
lives in this one file.  Conventions follow ``bninception_spec``: every conv has a bias and is followed by
``BatchNorm2d(eps=1e-5)`` + ReLU, average pools count the padding.

Op tuples
---------
("conv", id, src, dst, dst_c0, cin, cout, kh, kw, stride, pad_h, pad_w)
("pool", id, kind, src, dst, dst_c0, k, stride, pad)       kind in {"max", "avg"}
("gap", id, src, dst)
"""

FEATURE_DIM = 2048


def build_manifest(in_channels=3, input_size=299):
    """Return (ops, tensors) where tensors maps name -> (C, H, W)."""
    ops, tensors = [], {"data": (in_channels, input_size, input_size)}

    def out_size(h, k, s, p):
        return (h + 2 * p - k) // s + 1

    def conv(id_, src, dst, c0, cout, kh, kw, s=1, ph=0, pw=0):
        cin, h, w = tensors[src]
        ho, wo = out_size(h, kh, s, ph), out_size(w, kw, s, pw)
        if dst not in tensors:
            tensors[dst] = (cout, ho, wo)
        assert tensors[dst][1:] == (ho, wo), (id_, tensors[dst], ho, wo)
        ops.append(("conv", id_, src, dst, c0, cin, cout, kh, kw, s, ph, pw))

    def pool(id_, kind, src, dst, c0, k, s, p):
        c, h, w = tensors[src]
        ho = out_size(h, k, s, p)
        if dst not in tensors:
            tensors[dst] = (c, ho, ho)
        assert tensors[dst][1:] == (ho, ho), (id_, tensors[dst], ho)
        ops.append(("pool", id_, kind, src, dst, c0, k, s, p))

    def chain(pre, src, steps, dst, c0):
        """steps: [(suffix, cout, kh, kw, stride, ph, pw)]; the last one writes channels [c0, ...) of dst."""
        cur = src
        for i, (suf, cout, kh, kw, s, ph, pw) in enumerate(steps):
            last = i == len(steps) - 1
            t = dst if last else pre + suf
            conv(pre + suf, cur, t, c0 if last else 0, cout, kh, kw, s, ph, pw)
            cur = t
        return steps[-1][1]

    # ---- stem: 299 -> 35
    conv("conv_1a_3x3", "data", "conv_1a", 0, 32, 3, 3, 2)
    conv("conv_2a_3x3", "conv_1a", "conv_2a", 0, 32, 3, 3)
    conv("conv_2b_3x3", "conv_2a", "conv_2b", 0, 64, 3, 3, 1, 1, 1)
    pool("pool_3a_3x3", "max", "conv_2b", "pool_3a", 0, 3, 2, 0)
    conv("conv_3b_1x1", "pool_3a", "conv_3b", 0, 80, 1, 1)
    conv("conv_4a_3x3", "conv_3b", "conv_4a", 0, 192, 3, 3)
    pool("pool_5a_3x3", "max", "conv_4a", "pool_5a", 0, 3, 2, 0)
    cur = "pool_5a"

    def block_out(name, cout):
        c, h, w = tensors[cur]
        tensors[name] = (cout, h, w)
        return name

    # ---- 3 x block A (35 x 35)
    for name, pf in (("mixed_5b", 32), ("mixed_5c", 64), ("mixed_5d", 64)):
        pre = name + "_"
        out = block_out(pre + "output", 64 + 64 + 96 + pf)
        c0 = 0
        c0 += chain(pre, cur, [("1x1", 64, 1, 1, 1, 0, 0)], out, c0)
        c0 += chain(pre, cur, [("5x5_reduce", 48, 1, 1, 1, 0, 0), ("5x5", 64, 5, 5, 1, 2, 2)], out, c0)
        c0 += chain(pre, cur, [("double_3x3_reduce", 64, 1, 1, 1, 0, 0), ("double_3x3_1", 96, 3, 3, 1, 1, 1),
                               ("double_3x3_2", 96, 3, 3, 1, 1, 1)], out, c0)
        pool(pre + "pool", "avg", cur, pre + "pool", 0, 3, 1, 1)
        c0 += chain(pre, pre + "pool", [("pool_proj", pf, 1, 1, 1, 0, 0)], out, c0)
        assert c0 == tensors[out][0]
        cur = out

    # ---- block B: 35 -> 17
    pre = "mixed_6a_"
    cin, h, _ = tensors[cur]
    ho = out_size(h, 3, 2, 0)
    out = pre + "output"
    tensors[out] = (384 + 96 + cin, ho, ho)
    c0 = 0
    c0 += chain(pre, cur, [("3x3", 384, 3, 3, 2, 0, 0)], out, c0)
    c0 += chain(pre, cur, [("double_3x3_reduce", 64, 1, 1, 1, 0, 0), ("double_3x3_1", 96, 3, 3, 1, 1, 1),
                           ("double_3x3_2", 96, 3, 3, 2, 0, 0)], out, c0)
    pool(pre + "pool", "max", cur, out, c0, 3, 2, 0)
    cur = out

    # ---- 4 x block C (17 x 17)
    for name, c7 in (("mixed_6b", 128), ("mixed_6c", 160), ("mixed_6d", 160), ("mixed_6e", 192)):
        pre = name + "_"
        out = block_out(pre + "output", 192 * 4)
        c0 = 0
        c0 += chain(pre, cur, [("1x1", 192, 1, 1, 1, 0, 0)], out, c0)
        c0 += chain(pre, cur, [("7x7_reduce", c7, 1, 1, 1, 0, 0), ("1x7", c7, 1, 7, 1, 0, 3),
                               ("7x1", 192, 7, 1, 1, 3, 0)], out, c0)
        c0 += chain(pre, cur, [("double_7x7_reduce", c7, 1, 1, 1, 0, 0), ("double_7x1_1", c7, 7, 1, 1, 3, 0),
                               ("double_1x7_1", c7, 1, 7, 1, 0, 3), ("double_7x1_2", c7, 7, 1, 1, 3, 0),
                               ("double_1x7_2", 192, 1, 7, 1, 0, 3)], out, c0)
        pool(pre + "pool", "avg", cur, pre + "pool", 0, 3, 1, 1)
        c0 += chain(pre, pre + "pool", [("pool_proj", 192, 1, 1, 1, 0, 0)], out, c0)
        assert c0 == tensors[out][0]
        cur = out

    # ---- block D: 17 -> 8
    pre = "mixed_7a_"
    cin, h, _ = tensors[cur]
    ho = out_size(h, 3, 2, 0)
    out = pre + "output"
    tensors[out] = (320 + 192 + cin, ho, ho)
    c0 = 0
    c0 += chain(pre, cur, [("3x3_reduce", 192, 1, 1, 1, 0, 0), ("3x3", 320, 3, 3, 2, 0, 0)], out, c0)
    c0 += chain(pre, cur, [("7x7x3_reduce", 192, 1, 1, 1, 0, 0), ("7x7x3_1x7", 192, 1, 7, 1, 0, 3),
                           ("7x7x3_7x1", 192, 7, 1, 1, 3, 0), ("7x7x3_3x3", 192, 3, 3, 2, 0, 0)], out, c0)
    pool(pre + "pool", "max", cur, out, c0, 3, 2, 0)
    cur = out

    # ---- 2 x block E (8 x 8)
    for name in ("mixed_7b", "mixed_7c"):
        pre = name + "_"
        out = block_out(pre + "output", 320 + 768 + 768 + 192)
        c0 = 0
        c0 += chain(pre, cur, [("1x1", 320, 1, 1, 1, 0, 0)], out, c0)
        conv(pre + "3x3_reduce", cur, pre + "3x3_reduce", 0, 384, 1, 1)
        conv(pre + "3x3_1x3", pre + "3x3_reduce", out, c0, 384, 1, 3, 1, 0, 1)
        conv(pre + "3x3_3x1", pre + "3x3_reduce", out, c0 + 384, 384, 3, 1, 1, 1, 0)
        c0 += 768
        conv(pre + "double_3x3_reduce", cur, pre + "double_3x3_reduce", 0, 448, 1, 1)
        conv(pre + "double_3x3_1", pre + "double_3x3_reduce", pre + "double_3x3_1", 0, 384, 3, 3, 1, 1, 1)
        conv(pre + "double_3x3_1x3", pre + "double_3x3_1", out, c0, 384, 1, 3, 1, 0, 1)
        conv(pre + "double_3x3_3x1", pre + "double_3x3_1", out, c0 + 384, 384, 3, 1, 1, 1, 0)
        c0 += 768
        pool(pre + "pool", "avg", cur, pre + "pool", 0, 3, 1, 1)
        c0 += chain(pre, pre + "pool", [("pool_proj", 192, 1, 1, 1, 0, 0)], out, c0)
        assert c0 == tensors[out][0]
        cur = out

    tensors["global_pool"] = (tensors[cur][0], 1, 1)
    ops.append(("gap", "global_pool", cur, "global_pool"))
    assert tensors[cur][0] == FEATURE_DIM
    return ops, tensors


def conv_macs(ops, tensors):
    """Direct-convolution MACs per image."""
    total = 0
    for op in ops:
        if op[0] == "conv":
            _, _, _, dst, _, cin, cout, kh, kw, _, _, _ = op
            total += tensors[dst][1] * tensors[dst][2] * cin * cout * kh * kw
    return total
