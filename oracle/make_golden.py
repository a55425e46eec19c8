#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own classes (test infrastructure).

Runs only in the build container, where /root/reference exists; the fixtures it writes are
committed so the GPU box (which has no /root/reference) can check against them.

The reference's hot-path classes import and run unmodified on torch 2.10 CPU with three shims
(SURVEY.md section 8c): a ``torchvision`` stub, a ``model_zoo`` stub whose ``BNInception`` is the
oracle's torch-CPU backbone (the real one is an un-vendored submodule), and a no-op
``Tensor.cuda``.  Every fixture stores the reference's outputs for seeded inputs; the tests then
hold the oracle restatement (oracle/ssn_oracle.py) -- and on the GPU the HIP path -- to them.

    python oracle/make_golden.py            # writes tests/golden/ref_*.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ssn_oracle as O  # noqa: E402
import action_detection_amd  # noqa: E402,F401
from action_detection_amd.synthetic import init_backbone_synthetic, init_heads_synthetic, make_batch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def install_shims():
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    for name in ("Compose", "CenterCrop", "Scale"):
        setattr(tv.transforms, name, type(name, (), {"__init__": lambda self, *a, **k: None}))
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.modules["torchvision.transforms"] = tv.transforms
    mz = types.ModuleType("model_zoo")
    mz.BNInception = lambda: O.OracleBNInception()
    sys.modules["model_zoo"] = mz
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def golden_stpp(ref_ops):
    rng = np.random.RandomState(11)
    out = {}
    cases = [((1, 1, 1), [2, 7, 9]), ((1, (1, 2), 1), [2, 7, 9]), ((2, (1, 2, 3), 1), [2, 7, 9]),
             ((1, (1, 2), (1, 2)), [1, 4, 6])]
    for ci, (cfg, split) in enumerate(cases):
        p, d = 5, 16
        ft = rng.standard_normal((p * split[2], d)).astype(np.float32)
        sc = rng.uniform(0, 1, (p, 2)).astype(np.float32)
        mod = ref_ops.StructuredTemporalPyramidPooling(d, True, configs=cfg)
        ft_t = torch.from_numpy(ft).requires_grad_()
        act, stpp = mod(ft_t, torch.from_numpy(sc), split)
        ga = rng.standard_normal(act.shape).astype(np.float32)
        gs = rng.standard_normal(stpp.shape).astype(np.float32)
        (act * torch.from_numpy(ga)).sum().add((stpp * torch.from_numpy(gs)).sum()).backward()
        out.update({"c%d_ft" % ci: ft, "c%d_sc" % ci: sc, "c%d_act" % ci: act.detach().numpy(),
                    "c%d_stpp" % ci: stpp.detach().numpy(), "c%d_ga" % ci: ga, "c%d_gs" % ci: gs,
                    "c%d_dft" % ci: ft_t.grad.numpy(), "c%d_split" % ci: np.array(split),
                    "c%d_cfg" % ci: np.array(repr(cfg))})
    # known-answer table of the integer tick truncation (ops/ssn_ops.py:53-55)
    kat = []
    for length in range(1, 10):
        for n_part in range(1, 5):
            t = torch.arange(0, length + 1e-5, length / n_part)
            kat.append([length, n_part] + [int(v) for v in t] + [-1] * (6 - len(t)))
    out["ticks_kat"] = np.array(kat, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_stpp.npz"), **out)


def golden_losses(ref_ops):
    rng = np.random.RandomState(12)
    out = {}
    for ci, v in enumerate((4, 16)):
        c = 20
        pred = (rng.standard_normal((7 * v, c)) * 1.5).astype(np.float32)
        labels = rng.randint(1, c + 1, size=7 * v).astype(np.int64)
        pt = torch.from_numpy(pred).requires_grad_()
        loss = ref_ops.CompletenessLoss()(pt, torch.from_numpy(labels), 1, 7)
        loss.backward()
        out.update({"comp%d_pred" % ci: pred, "comp%d_labels" % ci: labels,
                    "comp%d_loss" % ci: loss.detach().numpy(), "comp%d_grad" % ci: pt.grad.numpy()})
    n, c = 6, 20
    pred = (rng.standard_normal((n, c, 2)) * 1.2).astype(np.float32)
    labels = rng.randint(1, c + 1, size=n).astype(np.int64)
    tg = rng.standard_normal((n, 2)).astype(np.float32)
    pt = torch.from_numpy(pred).requires_grad_()
    loss = ref_ops.ClassWiseRegressionLoss()(pt, torch.from_numpy(labels), torch.from_numpy(tg))
    loss.backward()
    out.update({"reg_pred": pred, "reg_labels": labels, "reg_targets": tg, "reg_loss": loss.detach().numpy(),
                "reg_grad": pt.grad.numpy()})
    # OHEM hinge alone, ratio 0.5 on groups of 4 (exercises keep > 1)
    pred = (rng.standard_normal((16, 5)) * 1.5).astype(np.float32)
    labels = rng.randint(1, 6, size=16).astype(np.int64)
    pt = torch.from_numpy(pred).requires_grad_()
    loss = ref_ops.OHEMHingeLoss.apply(pt, torch.from_numpy(labels), -1, 0.5, 4)
    loss.backward()
    out.update({"ohem_pred": pred, "ohem_labels": labels, "ohem_loss": loss.detach().numpy(),
                "ohem_grad": pt.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, "ref_losses.npz"), **out)


def golden_reorg(ref_ops):
    rng = np.random.RandomState(13)
    out = {}
    for ci, cfg in enumerate(((1, 1, 1), (1, (1, 2), 1))):
        mult = sum(sum(c) if isinstance(c, tuple) else c for c in cfg)
        a, c, r = 21, 20, 40
        d = a + (c + r) * mult
        t = 40
        scores = rng.standard_normal((t, d)).astype(np.float32)
        starts = rng.randint(0, t - 2, size=12)
        ends = np.minimum(starts + rng.randint(1, 12, size=12), t)
        dur = ends - starts
        # ticks are clamped to [0, T] and non-decreasing by construction in the reference
        # (real_rel_starting / real_rel_ending, ssn_dataset.py:417-424)
        ticks = np.stack([np.maximum(starts - dur // 2, 0), starts, ends, np.minimum(ends + dur // 2, t)], 1)
        ticks[0] = [0, 0, 3, 6]        # proposal at the very start (starting stage of length 0 -> 1 row)
        ticks[1] = [30, 38, 40, 40]    # proposal ending at T: ending stage starts at T -> skipped (:140-142)
        ticks[2] = [5, 5, 5, 5]        # degenerate (length-0 stages -> max(t+1, .))
        ticks[3] = [0, 0, 0, 2]
        scaling = rng.uniform(0, 1, (12, 2))
        mod = ref_ops.STPPReorgainzed(d, a, c, r, True, True, stpp_cfg=cfg)
        oa, oc, orr = mod.forward(torch.from_numpy(scores), torch.from_numpy(ticks.astype(np.int64)), scaling)
        out.update({"r%d_scores" % ci: scores, "r%d_ticks" % ci: ticks.astype(np.int64), "r%d_scaling" % ci: scaling,
                    "r%d_act" % ci: oa.numpy(), "r%d_comp" % ci: oc.numpy(), "r%d_reg" % ci: orr.numpy()})
    # [r6] the form WITHOUT the stand-alone activity classifier (ops/ssn_ops.py:160-161: the activity scores go through pspool too; their
    # block is act_len * multiplier wide).  Own generator, behind the cases above: r0 / r1 stay byte-identical.
    rng = np.random.RandomState(131)
    cfg = (1, (1, 2), 1)
    mult, (a, c, r), t = 5, (21, 20, 40), 40
    d = (a + c + r) * mult
    scores = rng.standard_normal((t, d)).astype(np.float32)
    starts = rng.randint(0, t - 2, size=10)
    ends = np.minimum(starts + rng.randint(1, 12, size=10), t)
    dur = ends - starts
    ticks = np.stack([np.maximum(starts - dur // 2, 0), starts, ends, np.minimum(ends + dur // 2, t)], 1)
    ticks[0] = [0, 0, 3, 6]
    ticks[1] = [30, 38, 40, 40]
    ticks[2] = [5, 5, 5, 5]
    scaling = rng.uniform(0, 1, (10, 2))
    mod = ref_ops.STPPReorgainzed(d, a, c, r, False, True, stpp_cfg=cfg)
    oa, oc, orr = mod.forward(torch.from_numpy(scores), torch.from_numpy(ticks.astype(np.int64)), scaling)
    out.update({"r2_scores": scores, "r2_ticks": ticks.astype(np.int64), "r2_scaling": scaling,
                "r2_act": oa.numpy(), "r2_comp": oc.numpy(), "r2_reg": orr.numpy()})
    np.savez_compressed(os.path.join(OUT, "ref_reorg.npz"), **out)
    if os.environ.get("GOLDEN_ONLY") == "reorg":
        raise SystemExit(0)


def golden_ssn(ref_models, ref_ops):
    """Reference SSN wiring (heads, row selection, losses, test_fc folding, optimiser groups)."""
    out = {}
    for tag, modality, cfg, size in (("rgb", "RGB", (1, 1, 1), 32), ("flow", "Flow", (1, (1, 2), 1), 32)):
        torch.manual_seed(0)
        c, v = 20, 2
        m = ref_models.SSN(c, 2, 5, 2, modality, base_model="BNInception", dropout=0, stpp_cfg=cfg)
        init_backbone_synthetic(m.base_model)
        init_heads_synthetic(m)
        m.train()
        batch = make_batch(v, modality, c, seed=3, input_size=size)
        res = m(*batch)
        act_l = torch.nn.CrossEntropyLoss()(res[0], res[1])
        comp_l = ref_ops.CompletenessLoss()(res[2], res[3], 1, 7)
        reg_l = ref_ops.ClassWiseRegressionLoss()(res[4], res[5], res[6])
        loss = act_l + 0.1 * comp_l + 0.1 * reg_l
        loss.backward()
        for i, t in enumerate(res):
            out["%s_out%d" % (tag, i)] = t.detach().numpy()
        out["%s_losses" % tag] = np.array([act_l.item(), comp_l.item(), reg_l.item(), loss.item()], np.float64)
        gn, names = [], []
        for n, p in m.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append([float(p.grad.double().norm()), float(p.grad.double().sum())])
        out["%s_grad_names" % tag] = np.array(names)
        out["%s_grad_stats" % tag] = np.array(gn)
        out["%s_grad_act_w" % tag] = m.activity_fc.weight.grad.numpy()
        out["%s_grad_conv1_b" % tag] = m.base_model.conv1_7x7_s2.bias.grad.numpy()
        out["%s_grad_5b_1x1_w" % tag] = m.base_model.inception_5b_1x1.weight.grad.numpy()
        pol = m.get_optim_policies()
        out["%s_policy_sizes" % tag] = np.array([[len(g["params"]), sum(p.numel() for p in g["params"])] for g in pol])
        out["%s_state_keys" % tag] = np.array(sorted(m.state_dict().keys()))
        # dense-test path
        m.test_mode = True
        m.prepare_test_fc()
        m.eval()
        with torch.no_grad():
            frames = batch[0].reshape(-1, batch[0].shape[1] // 72 * 1, size, size)[:12]
            scores, base = m(frames.reshape(12, -1, size, size), None, None, None, None)
        out["%s_test_scores" % tag] = scores.numpy()
        out["%s_test_base" % tag] = base.numpy()
    np.savez_compressed(os.path.join(OUT, "ref_ssn.npz"), **out)


def golden_ssn_bn(ref_models, ref_ops):
    """bn_mode 'partial' / 'full' of the reference SSN (/root/reference/ssn_models.py:95-105,156-174): the first /
    every BatchNorm2d normalises with batch statistics, updates its running statistics and gets gamma / beta gradients."""
    out = {}
    for mode in ("partial", "full"):
        torch.manual_seed(0)
        c, v, size = 20, 2, 32
        m = ref_models.SSN(c, 2, 5, 2, "RGB", base_model="BNInception", dropout=0, stpp_cfg=(1, 1, 1), bn_mode=mode)
        init_backbone_synthetic(m.base_model)
        init_heads_synthetic(m)
        m.train()
        out["%s_training_bn" % mode] = np.array([n for n, mod in m.base_model.named_modules()
                                                if isinstance(mod, torch.nn.BatchNorm2d) and mod.training])
        batch = make_batch(v, "RGB", c, seed=3, input_size=size)
        res = m(*batch)
        act_l = torch.nn.CrossEntropyLoss()(res[0], res[1])
        comp_l = ref_ops.CompletenessLoss()(res[2], res[3], 1, 7)
        reg_l = ref_ops.ClassWiseRegressionLoss()(res[4], res[5], res[6])
        loss = act_l + 0.1 * comp_l + 0.1 * reg_l
        loss.backward()
        for i, t in enumerate(res):
            out["%s_out%d" % (mode, i)] = t.detach().numpy()
        out["%s_losses" % mode] = np.array([act_l.item(), comp_l.item(), reg_l.item(), loss.item()], np.float64)
        gn, names = [], []
        for n, p in m.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append([float(p.grad.double().norm()), float(p.grad.double().sum())])
        out["%s_grad_names" % mode] = np.array(names)
        out["%s_grad_stats" % mode] = np.array(gn)
        bn1 = m.base_model.conv1_7x7_s2_bn
        out["%s_bn1_running_mean" % mode] = bn1.running_mean.numpy().copy()
        out["%s_bn1_running_var" % mode] = bn1.running_var.numpy().copy()
        out["%s_bn1_dgamma" % mode] = bn1.weight.grad.numpy().copy()
        out["%s_bn1_dbeta" % mode] = bn1.bias.grad.numpy().copy()
        out["%s_conv1_dw" % mode] = m.base_model.conv1_7x7_s2.weight.grad.numpy().copy()
        bn5 = m.base_model.inception_5b_1x1_bn
        out["%s_bn5b_running_mean" % mode] = bn5.running_mean.numpy().copy()
        out["%s_bn5b_running_var" % mode] = bn5.running_var.numpy().copy()
        out["%s_bn1_batches" % mode] = np.array([int(bn1.num_batches_tracked), int(bn5.num_batches_tracked)])
        pol = m.get_optim_policies()
        out["%s_policy_sizes" % mode] = np.array([[len(g["params"]), sum(p.numel() for p in g["params"])] for g in pol])
    np.savez_compressed(os.path.join(OUT, "ref_ssn_bn.npz"), **out)


def golden_ssn_rgbdiff(ref_models, ref_ops):
    """RGBDiff modality of the reference SSN (_get_diff + _construct_diff_model, ssn_models.py:302-316,345-376).
    The reference's _construct_diff_model subscripts the result of ``filter`` (a Python-2 list); the shim below gives the
    module a list-returning ``filter`` -- the fourth shim next to torchvision / model_zoo / .cuda(); no source edit."""
    import builtins
    ref_models.filter = lambda f, it: list(builtins.filter(f, it))
    out = {}
    torch.manual_seed(0)
    c, v, size = 20, 2, 32
    m = ref_models.SSN(c, 2, 5, 2, "RGBDiff", base_model="BNInception", dropout=0, stpp_cfg=(1, 1, 1))
    init_backbone_synthetic(m.base_model)
    init_heads_synthetic(m)
    m.train()
    batch = make_batch(v, "RGBDiff", c, seed=3, input_size=size)
    res = m(*batch)
    act_l = torch.nn.CrossEntropyLoss()(res[0], res[1])
    comp_l = ref_ops.CompletenessLoss()(res[2], res[3], 1, 7)
    reg_l = ref_ops.ClassWiseRegressionLoss()(res[4], res[5], res[6])
    loss = act_l + 0.1 * comp_l + 0.1 * reg_l
    loss.backward()
    for i, t in enumerate(res):
        out["out%d" % i] = t.detach().numpy()
    out["losses"] = np.array([act_l.item(), comp_l.item(), reg_l.item(), loss.item()], np.float64)
    gn, names = [], []
    for n, p in m.named_parameters():
        if p.grad is not None:
            names.append(n)
            gn.append([float(p.grad.double().norm()), float(p.grad.double().sum())])
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.array(gn)
    out["conv1_weight_shape"] = np.array(m.base_model.conv1_7x7_s2.weight.shape)
    out["conv1_dw"] = m.base_model.conv1_7x7_s2.weight.grad.numpy().copy()
    out["diff_sample"] = m._get_diff(batch[0]).reshape(-1, 15, size, size)[:3].numpy().copy()
    out["input_mean"] = np.array(m.input_mean)
    out["state_keys"] = np.array(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "ref_ssn_rgbdiff.npz"), **out)


def golden_binary(ref_binary, ref_ops):
    """BinaryClassifier (binary_model.py) on the stub backbone: train forward + CE loss + gradients, test path."""
    ref_binary.Identity = ref_ops.Identity      # the reference forgets to import it (NameError at dropout == 0)
    out = {}
    for tag, modality, size in (("rgb", "RGB", 32), ("flow", "Flow", 32)):
        torch.manual_seed(0)
        seg, n = 3, 4
        m = ref_binary.BinaryClassifier(2, seg, modality, base_model="BNInception", dropout=0)
        init_backbone_synthetic(m.base_model)
        rs = np.random.RandomState(7)
        with torch.no_grad():
            m.classifier_fc.weight.copy_(torch.from_numpy(rs.standard_normal(m.classifier_fc.weight.shape).astype(np.float32) * 0.05))
            m.classifier_fc.bias.copy_(torch.from_numpy(rs.standard_normal(2).astype(np.float32) * 0.05))
        m.train()
        c = 3 if modality == "RGB" else 10
        x = torch.from_numpy(rs.randint(0, 256, (n, seg * c, size, size)).astype(np.float32)) - 110.0
        tgt = torch.from_numpy(rs.randint(0, 2, (n, 1)).astype(np.int64))
        logits, t = m(x, tgt)
        loss = torch.nn.CrossEntropyLoss()(logits, t)
        loss.backward()
        out.update({tag + "_x": x.numpy(), tag + "_target": tgt.numpy(), tag + "_logits": logits.detach().numpy(),
                    tag + "_loss": np.array([loss.item()]), tag + "_fc_w": m.classifier_fc.weight.detach().numpy(),
                    tag + "_fc_b": m.classifier_fc.bias.detach().numpy(),
                    tag + "_grad_fc_w": m.classifier_fc.weight.grad.numpy(),
                    tag + "_grad_conv1_w": m.base_model.conv1_7x7_s2.weight.grad.numpy(),
                    tag + "_grad_5b_1x1_b": m.base_model.inception_5b_1x1.bias.grad.numpy()})
        pol = m.get_optim_policies()
        out[tag + "_policy_sizes"] = np.array([[len(g["params"]), sum(p.numel() for p in g["params"])] for g in pol])
        out[tag + "_state_keys"] = np.array(sorted(m.state_dict().keys()))
        m.test_mode = True
        m.prepare_test_fc()
        m.eval()
        with torch.no_grad():
            sc, base = m(x[:, :c], None)
        out[tag + "_test_scores"] = sc.numpy()
        out[tag + "_test_base"] = base.numpy()
    np.savez_compressed(os.path.join(OUT, "ref_binary.npz"), **out)


def golden_proposal_io():
    """ops/io.py: parse a normalised list (a slice of the reference's own ActivityNet list + synthetic corner cases:
    a video without proposals, one without ground truth) and convert it to a processed list."""
    import json
    from ops import io as ref_io
    src = open(os.path.join(REF, "data", "activitynet1.2_tag_val_normalized_proposal_list.txt")).read().split("# ")
    text = "".join("# " + r for r in src[1:6])
    text += "# 6\nvid_no_props\n1\n1\n1\n3 0.1000 0.9000\n0\n# 7\nvid_no_gt\n1\n1\n0\n2\n0 0.0000 0.0000 0.1000 0.4000\n7 0.5000 1.0000 0.2500 0.7500\n"
    norm_path = os.path.join(OUT, "proposal_list_norm.txt")
    open(norm_path, "w").write(text)
    parsed = ref_io.load_proposal_file(norm_path)
    frame_dict = {rec[0]: ("frames/" + rec[0], 900 + 37 * i, 0) for i, rec in enumerate(parsed)}
    proc_path = os.path.join(OUT, "proposal_list_processed.txt")
    ref_io.process_proposal_list(norm_path, proc_path, frame_dict)
    reparsed = ref_io.load_proposal_file(proc_path)
    json.dump({"parsed": parsed, "frame_dict": frame_dict, "reparsed": reparsed},
              open(os.path.join(OUT, "proposal_list_expected.json"), "w"))


def golden_sampling():
    """SSNDataSet (ssn_dataset.py) without image files: `_load_image` returns the frame INDEX it was asked for and the
    transform stacks those, so `get_training_data` yields exactly which frames the reference would have loaded."""
    import json
    np.int = int                                   # ssn_dataset.py:397 uses the alias numpy removed
    import ssn_dataset as ref_ds
    prop_file = os.path.join(OUT, "proposal_list_processed.txt")
    out = {}
    for tag, kw in (("train", dict(random_shift=True)), ("val", dict(random_shift=False)),
                    ("flow", dict(random_shift=True, new_length=5))):
        ds = ref_ds.SSNDataSet("", prop_file, transform=lambda fr: torch.tensor(fr, dtype=torch.int64), verbose=False, **kw)
        ds._load_image = lambda directory, idx: [idx]
        rec = {"stats": np.asarray(ds.stats).tolist(), "n_videos": len(ds.video_list),
               "pools": [len(ds.fg_pool), len(ds.incomp_pool), len(ds.bg_pool)], "samples": [], "tests": []}
        for i in range(len(ds.video_list)):
            for seed in (0, 1):
                np.random.seed(100 * i + seed)
                fr, plen, scal, ptype, lab, reg, split = ds.get_training_data(i)
                rec["samples"].append({"video": i, "seed": 100 * i + seed, "frames": fr.tolist(),
                                       "scaling": scal.numpy().tolist(), "prop_type": ptype.tolist(),
                                       "labels": lab.tolist(), "reg_targets": reg.numpy().tolist(),
                                       "stage_split": split.tolist()})
            gen, n_ticks, rel, pticks, scaling = ds.get_test_data(ds.video_list[i], ds.test_interval)
            rec["tests"].append({"video": i, "n_ticks": int(n_ticks), "rel": rel.numpy().tolist(),
                                 "ticks": pticks.numpy().tolist(), "scaling": scaling.numpy().tolist()})
        rec["all_gt"] = ds.get_all_gt()
        out[tag] = rec
    json.dump(out, open(os.path.join(OUT, "sampling_expected.json"), "w"))


def _reference_functions(path, names):
    """Compile the named top-level functions of a reference SCRIPT (one that parses argv and loads files at import
    time, so it cannot be imported) from the file where it lies; returns the namespace they were defined in -- the
    caller fills in the module globals they use.  Nothing is copied: the source is read and compiled at run time."""
    import ast
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names)
    ns = {}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def golden_detection():
    """eval_detection_results.py:91-128 (gen_detection_results), :158-171 (perform_regression) and
    ops/utils.py:35-37, 56-82 (softmax, temporal_nms) -- the reference's own functions -- on seeded videos.  No case
    holds two exactly equal fused scores (asserted), so numpy's unstable default sort cannot influence the result."""
    from ops import utils as ref_utils
    out = {}
    cases = [  # P, C, top_k, nms, no_regression, with_reg, seed, spoil
        (60, 6, 100, 0.2, False, True, 1, None),      # THUMOS14-style: top-k over all pairs
        (40, 4, 0, 0.4, False, True, 2, None),        # all-pairs branch (softmax incl. background)
        (30, 5, 10 ** 6, 0.3, True, True, 3, None),   # top_k > number of pairs, --no_regression
        (25, 3, 20, 0.6, False, False, 4, None),      # no regression scores in the pickle (None)
        (1, 3, 2, 0.5, False, True, 5, None),         # a single proposal
        (50, 4, 40, 0.5, False, True, 6, "overflow"),  # completeness scores that overflow exp(): one inf, one NaN
    ]
    for ci, (p, c, top_k, thr, no_reg, with_reg, seed, spoil) in enumerate(cases):
        rs = np.random.RandomState(100 + seed)
        start = rs.uniform(0, 0.8, p)
        rel = np.stack([start, np.minimum(start + rs.uniform(0.02, 0.5, p), 1.0)], axis=1)
        act = (rs.standard_normal((p, c + 1)) * 2).astype(np.float32)
        comp = rs.standard_normal((p, c)).astype(np.float32)
        reg = (rs.standard_normal((p, c, 2)) * 0.3).astype(np.float32) if with_reg else None
        if spoil == "overflow":
            comp[7, 1] = 95.0            # exp -> inf, times a positive softmax -> inf
            comp[11, 2] = 120.0          # exp -> inf ...
            act[11, :] = [0, 0, 0, -200, 0][:c + 1]     # ... times a softmax that underflows to 0 -> NaN
        ns = _reference_functions(os.path.join(REF, "eval_detection_results.py"),
                                  ["gen_detection_results", "perform_regression"])
        ns.update(np=np, softmax=ref_utils.softmax, num_class=c, top_k=top_k, cls_score_dict=None,
                  dataset_detections=[dict() for _ in range(c)])
        with np.errstate(all="ignore"):
            ns["gen_detection_results"]("v", (rel[None], act, comp, reg))
            dets = [{k: ref_utils.temporal_nms(v, thr) for k, v in d.items()} for d in ns["dataset_detections"]]
            if not no_reg:
                dets = [{k: ns["perform_regression"](v) for k, v in d.items()} for d in dets]
            sm = ref_utils.softmax(act)[:, 1:] if top_k <= 0 else ref_utils.softmax(act[:, 1:])
            combined = sm * np.exp(comp)
        flat = combined.ravel()
        finite = flat[np.isfinite(flat)]
        assert len(np.unique(finite)) == len(finite) and (~np.isfinite(flat)).sum() <= 2, "exact ties in fixture %d" % ci
        rows = [d["v"] if "v" in d else np.zeros((0, 5)) for d in dets]
        out.update({"d%d_rel" % ci: rel, "d%d_act" % ci: act, "d%d_comp" % ci: comp,
                    "d%d_reg" % ci: reg if with_reg else np.zeros(0, np.float32),
                    "d%d_cfg" % ci: np.array([p, c, top_k, int(no_reg), int(with_reg)], np.int64),
                    "d%d_thr" % ci: np.array([thr]), "d%d_counts" % ci: np.array([len(r) for r in rows], np.int64),
                    "d%d_dets" % ci: np.concatenate(rows, 0), "d%d_combined" % ci: combined.astype(np.float32)})
    out["n_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "ref_detection.npz"), **out)


def golden_detection_cls():
    """The `--cls_scores` branch of eval_detection_results.py (:82-90, :130-144): detections only for the `--cls_top_k` classes
    an external video-level classifier ranks highest, every proposal kept, fused score = (softmax(act)[:, 1:] with
    `--softmax_before_filter`, else the raw class scores act[:, 1:]) * exp(comp); then temporal_nms and perform_regression as
    in golden_detection.  The reference's own functions on seeded videos."""
    import argparse
    from ops import utils as ref_utils
    out = {}
    cases = [  # P, C, cls_top_k, softmax_bf, nms, no_regression, with_reg, seed
        (40, 6, 1, False, 0.4, False, True, 1),       # the defaults: one class, raw activity scores
        (35, 5, 2, True, 0.3, False, True, 2),        # --cls_top_k 2 --softmax_before_filter
        (20, 4, 3, False, 0.5, True, False, 3),       # --no_regression, no regression scores in the pickle
        (1, 3, 1, True, 0.5, False, True, 4),         # a single proposal
    ]
    for ci, (p, c, ktop, sbf, thr, no_reg, with_reg, seed) in enumerate(cases):
        rs = np.random.RandomState(300 + seed)
        start = rs.uniform(0, 0.8, p)
        rel = np.stack([start, np.minimum(start + rs.uniform(0.02, 0.5, p), 1.0)], axis=1)
        act = (rs.standard_normal((p, c + 1)) * 2).astype(np.float32)
        comp = rs.standard_normal((p, c)).astype(np.float32)
        reg = (rs.standard_normal((p, c, 2)) * 0.3).astype(np.float32) if with_reg else None
        vcls = rs.standard_normal(c).astype(np.float32)        # the external classifier's scores of this video
        ns = _reference_functions(os.path.join(REF, "eval_detection_results.py"),
                                  ["gen_detection_results", "perform_regression"])
        ns.update(np=np, os=os, softmax=ref_utils.softmax, num_class=c, top_k=0, cls_score_dict={"v": vcls}, softmax_bf=sbf,
                  args=argparse.Namespace(cls_top_k=ktop), dataset_detections=[dict() for _ in range(c)])
        ns["gen_detection_results"]("some/dir/v.mp4", (rel[None], act, comp, reg))
        dets = [{k: ref_utils.temporal_nms(v, thr) for k, v in d.items()} for d in ns["dataset_detections"]]
        if not no_reg:
            dets = [{k: ns["perform_regression"](v) for k, v in d.items()} for d in dets]
        combined = (ref_utils.softmax(act)[:, 1:] if sbf else act[:, 1:]) * np.exp(comp)
        assert len(np.unique(combined.ravel())) == combined.size, "exact ties in fixture %d" % ci
        rows = [d["some/dir/v.mp4"] if "some/dir/v.mp4" in d else np.zeros((0, 5)) for d in dets]
        assert sum(len(r) > 0 for r in rows) == min(ktop, c)
        out.update({"e%d_rel" % ci: rel, "e%d_act" % ci: act, "e%d_comp" % ci: comp, "e%d_vcls" % ci: vcls,
                    "e%d_reg" % ci: reg if with_reg else np.zeros(0, np.float32),
                    "e%d_cfg" % ci: np.array([p, c, ktop, int(sbf), int(no_reg), int(with_reg)], np.int64),
                    "e%d_thr" % ci: np.array([thr]), "e%d_counts" % ci: np.array([len(r) for r in rows], np.int64),
                    "e%d_dets" % ci: np.concatenate(rows, 0), "e%d_combined" % ci: combined.astype(np.float32)})
    out["n_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "ref_detection_cls.npz"), **out)


def golden_transforms():
    """transforms.py on PIL images: the test chain GroupOverSample -> Stack(roll) -> ToTorchFormatTensor(div=False) ->
    GroupNormalize (ssn_test.py:101-112, ssn_dataset.py:434-450) and the training augmentation
    GroupMultiScaleCrop + GroupRandomHorizontalFlip (ssn_models.py:386-395) with seeded `random`."""
    import random
    from PIL import Image
    import transforms as ref_tf
    rs = np.random.RandomState(3)
    out = {}

    def chain(group, roll, mean, std):
        x = ref_tf.Stack(roll=roll)(group)
        x = ref_tf.ToTorchFormatTensor(div=False)(x)
        return ref_tf.GroupNormalize(list(mean), list(std))(x).numpy()

    rgb = rs.randint(0, 256, size=(3, 32, 43, 3)).astype(np.uint8)
    flow = rs.randint(0, 256, size=(10, 30, 40)).astype(np.uint8)
    out["rgb_frames"], out["flow_frames"] = rgb, flow
    rgb_imgs = [Image.fromarray(f, "RGB") for f in rgb]
    flow_imgs = [Image.fromarray(f, "L") for f in flow]
    out["over_rgb_roll"] = chain(ref_tf.GroupOverSample(24)(rgb_imgs), True, [104, 117, 128], [1])
    out["over_rgb_std"] = chain(ref_tf.GroupOverSample((28, 20))(rgb_imgs), False, [0.485, 0.456, 0.406],
                                [0.229, 0.224, 0.225])
    out["over_flow"] = chain(ref_tf.GroupOverSample(22)(flow_imgs), True, [128], [1])
    # training augmentation (host side; PIL crop + bilinear resize + flip) at several seeds
    big_rgb = rs.randint(0, 256, size=(2, 64, 86, 3)).astype(np.uint8)
    big_flow = rs.randint(0, 256, size=(4, 64, 86)).astype(np.uint8)
    out["aug_rgb_frames"], out["aug_flow_frames"] = big_rgb, big_flow
    for tag, imgs, scales, is_flow in (("rgb", [Image.fromarray(f, "RGB") for f in big_rgb], [1, .875, .75, .66], False),
                                       ("flow", [Image.fromarray(f, "L") for f in big_flow], [1, .875, .75], True)):
        for seed in range(6):
            random.seed(seed)
            g = ref_tf.GroupMultiScaleCrop(56, scales)(imgs)
            g = ref_tf.GroupRandomHorizontalFlip(is_flow=is_flow)(g)
            out["aug_%s_%d" % (tag, seed)] = np.stack([np.asarray(im) for im in g])
    # crop-parameter sampling alone, for a sweep of frame sizes (no pixels involved)
    rec = []
    for (w, h) in ((340, 256), (320, 240), (455, 256), (256, 256), (224, 224)):
        for seed in range(8):
            random.seed(1000 + seed)
            cw, ch, ow, oh = ref_tf.GroupMultiScaleCrop(224, [1, .875, .75, .66])._sample_crop_size((w, h))
            rec.append([w, h, 1000 + seed, cw, ch, ow, oh, int(random.random() < 0.5)])
    out["crop_params"] = np.array(rec, np.int64)
    # the training augmentation at the REAL sizes (256 x 340 decoded frame -> 224 network input; every scale of the jitter:
    # 256 -> 224 shrinks with 5 taps, 192 / 168 -> 224 enlarge with 3, 224 is copied): pins the GPU restatement of PIL's resize
    full = rs.randint(0, 256, size=(1, 256, 340, 3)).astype(np.uint8)
    # (smooth content in the right half: noise alone would hide a wrong tap position behind its own variance)
    yy, xx = np.mgrid[0:256, 0:340]
    smooth = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 64) % 256], axis=2).astype(np.uint8)
    full[0, :, 170:] = smooth[:, 170:]
    out["aug_full_frames"] = full
    imgs = [Image.fromarray(f, "RGB") for f in full]
    boxes = []
    for seed in (0, 1, 3, 4, 7):        # crop sizes 256x224, 168x168, 168x192, 256x256, 224x224 (w x h)
        random.seed(100 + seed)
        msc = ref_tf.GroupMultiScaleCrop(224, [1, .875, .75, .66])
        st = random.getstate()
        boxes.append(list(msc._sample_crop_size((340, 256))))
        random.setstate(st)
        g = ref_tf.GroupRandomHorizontalFlip(is_flow=False)(msc(imgs))
        out["aug_full_%d" % seed] = np.stack([np.asarray(im) for im in g])
    out["aug_full_boxes"] = np.array(boxes, np.int64)        # (crop_w, crop_h, offset_w, offset_h) per seed
    np.savez_compressed(os.path.join(OUT, "ref_transforms.npz"), **out)


def main():
    assert os.path.isdir(REF), "this script needs /root/reference (build container only)"
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    if os.environ.get("GOLDEN_ONLY") == "detection_cls":      # (only this fixture: the others stay byte-identical)
        golden_detection_cls()
        return
    import ssn_models as ref_models
    from ops import ssn_ops as ref_ops
    golden_stpp(ref_ops)
    golden_losses(ref_ops)
    golden_reorg(ref_ops)
    if not os.environ.get("GOLDEN_ONLY"):
        golden_ssn(ref_models, ref_ops)
    golden_ssn_bn(ref_models, ref_ops)
    golden_ssn_rgbdiff(ref_models, ref_ops)
    if os.environ.get("GOLDEN_ONLY") == "ssn_bn":
        return
    import binary_model as ref_binary
    golden_binary(ref_binary, ref_ops)
    golden_proposal_io()
    golden_detection()
    golden_detection_cls()
    golden_transforms()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
