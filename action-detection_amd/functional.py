"""autograd glue: each Function is a thin pair of C-ABI launches (include/ssn_hip.h).

No arithmetic happens in torch here -- tensors are allocated with ``torch.empty`` and filled by
the HIP kernels on the current stream.
"""
import torch

from . import kernels as K


def _new(ref, shape, dtype=torch.float32):
    return torch.empty(shape, device=ref.device, dtype=dtype)


class LinearFn(torch.autograd.Function):
    """nn.Linear (/root/reference/ssn_models.py:272-273,283,300)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        out = _new(x, (x.shape[0], w.shape[0]))
        K.linear_fwd(x, w.detach().contiguous(), None if b is None else b.detach(), out)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dout = dout.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad
        dx = _new(x, x.shape) if need_x else None
        dw = _new(w, w.shape) if need_w else None
        db = _new(w, (w.shape[0],)) if (need_b and ctx.has_bias) else None
        K.linear_bwd(dout, x, w.detach().contiguous(), dx, dw, db)
        return dx, dw, db


class DropoutFn(torch.autograd.Function):
    """nn.Dropout that replaces the backbone's fc (/root/reference/ssn_models.py:74)."""

    @staticmethod
    def forward(ctx, x, p, seed, counter=None):
        x = x.contiguous()
        y = _new(x, x.shape)
        mask = _new(x, x.shape, torch.uint8)
        K.dropout_fwd(x, y, mask, p, seed, counter)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _new(dy, dy.shape)
        K.dropout_bwd(dy, mask, dx, ctx.p)
        return dx, None, None, None


class StppFn(torch.autograd.Function):
    """StructuredTemporalPyramidPooling.forward (/root/reference/ops/ssn_ops.py:39-70)."""

    @staticmethod
    def forward(ctx, ft, scaling, table, n_seg):
        ft = ft.contiguous()
        scaling = scaling.reshape(-1, 2).contiguous().float()
        d = ft.shape[1]
        if ft.shape[0] % n_seg:
            raise ValueError("feature rows (%d) are not a multiple of %d segments" % (ft.shape[0], n_seg))
        p = ft.shape[0] // n_seg
        if scaling.shape[0] != p:
            raise ValueError("scaling has %d rows, expected %d proposals" % (scaling.shape[0], p))
        act = _new(ft, (p, d))
        stpp = _new(ft, (p, table.n_parts * d))
        K.stpp_fwd(ft, scaling, act, stpp, table)
        ctx.save_for_backward(scaling)
        ctx.table = table
        ctx.shape = ft.shape
        return act, stpp

    @staticmethod
    def backward(ctx, d_act, d_stpp):
        (scaling,) = ctx.saved_tensors
        d_ft = _new(scaling, ctx.shape)
        K.stpp_bwd(d_act.contiguous(), d_stpp.contiguous(), scaling, d_ft, ctx.table)
        return d_ft, None, None, None


class HeadsFn(torch.autograd.Function):
    """STPP + activity / completeness / regression heads + the prop_type row selection of ``SSN.train_forward``
    (/root/reference/ssn_models.py:268-289) as one launch forward and one backward (csrc/heads_losses.hip: ssn_heads_*) -- the same
    arithmetic as ``StppFn`` + ``LinearFn`` x 3 + ``RowGatherFn`` x 3, which remain the path of every other caller."""

    @staticmethod
    def forward(ctx, ft, scaling, table, n_seg, idx, pos, w0, b0, w1, b1, w2, b2):
        ft = ft.contiguous()
        scaling = scaling.reshape(-1, 2).contiguous().float()
        d = ft.shape[1]
        if ft.shape[0] % n_seg:
            raise ValueError("feature rows (%d) are not a multiple of %d segments" % (ft.shape[0], n_seg))
        p = ft.shape[0] // n_seg
        if scaling.shape[0] != p:
            raise ValueError("scaling has %d rows, expected %d proposals" % (scaling.shape[0], p))
        ws = [w.detach().contiguous() if w is not None else None for w in (w0, w1, w2)]
        bs = [b.detach().contiguous() if b is not None else None for b in (b0, b1, b2)]
        act = _new(ft, (p, d))
        stpp = _new(ft, (p, table.n_parts * d))
        outs = [None if w is None else _new(ft, (i.numel(), w.shape[0])) for w, i in zip(ws, idx)]
        K.heads_fwd(ft, scaling, ws, bs, pos, idx, outs, act, stpp, table)
        ctx.save_for_backward(scaling, act, stpp, *[t for t in ws + bs if t is not None])
        ctx.layout = ([w is not None for w in ws], [b is not None for b in bs])
        ctx.table, ctx.idx, ctx.pos, ctx.ft_shape = table, idx, pos, ft.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        saved = list(ctx.saved_tensors)
        scaling, act, stpp = saved[:3]
        rest = saved[3:]
        ws = [rest.pop(0) if has else None for has in ctx.layout[0]]
        bs = [rest.pop(0) if has else None for has in ctx.layout[1]]
        douts = [None if w is None else (torch.zeros((i.numel(), w.shape[0]), device=act.device) if g is None else g.contiguous())
                 for w, i, g in zip(ws, ctx.idx, douts)]
        d_ft = _new(act, ctx.ft_shape)
        dws = [None if w is None else _new(w, w.shape) for w in ws]
        dbs = [None if b is None else _new(b, b.shape) for b in bs]
        K.heads_bwd(scaling, ws, bs, ctx.pos, ctx.idx, douts, act, stpp, ctx.table, d_ft, dws, dbs)
        return (d_ft, None, None, None, None, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2])


class RowGatherFn(torch.autograd.Function):
    """prop_type row selection (/root/reference/ssn_models.py:275-289)."""

    @staticmethod
    def forward(ctx, src, index):
        src = src.contiguous()
        out = _new(src, (index.numel(),) + tuple(src.shape[1:]))
        K.row_gather(src, index, out)
        ctx.save_for_backward(index)
        ctx.shape = src.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (index,) = ctx.saved_tensors
        dsrc = _new(dout, ctx.shape)
        K.row_scatter(dout.contiguous(), index, dsrc)
        return dsrc, None


class CrossEntropyFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss, mean reduction (/root/reference/ssn_train.py:133,210)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous()
        target = target.contiguous().long()
        r = logits.shape[0]
        loss = _new(logits, (1,))
        ws = _new(logits, (2 * r,))
        K.ce_loss_fwd(logits, target, loss, ws)
        ctx.save_for_backward(logits, target, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, ws = ctx.saved_tensors
        d = _new(logits, logits.shape)
        K.ce_loss_bwd(logits, target, ws, gout.contiguous().reshape(1), d)
        return d, None


class CompletenessFn(torch.autograd.Function):
    """CompletenessLoss + OHEMHingeLoss (/root/reference/ops/ssn_ops.py:173-239)."""

    @staticmethod
    def forward(ctx, pred, labels, group, split, keep_pos, keep_neg, den):
        pred = pred.contiguous()
        labels = labels.contiguous().long()
        r = pred.shape[0]
        loss = _new(pred, (1,))
        coef = _new(pred, (r,))
        ws = _new(pred, (2 * r,))
        K.completeness_loss_fwd(pred, labels, loss, coef, ws, group, split, keep_pos, keep_neg, den)
        ctx.save_for_backward(labels, coef)
        ctx.den = den
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        labels, coef = ctx.saved_tensors
        d = _new(coef, ctx.shape)
        K.completeness_loss_bwd(labels, coef, gout.contiguous().reshape(1), d, ctx.den)
        return d, None, None, None, None, None, None


class TotalLossFn(torch.autograd.Function):
    """[r6] activity CE + w_comp * CompletenessLoss + w_reg * ClassWiseRegressionLoss (/root/reference/ssn_train.py:210-214) as one
    launch forward and one backward (csrc/heads_losses.hip: ssn_total_loss_*): the same loss bodies and summation orders as CeLossFn /
    CompletenessFn / ClassWiseRegressionFn, bit-identical components.  Returns (total [], parts [3] = the three losses, not
    differentiable: for the meters of ssn_train.py:216-224)."""

    @staticmethod
    def forward(ctx, act, act_t, comp, comp_t, reg, reg_lbl, reg_t, group, split, keep_pos, keep_neg, den, w_comp, w_reg):
        act, comp = act.contiguous(), comp.contiguous()
        act_t, comp_t = act_t.contiguous().long(), comp_t.contiguous().long()
        has = reg is not None
        if has:
            reg, reg_lbl, reg_t = reg.contiguous(), reg_lbl.contiguous().long(), reg_t.contiguous().float()
        ra, rc = act.shape[0], comp.shape[0]
        losses = _new(act, (4,))
        lse, coef = _new(act, (ra,)), _new(act, (rc,))
        diff = _new(act, (2 * reg.shape[0],)) if has else None
        K.total_loss_fwd(act, act_t, comp, comp_t, reg, reg_lbl, reg_t, group, split, keep_pos, keep_neg, den, w_comp, w_reg, losses, lse,
                         coef, diff, _new(act, (max(2 * ra, 2 * rc),)))
        ctx.save_for_backward(act, act_t, comp_t, lse, coef, *([reg_lbl, diff] if has else []))
        ctx.meta = (tuple(comp.shape), tuple(reg.shape) if has else None, den, w_comp, w_reg)
        parts = losses[:3]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)      # (no zero tensor for the parts' absent gradient: one fill launch per step)
        return losses[3], parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        saved = ctx.saved_tensors
        act, act_t, comp_t, lse, coef = saved[:5]
        comp_shape, reg_shape, den, w_comp, w_reg = ctx.meta
        has = reg_shape is not None
        reg_lbl, diff = (saved[5], saved[6]) if has else (None, None)
        d_act, d_comp = _new(act, act.shape), _new(act, comp_shape)
        d_reg = _new(act, reg_shape) if has else None
        if g_total is None:
            return (None,) * 14
        K.total_loss_bwd(act, act_t, comp_t, comp_shape, reg_lbl, reg_shape, den, w_comp, w_reg, lse, coef, diff,
                         g_total.contiguous().reshape(1), d_act, d_comp, d_reg)
        return (d_act, None, d_comp, None, d_reg) + (None,) * 9


class ClassWiseRegressionFn(torch.autograd.Function):
    """ClassWiseRegressionLoss (/root/reference/ops/ssn_ops.py:242-258)."""

    @staticmethod
    def forward(ctx, pred, labels, targets):
        pred = pred.contiguous()
        labels = labels.contiguous().long()
        targets = targets.contiguous().float()
        n = pred.shape[0]
        loss = _new(pred, (1,))
        diff = _new(pred, (2 * n,))
        K.cw_smoothl1_fwd(pred, labels, targets, loss, diff)
        ctx.save_for_backward(labels, diff)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        labels, diff = ctx.saved_tensors
        d = _new(diff, ctx.shape)
        K.cw_smoothl1_bwd(labels, diff, gout.contiguous().reshape(1), d)
        return d, None, None
