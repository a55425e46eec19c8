"""SGD with the reference's parameter-group policy, stepping through the ssn_sgd_step kernel.

Mirrors ``torch.optim.SGD(policies, lr, momentum, weight_decay)`` +
``adjust_learning_rate`` of /root/reference/ssn_train.py:141-144,391-398: every group carries
``lr_mult`` / ``decay_mult`` (from ``SSN.get_optim_policies``) and the effective
``lr = base_lr * lr_mult``, ``weight_decay = base_wd * decay_mult``.
"""
import torch

from . import kernels as K


class SSNSGD(torch.optim.Optimizer):
    def __init__(self, policies, lr, momentum=0.9, weight_decay=5e-4):
        groups = []
        for g in policies:
            if len(g["params"]) == 0:
                continue
            g = dict(g)
            g.setdefault("lr_mult", 1)
            g.setdefault("decay_mult", 1)
            groups.append(g)
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, lr_mult=1, decay_mult=1)
        super().__init__(groups, defaults)
        self.base_lr = lr
        self.base_wd = weight_decay
        self.adjust_learning_rate(0, [])

    def adjust_learning_rate(self, epoch, lr_steps):
        """lr = base * 0.1 ** (#steps passed); per-group multipliers (ssn_train.py:391-398)."""
        decay = 0.1 ** sum(1 for s in lr_steps if epoch >= s)
        for g in self.param_groups:
            g["lr"] = self.base_lr * decay * g["lr_mult"]
            g["weight_decay"] = self.base_wd * g["decay_mult"]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, skip_flag=None):
        """All parameter tensors in ceil(#tensors / 48) fused launches (ssn_sgd_step_multi).  skip_flag: device int32 tensor
        (``SSN.scale_fault_flag()``); while its first word is non-zero the launches leave weights and momentum untouched -- the
        range guard of the planes path flagged this step's gradients, the step is to be repeated (needed where the host cannot
        look before the update runs: inside a hipGraph replay)."""
        batches = {}   # momentum -> lists
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "momentum_buffer" not in st:
                    # ZERO-initialised, and every step takes the kernel's general branch: momentum * 0 + g is exactly the `buf = g`
                    # torch.optim.SGD starts with (dampening 0), and -- unlike an uninitialised buffer with a host-side "first step"
                    # marker -- it stays right when the device SKIPS this launch (skip_flag set by an earlier pass, by another rank
                    # through the MAX-reduced word, or for a parameter whose first gradient arrives in a flagged step): the next
                    # step then starts from zeros again instead of from whatever the allocation held
                    st["momentum_buffer"] = torch.zeros_like(p)
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                b = batches.setdefault(g["momentum"], ([], [], [], [], []))
                b[0].append(p.data)
                b[1].append(grad)
                b[2].append(st["momentum_buffer"])
                b[3].append(g["lr"])
                b[4].append(g["weight_decay"])
        for momentum, (ws, grads, bufs, lrs, wds) in batches.items():
            K.sgd_step_multi(ws, grads, bufs, lrs, wds, momentum, grad_scale, False, skip_flag)
        return None


def clip_grad_norm(parameters, max_norm):
    """``torch.nn.utils.clip_grad_norm`` as the reference's loop calls it (/root/reference/ssn_train.py:245-248) on the
    ssn_sumsq / ssn_scale kernels: total 2-norm over all gradients, gradients scaled by ``max_norm / (norm + 1e-6)``
    when that is below 1.  Returns the total norm as a Python float (one host sync, as in the reference)."""
    ps = [p for p in parameters if p.grad is not None]
    if not ps:
        return 0.0
    dev = ps[0].grad.device
    out = torch.zeros(1, device=dev, dtype=torch.float32)
    ws = torch.empty(1024, device=dev, dtype=torch.float32)
    for i, p in enumerate(ps):
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        K.sumsq(g, out, i > 0, ws)
    total_norm = float(out.item()) ** 0.5
    clip_coef = float(max_norm) / (total_norm + 1e-6)
    if clip_coef < 1:
        for p in ps:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            K.scale_(p.grad, None, clip_coef)
    return total_norm
