"""``Adam`` with torch.optim.Adam's surface (train.py:26-27: ``Adam(model.parameters(), lr=0.001)``), fused.

When every parameter is a view of one ``ParamArena`` (always the case for ``UNetModel.parameters()``) a step is a
single streaming launch of ``pg_adam_step`` over the whole arena: 28 B of HBM traffic per parameter, no per-tensor
launches.  ``zero_grad()`` is free: wgrad / BN-backward kernels overwrite the gradient arena.
"""
import torch

from . import ops


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay or amsgrad:
            raise NotImplementedError("phasegen Adam implements the reference's configuration (defaults) only")
        self.params = list(params)
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        arenas = {id(getattr(p, "_pg_arena", None)) for p in self.params}
        self.arena = getattr(self.params[0], "_pg_arena", None) if len(arenas) == 1 else None
        if self.arena is not None:
            self.m = torch.zeros_like(self.arena.flat)
            self.v = torch.zeros_like(self.arena.flat)
        else:
            self.m = [torch.zeros_like(p) for p in self.params]
            self.v = [torch.zeros_like(p) for p in self.params]
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps}]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        if self.arena is not None:
            a = self.arena
            a.touch()
            for p in self.params:        # grads that autograd materialised outside the arena are folded back in
                if p.grad is not None and p.grad.data_ptr() != a.g(p._pg_key).data_ptr():
                    a.g(p._pg_key).copy_(p.grad)
            ops.adam_step(a.flat, a.grad, self.m, self.v, self.step_count, lr, b1, b2, self.eps, grad_scale)
        else:
            for p, m, v in zip(self.params, self.m, self.v):
                if p.grad is None:
                    continue
                ops.adam_step(p.data.view(-1), p.grad.contiguous().view(-1), m.view(-1), v.view(-1), self.step_count,
                              lr, b1, b2, self.eps, grad_scale)

    # -- per-layer stepping (phasegen.trainer overlaps these with the rest of backward on a side stream) -------------------
    def begin_step(self):
        self.step_count += 1
        if self.arena is not None:
            self.arena.touch()

    @torch.no_grad()
    def step_range(self, start, end, grad_scale=1.0, thin=False):
        """The update of arena elements [start, end) for the step begun with begin_step() -- elementwise, so stepping the
        arena range by range gives bit-identical results to one step() over the whole arena."""
        a = self.arena
        b1, b2 = self.betas
        ops.adam_step(a.flat[start:end], a.grad[start:end], self.m[start:end], self.v[start:end], self.step_count,
                      self.param_groups[0]["lr"], b1, b2, self.eps, grad_scale, thin=thin)

    def fused_args(self, key, grad_scale=1.0):
        """pg_adam_args of ONE parameter for the step begun with begin_step(): handed to ``ops.conv_wgrad(adam=)`` so that the
        update of a conv weight runs in its wgrad kernel's epilogue (same arithmetic as step_range, bit for bit)."""
        a = self.arena
        b1, b2 = self.betas
        return ops.adam_args(a.p(key), a.view(key, self.m), a.view(key, self.v), self.step_count, self.param_groups[0]["lr"],
                             b1, b2, self.eps, grad_scale)

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        if self.arena is not None:
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        else:
            for d, s in zip(self.m, sd["m"]):
                d.copy_(s)
            for d, s in zip(self.v, sd["v"]):
                d.copy_(s)
