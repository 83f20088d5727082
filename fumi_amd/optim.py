"""Optimizers for the outer loop.  ``Adam`` is torch.optim.Adam (same constructor, ``param_groups``, ``state_dict`` layout:
``step`` / ``exp_avg`` / ``exp_avg_sq``) whose ``step()`` runs as ONE fused HIP launch over all parameter tensors when they
are fp32 GPU tensors (csrc/adam.hip); any other configuration (amsgrad, maximize, CPU tensors, sparse grads ...) uses
torch's own implementation unchanged."""
import torch

from . import hip


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._fused_args = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._fused_args = {}                     # the moment tensors were replaced: rebuild the cached pointer tables

    def _fusable(self, group, params):
        return (params and not group.get("amsgrad") and not group.get("maximize") and not group.get("capturable")
                and not group.get("differentiable") and len(params) <= 32
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and not p.grad.is_sparse and p.grad.dtype == torch.float32 for p in params)
                and not isinstance(group["lr"], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        if closure is not None:
            return super().step(closure)
        plans = []
        for gi, (group, params) in enumerate(groups):
            if not params:
                continue
            grads = [p.grad for p in params]
            cached = self._fused_args.get(gi)
            # same tensors as last step (the usual case: parameters and the flat gradient views are stable): skip the checks
            ident = tuple(p.data_ptr() for p in params) + tuple(g.data_ptr() for g in grads)
            if cached is None or cached.ident != ident:
                if not self._fusable(group, params):
                    return super().step(closure)
                for p in params:
                    st = self.state[p]
                    if len(st) == 0:
                        st["step"] = torch.tensor(0.0, dtype=torch.float32)        # same layout as torch.optim.Adam
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ms = [self.state[p]["exp_avg"] for p in params]
                vs = [self.state[p]["exp_avg_sq"] for p in params]
                cached = self._fused_args[gi] = hip.AdamArgs(list(params), grads, ms, vs)
                cached.ident = ident
                cached.steps = [self.state[p]["step"] for p in params]
                cached.dev = params[0].device
            elif isinstance(group["lr"], torch.Tensor) or group.get("amsgrad") or group.get("maximize"):
                return super().step(closure)
            plans.append((group, cached))
        for group, args in plans:
            step = int(args.steps[0]) + 1
            torch._foreach_add_(args.steps, 1)                                  # the per-parameter `step` tensors stay in sync
            b1, b2 = group["betas"]
            hip.adam_step(hip.Workspace.get(args.dev), args, group["lr"], b1, b2, group["eps"], group["weight_decay"], step, args.dev)
        return None
