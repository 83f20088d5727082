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
        self._fused_args = {}                     # the moment tensors were replaced: rebuild the cached pointer tables (and counts)

    def _fusable(self, group, params):
        return (params and not group.get("amsgrad") and not group.get("maximize") and not group.get("capturable")
                and not group.get("differentiable") and len(params) <= 32
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and not p.grad.is_sparse and p.grad.dtype == torch.float32 for p in params)
                and not isinstance(group["lr"], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            return super().step(closure)
        if not self._step_impl():
            return super().step(closure)
        return None

    def step_fused(self):
        """``step()`` without torch.optim's per-call wrapper (profiler record, pre / post hook dispatch: ~8 us of host time a
        step): the models' ``evaluate`` calls this when the optimizer offers it.  Optimizer step hooks are NOT run here."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks:
            return self.step()
        if hasattr(self.step, "_wrapped_by_lr_sched"):
            self._opt_called = True               # what the lr_scheduler's wrapper around step() records (its order check)
        with torch.no_grad():
            if not self._step_impl():
                return torch.optim.Adam.step(self)
        return None

    def defer_step(self, device):
        """Registers this step with the device's workspace instead of launching it: the engine folds the update into the last
        launch of the training meta-step that follows (``fumi_hip_adam_step_deferred``; single process only -- the caller checks).
        True when registered; the caller then runs the meta-step and ``finish_deferred``.  False (nothing done) when the step
        has to go the ordinary way: first step (no gradient views yet), several groups, hooks, a non-fusable configuration."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or len(self.param_groups) != 1:
            return False
        group = self.param_groups[0]
        cached = self._fused_args.get(0)
        params = [p for p in group["params"] if p.grad is not None]
        if cached is None or not params or len(params) > 24:
            return False
        ident = tuple(p.data_ptr() for p in params) + tuple(p.grad.data_ptr() for p in params)
        if cached.ident != ident or isinstance(group["lr"], torch.Tensor) or group.get("amsgrad") or group.get("maximize"):
            return False
        if hasattr(self.step, "_wrapped_by_lr_sched"):
            self._opt_called = True
        cached.count += 1
        b1, b2 = group["betas"]
        hip.adam_step_deferred(hip.Workspace.get(device), cached, group["lr"], b1, b2, group["eps"], group["weight_decay"], cached.count)
        return True

    def finish_deferred(self, device):
        """After the meta-step: launches the registered update on its own if the step could not fold it."""
        return hip.adam_flush(hip.Workspace.get(device), device)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def _sync_steps(self):
        """The per-parameter ``step`` tensors (torch.optim.Adam's state layout) are brought up to date lazily: the fused launch only
        needs the count, and eight tiny tensor increments cost ~5 us of host time a step."""
        for args in self._fused_args.values():
            lag = args.count - args.synced
            if lag:
                torch._foreach_add_(args.steps, lag)
                args.synced = args.count

    def _fall_back(self):
        """torch's own step is about to run for EVERY group and will increment every `step` tensor itself: bring the tensors
        up to date and forget the cached plans, so the next fused step reads its counts from the tensors again."""
        self._sync_steps()
        self._fused_args = {}
        return False

    def __getstate__(self):
        self._sync_steps()                        # pickling / deepcopy read optimizer.state directly
        return super().__getstate__()

    def __deepcopy__(self, memo):
        import copy
        self._sync_steps()
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, {} if k == "_fused_args" else copy.deepcopy(v, memo))
        return new

    def _step_impl(self):
        """True when every group went through the fused launch; False (nothing done) when torch's implementation must run."""
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        plans = []
        for gi, (group, params) in enumerate(groups):
            if not params:
                continue
            grads = [p.grad for p in params]
            cached = self._fused_args.get(gi)
            # same tensors as last step (the usual case: parameters and the flat gradient views are stable): skip the checks
            ident = tuple(p.data_ptr() for p in params) + tuple(g.data_ptr() for g in grads)
            if cached is None or cached.ident != ident:
                if cached is not None:
                    self._sync_steps()            # (the replaced entry's pending count goes into the `step` tensors first)
                if not self._fusable(group, params):
                    return self._fall_back()
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
                cached.count = cached.synced = int(cached.steps[0])
                cached.dev = params[0].device
            elif isinstance(group["lr"], torch.Tensor) or group.get("amsgrad") or group.get("maximize"):
                return self._fall_back()
            plans.append((group, cached))
        for group, args in plans:
            args.count += 1                                                     # (the `step` tensors follow in _sync_steps)
            b1, b2 = group["betas"]
            hip.adam_step(hip.Workspace.get(args.dev), args, group["lr"], b1, b2, group["eps"], group["weight_decay"], args.count, args.dev)
        return True
