"""Functional-parameter protocol of the reference's third-party dependency torchmeta 1.7.0
(requirements.txt:10; call sites fumi/models/fumi.py:5-6,91,96,100,159 and fumi/models/maml.py:8-9,15-33).

The *protocol* lives here -- ``module(x, params=OrderedDict)``, ``meta_named_parameters()``, ``get_subdict``,
``gradient_update_parameters`` -- so that ``im_net`` / ``PureImageNetwork`` keep the reference's attribute names and
``state_dict`` keys.  The arithmetic of ``evaluate``'s inner loop is NOT done through these modules (it hands the parameter
tensors to the fused HIP step); ``forward`` here runs on the engine's exported linear ops and is differentiable to any order
(``_EngineMatmulNT``: its backward is expressed in the same op), so a hand-written loop in the reference's style --
``loss = F.cross_entropy(model(x, params=p), y); p = gradient_update_parameters(model, loss, params=p, ...)`` -- works on the
GPU, second-order outer gradient included.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import engine as _engine


class MetaModule(nn.Module):
    def meta_named_parameters(self, prefix='', recurse=True):
        mods = self.named_modules(prefix=prefix) if recurse else [(prefix, self)]
        seen = set()
        for mod_prefix, mod in mods:
            if not isinstance(mod, MetaModule):
                continue
            for name, p in mod._parameters.items():
                if p is None or id(p) in seen:
                    continue
                seen.add(id(p))
                yield (mod_prefix + ('.' if mod_prefix else '') + name), p

    def meta_parameters(self, recurse=True):
        for _, p in self.meta_named_parameters(recurse=recurse):
            yield p

    def get_subdict(self, params, key=None):
        if params is None:
            return None
        if key is None:
            return params
        head = key + '.'
        picked = OrderedDict((k[len(head):], v) for k, v in params.items() if k.startswith(head))
        return picked or None


class _EngineMatmulNT(torch.autograd.Function):
    """y [M,N] = x [M,K] . w [N,K]^T on fumi_hip_linear_fwd.  Both gradients are products of the same form
    (dx = dy . (w^T)^T, dw = dy^T . (x^T)^T), built from this Function again: double backward (create_graph=True) works."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _engine.get_engine().linear(x.contiguous(), w.contiguous(), None, act=0)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _EngineMatmulNT.apply(dy, w.t()) if ctx.needs_input_grad[0] else None
        dw = _EngineMatmulNT.apply(dy.t(), x.t()) if ctx.needs_input_grad[1] else None
        return dx, dw


class _EngineSgdUpdate(torch.autograd.Function):
    """p - step_size * g on fumi_hip_sgd_axpy (the reference's fast-weight update, torchmeta gradient_update_parameters)."""

    @staticmethod
    def forward(ctx, p, g, step_size):
        ctx.step_size = step_size
        return _engine.get_engine().sgd_axpy(p.contiguous(), step_size, g.contiguous())

    @staticmethod
    def backward(ctx, d):
        return d, -ctx.step_size * d, None


def engine_linear(x, w, b=None):
    """F.linear on the engine's GEMM, differentiable to any order."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if not (x2.requires_grad or w.requires_grad or (b is not None and b.requires_grad)) or not torch.is_grad_enabled():
        return _engine.get_engine().linear(x2, w, b, act=0).reshape(*lead, w.shape[0])
    y = _EngineMatmulNT.apply(x2, w)
    if b is not None:
        y = y + b
    return y.reshape(*lead, w.shape[0])


def gradient_update_parameters(model, loss, params=None, step_size=0.5, first_order=False):
    """torchmeta.utils.gradient_based.gradient_update_parameters (call sites fumi/models/fumi.py:172-176,
    fumi/models/maml.py:173-177): one SGD step on ``params`` (default: the model's meta-parameters), keeping the graph
    unless ``first_order``.  ``step_size`` may be a float or a dict of per-parameter step sizes."""
    if not isinstance(model, MetaModule):
        raise ValueError('The model must be an instance of `MetaModule`, got `{0}`'.format(type(model)))
    if params is None:
        params = OrderedDict(model.meta_named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()), create_graph=not first_order)
    updated = OrderedDict()
    for (name, p), g in zip(params.items(), grads):
        ss = step_size[name] if isinstance(step_size, (dict, OrderedDict)) else step_size
        updated[name] = _EngineSgdUpdate.apply(p, g, float(ss))
    return updated


class MetaLinear(nn.Linear, MetaModule):
    def forward(self, input, params=None):
        if params is None:
            params = OrderedDict(self.named_parameters())
        return engine_linear(input, params['weight'], params.get('bias', None))


class MetaSequential(nn.Sequential, MetaModule):
    def forward(self, input, params=None):
        for name, module in self._modules.items():
            if isinstance(module, MetaModule):
                input = module(input, params=self.get_subdict(params, name))
            else:
                input = module(input)
        return input
