"""Functional-parameter protocol of the reference's third-party dependency torchmeta 1.7.0
(requirements.txt:10; call sites fumi/models/fumi.py:5-6,91,96,100,159 and fumi/models/maml.py:8-9,15-33).

Only the *protocol* lives here -- ``module(x, params=OrderedDict)``, ``meta_named_parameters()``,
``get_subdict`` -- so that ``im_net`` / ``PureImageNetwork`` keep the reference's attribute names and
``state_dict`` keys.  The arithmetic of the inner loop is NOT done through these modules: ``evaluate`` hands the
parameter tensors to the HIP engine.  ``forward`` here is an inference helper that runs on the engine's linear op.
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


class MetaLinear(nn.Linear, MetaModule):
    def forward(self, input, params=None):
        if params is None:
            params = OrderedDict(self.named_parameters())
        w, b = params['weight'], params.get('bias', None)
        lead = input.shape[:-1]
        y = _engine.get_engine().linear(input.reshape(-1, input.shape[-1]), w, b, act=0)
        return y.reshape(*lead, w.shape[0])


class MetaSequential(nn.Sequential, MetaModule):
    def forward(self, input, params=None):
        for name, module in self._modules.items():
            if isinstance(module, MetaModule):
                input = module(input, params=self.get_subdict(params, name))
            else:
                input = module(input)
        return input
