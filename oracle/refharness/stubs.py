"""Import shims that let the reference's model files be imported UNMODIFIED in this container.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
(``fumi_amd/``); it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by
``bench.py``'s ``cpu_baseline`` leg as the checker / the reported CPU baseline.

The reference (/root/reference, s-a-malik/fumi) imports four third-party
packages that are absent from this image and cannot be installed (no network):

* ``wandb==0.10.26``       -- logging only, carries no arithmetic  -> no-op stub
* ``gensim==4.0.1``        -- GloVe download (needs network)       -> stub returning a
                              caller-provided fake KeyedVectors table
* ``transformers.AdamW``   -- removed from transformers 5.x        -> alias to torch.optim.AdamW
* ``torchmeta==1.7.0``     -- carries hot-path arithmetic (requirements.txt:10).  Its source
                              is NOT under /root/reference, so it is restated here from its
                              published contract.  Parity with the
                              *real* torchmeta is therefore "unpinned"; everything the
                              reference's own files compute is pinned by running them.

Call sites of the torchmeta symbols in the reference:
  fumi/models/fumi.py:5-6,91,96,100,159,172-176 ; fumi/models/maml.py:8-9,15,25,28,29,32,173-177
"""
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# torchmeta 1.7.0 behavioural restatement (modules + gradient_based)
# ----------------------------------------------------------------------------------------------
class MetaModule(nn.Module):
    """torchmeta.modules.MetaModule: an nn.Module whose forward takes ``params=``."""

    def __init__(self):
        super().__init__()
        self._children_modules_parameters_cache = dict()

    def meta_named_parameters(self, prefix='', recurse=True):
        gen = self._named_members(
            lambda module: module._parameters.items() if isinstance(module, MetaModule) else [],
            prefix=prefix, recurse=recurse)
        for elem in gen:
            yield elem

    def meta_parameters(self, recurse=True):
        for _, param in self.meta_named_parameters(recurse=recurse):
            yield param

    def get_subdict(self, params, key=None):
        if params is None:
            return None
        if key is None:
            return params
        pre = key + '.'
        sub = OrderedDict((k[len(pre):], v) for k, v in params.items() if k.startswith(pre))
        return sub if len(sub) > 0 else None


class MetaLinear(nn.Linear, MetaModule):
    __doc__ = nn.Linear.__doc__

    def forward(self, input, params=None):
        if params is None:
            params = OrderedDict(self.named_parameters())
        bias = params.get('bias', None)
        return F.linear(input, params['weight'], bias)


class MetaSequential(nn.Sequential, MetaModule):
    __doc__ = nn.Sequential.__doc__

    def forward(self, input, params=None):
        for name, module in self._modules.items():
            if isinstance(module, MetaModule):
                input = module(input, params=self.get_subdict(params, name))
            elif isinstance(module, nn.Module):
                input = module(input)
            else:
                raise TypeError(type(module))
        return input


def gradient_update_parameters(model, loss, params=None, step_size=0.5, first_order=False):
    """torchmeta.utils.gradient_based.gradient_update_parameters (one SGD step, graph kept)."""
    if not isinstance(model, MetaModule):
        raise ValueError('model must be a MetaModule')
    if params is None:
        params = OrderedDict(model.meta_named_parameters())
    grads = torch.autograd.grad(loss, params.values(), create_graph=not first_order)
    updated = OrderedDict()
    if isinstance(step_size, (dict, OrderedDict)):
        for (name, param), grad in zip(params.items(), grads):
            updated[name] = param - step_size[name] * grad
    else:
        for (name, param), grad in zip(params.items(), grads):
            updated[name] = param - step_size * grad
    return updated


# ----------------------------------------------------------------------------------------------
# stub installation
# ----------------------------------------------------------------------------------------------
class _Anything:
    """Absorbs any attribute access / call (wandb.run.dir, wandb.config.update, ...)."""

    def __init__(self, name='stub'):
        self._name = name

    def __getattr__(self, k):
        return _Anything(self._name + '.' + k)

    def __call__(self, *a, **k):
        return _Anything(self._name + '()')

    def __str__(self):
        return '/tmp/fumi_oracle_wandb'

    __fspath__ = __str__


FAKE_KEYED_VECTORS = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(fake_vectors=None):
    """Install the stubs into sys.modules (idempotent)."""
    if fake_vectors is not None:
        FAKE_KEYED_VECTORS.clear()
        FAKE_KEYED_VECTORS.update(fake_vectors)

    if 'wandb' not in sys.modules or not getattr(sys.modules['wandb'], '_is_fumi_stub', False):
        w = _mod('wandb', _is_fumi_stub=True)
        for fn in ('init', 'log', 'watch', 'save', 'restore', 'finish'):
            setattr(w, fn, _Anything('wandb.' + fn))
        w.run = _Anything('wandb.run')
        w.config = _Anything('wandb.config')

    if 'gensim' not in sys.modules:
        g = _mod('gensim')
        gd = _mod('gensim.downloader', load=lambda name: FAKE_KEYED_VECTORS)
        g.downloader = gd

    import transformers  # present (5.x) but without AdamW
    if not hasattr(transformers, 'AdamW'):
        transformers.AdamW = torch.optim.AdamW

    if 'torchmeta' not in sys.modules:
        tm = _mod('torchmeta')
        tmm = _mod('torchmeta.modules', MetaModule=MetaModule, MetaLinear=MetaLinear,
                   MetaSequential=MetaSequential)
        tmu = _mod('torchmeta.utils')
        tmg = _mod('torchmeta.utils.gradient_based',
                   gradient_update_parameters=gradient_update_parameters)
        tm.modules, tm.utils, tmu.gradient_based = tmm, tmu, tmg


REFERENCE_ROOT = '/root/reference/fumi'


def import_reference():
    """Import the reference's model modules unmodified.  Returns (fumi, maml, am3, utils, common)."""
    import os
    if not os.path.isdir(REFERENCE_ROOT):
        raise FileNotFoundError(REFERENCE_ROOT + ' is not present (it never travels to the GPU box)')
    install()
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models.fumi as ref_fumi
    import models.maml as ref_maml
    import models.am3 as ref_am3
    import models.common as ref_common
    import utils.utils as ref_utils
    return ref_fumi, ref_maml, ref_am3, ref_utils, ref_common


def import_reference_clip():
    """models.clip of the reference, unmodified (after import_reference)."""
    import_reference()
    import models.clip as ref_clip
    return ref_clip
