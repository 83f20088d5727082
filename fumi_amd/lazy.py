"""Deferred device->host read-back of a step's scalar statistics.

The reference returns ``outer_loss.detach().cpu().numpy()`` (fumi/models/fumi.py:195), a blocking copy that idles the GPU
until the host has queued the next meta-batch.  Here the copy of the two floats is issued asynchronously into pinned memory
right after the step's kernels and the returned objects behave like the reference's 0-d arrays (``float(x)``,
``np.asarray(x)``, arithmetic, comparisons, formatting) but only wait -- for their own copy, via an event -- when first
read.  Set ``FUMI_SYNC_STATS=1`` to get plain ``numpy`` 0-d arrays (one synchronisation per step, like the reference)."""
import os

import numpy as np
import torch

SYNC = os.environ.get("FUMI_SYNC_STATS", "0") == "1"


_RING = {}          # (shape, dtype) -> [list of (pinned host tensor, event, owner LazyStats or None), next index]
_RING_DEPTH = 16


class LazyStats:
    """Pinned staging buffers and events come from a small ring (allocating pinned memory every step costs ~15 us of host
    time); a buffer is only reused after its previous owner has been read or is forced to materialise first."""

    def __init__(self, dev_tensor):
        t = dev_tensor.detach()
        self._np = None
        if t.is_cuda:
            key = (tuple(t.shape), t.dtype)
            ring = _RING.get(key)
            if ring is None:
                ring = _RING[key] = [[[torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event(), None]
                                      for _ in range(_RING_DEPTH)], 0]
            slot = ring[0][ring[1]]
            ring[1] = (ring[1] + 1) % _RING_DEPTH
            if slot[2] is not None:
                slot[2]._detach()                 # an unread predecessor keeps its value (copy out of the pinned buffer)
            self._host, self._ev, self._slot = slot[0], slot[1], slot
            slot[2] = self
            self._host.copy_(t, non_blocking=True)
            self._ev.record(torch.cuda.current_stream(t.device))
        else:
            self._host, self._ev, self._slot = t.clone(), None, None

    def _detach(self):
        if self._np is None:
            self._ev.synchronize()
            self._np = self._host.numpy().copy()
        if self._slot is not None:
            self._slot[2] = None
            self._slot = None

    def get(self):
        if self._np is None:
            if self._ev is not None:
                self._ev.synchronize()
                self._np = self._host.numpy().copy()
                if self._slot is not None:
                    self._slot[2] = None
                    self._slot = None
            else:
                self._np = self._host.numpy()
        return self._np


class LazyScalar(np.lib.mixins.NDArrayOperatorsMixin):
    """0-d array look-alike backed by one element of a LazyStats buffer."""
    __slots__ = ("_s", "_i")
    shape, ndim = (), 0

    def __init__(self, stats, i):
        self._s, self._i = stats, i

    def _v(self):
        return self._s.get()[self._i]

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._v())
        return a.astype(dtype) if dtype is not None else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        inputs = tuple(np.asarray(x) if isinstance(x, LazyScalar) else x for x in inputs)
        return getattr(ufunc, method)(*inputs, **kw)

    def __float__(self):
        return float(self._v())

    def __int__(self):
        return int(self._v())

    def __bool__(self):
        return bool(self._v())

    def item(self):
        return self._v().item()

    @property
    def dtype(self):
        return self._s.get().dtype

    def __repr__(self):
        return repr(self._v())

    __str__ = __repr__

    def __format__(self, spec):
        return format(self._v(), spec)


def scalars(dev_tensor, n):
    """n scalars read back from the first n elements of a device tensor (lazily unless FUMI_SYNC_STATS=1)."""
    if SYNC:
        host = dev_tensor.detach().cpu().numpy()
        return tuple(host[i] for i in range(n))
    st = LazyStats(dev_tensor)
    return tuple(LazyScalar(st, i) for i in range(n))
