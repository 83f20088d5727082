"""Deferred device->host read-back of a step's scalar statistics.

The reference returns ``outer_loss.detach().cpu().numpy()`` (fumi/models/fumi.py:195), a blocking copy that idles the GPU
until the host has queued the next meta-batch.  Here a one-wave launch right after the step's kernels stores the two floats
and then a sequence number into pinned host memory with system-scope stores (``fumi_hip_publish_scalars``), and the returned
objects behave like the reference's 0-d arrays (``float(x)``, ``np.asarray(x)``, arithmetic, comparisons, formatting) but
only wait -- by polling their sequence number -- when first read.  (An async copy plus an event record did the same job with
a ~10 us bubble on the stream per step.)  Set ``FUMI_SYNC_STATS=1`` to get plain ``numpy`` 0-d arrays (one synchronisation per
step, like the reference); ``FUMI_STATS_EVENT=1`` selects the copy + event form."""
import os
import time

import numpy as np
import torch

SYNC = os.environ.get("FUMI_SYNC_STATS", "0") == "1"
USE_EVENT = os.environ.get("FUMI_STATS_EVENT", "0") == "1"
_SEQ = [0]
_PUB_RING = [[], 0]    # [[pinned float32[16], uint64 view of its last 8 bytes, owner], ...], next index


_RING = {}          # (shape, dtype) -> [list of (pinned host tensor, event, owner LazyStats or None), next index]
_RING_DEPTH = 16


class LazyStats:
    """Pinned staging buffers and events come from a small ring (allocating pinned memory every step costs ~15 us of host
    time); a buffer is only reused after its previous owner has been read or is forced to materialise first."""

    def __init__(self, dev_tensor, defer=False):
        t = dev_tensor.detach()
        self._np = None
        self._seq = 0
        if (t.is_cuda and not USE_EVENT and t.dtype == torch.float32 and t.dim() == 1 and t.numel() <= 14
                and t.is_contiguous()):
            from . import hip
            ring = _PUB_RING
            if not ring[0]:
                for _ in range(_RING_DEPTH):
                    h = torch.zeros(16, dtype=torch.float32, pin_memory=True)
                    ring[0].append([h, h.numpy().view(np.uint64)[7:8], None])
            slot = ring[0][ring[1]]
            ring[1] = (ring[1] + 1) % _RING_DEPTH
            if slot[2] is not None:
                slot[2]._detach()                 # an unread predecessor takes its value out first (and its launch has run)
            _SEQ[0] += 1
            self._seq, self._n = _SEQ[0], t.numel()
            self._host, self._flag, self._slot, self._ev = slot[0], slot[1], slot, None
            slot[2] = self
            self._src = t                          # (kept alive until the deferred stores have been issued)
            if defer:
                hip.publish_flush(hip.Workspace.get(t.device), t.device)      # (a publication stranded by an exception)
            hip.publish_scalars(hip.Workspace.get(t.device), t, self._n, self._host, self._seq, defer=defer)
        elif t.is_cuda:
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

    def _wait_seq(self):
        flag, seq = self._flag, self._seq
        if flag[0] != seq:
            t0 = time.perf_counter()
            while flag[0] != seq:
                if time.perf_counter() - t0 > 20.0:
                    torch.cuda.synchronize()
                    if flag[0] != seq:
                        raise RuntimeError("the step's statistics never arrived in host memory (GPU fault?)")
        self._np = self._host.numpy()[:self._n].copy()

    def _detach(self):
        if self._np is None:
            if self._seq:
                self._wait_seq()
            else:
                self._ev.synchronize()
                self._np = self._host.numpy().copy()
        if self._slot is not None:
            self._slot[2] = None
            self._slot = None

    def get(self):
        if self._np is None:
            if self._seq:
                self._wait_seq()
                if self._slot is not None:
                    self._slot[2] = None
                    self._slot = None
            elif self._ev is not None:
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


def scalars(dev_tensor, n, defer=False):
    """n scalars read back from the first n elements of a device tensor (lazily unless FUMI_SYNC_STATS=1).
    ``defer``: the values ride on the optimizer step's launch that follows; the caller must call ``flush`` after it."""
    if SYNC:
        host = dev_tensor.detach().cpu().numpy()
        return tuple(host[i] for i in range(n))
    st = LazyStats(dev_tensor, defer=defer and dev_tensor.is_cuda and not USE_EVENT)
    return tuple(LazyScalar(st, i) for i in range(n))


def flush(device):
    """Issue a deferred publication that no optimizer launch has carried (other optimizers, closures, CPU runs: no-op)."""
    if SYNC or USE_EVENT or getattr(device, "type", "cpu") != "cuda":
        return
    from . import hip
    hip.publish_flush(hip.Workspace.get(device), device)
