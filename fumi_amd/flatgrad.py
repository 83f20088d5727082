"""One flat fp32 buffer [all trainable gradients | sum loss | sum acc]: the engine writes gradients straight into
views of it, a single all-reduce covers everything (fumi_amd/dist.py), and ``.grad`` of every parameter is a view."""
import torch


class FlatGrads:
    def __init__(self, params, extra=2):
        self.params = list(params)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n + extra, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.tail = self.flat[off:]
        self._splits = {}

    def split(self, n):
        """(views[:n], views[n:]) as list objects that stay the same from call to call (hip.py validates a list once)."""
        sp = self._splits.get(n)
        if sp is None:
            sp = self._splits[n] = (self.views[:n], self.views[n:])
        return sp

    def matches(self, params):
        params = list(params)
        return (len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params))
                and self.flat.device == params[0].device)

    def attach(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not v:                  # stays attached from step to step: the engine overwrites the whole buffer
                p.grad = v


class ParamWatch:
    """Cheap per-step check that a model's cached parameter views are still current: every parameter must still be the object
    registered under its name in its owner module (``module._parameters``) and must still live at the same address
    (``param.data = ...`` and ``.to()`` move it).  ~1 us for a dozen parameters."""

    def __init__(self, module, params):
        byid = {}
        for m in module.modules():
            for n, q in m._parameters.items():
                if q is not None:
                    byid[id(q)] = (m._parameters, n)
        self.refs = [(byid[id(q)][0], byid[id(q)][1], q, q.data_ptr()) for q in params]

    def valid(self):
        for d, n, q, ptr in self.refs:
            if d.get(n) is not q or q.data_ptr() != ptr:
                return False
        return True
